"""`bench.py --gpus N` starts its own ranks when no launcher did: the spawn path on the CPU
(gloo), the exit code of a failing rank, and the refusal to run N ranks on fewer GPUs."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def test_gpus_flag_spawns_that_many_ranks():
    proc = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--launch-selftest"],
                          env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 3 and rec["rank_sum"] == 6.0 and rec["gpus_flag"] == 3


def test_failing_rank_fails_the_command():
    proc = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-selftest"],
                          env=_clean_env(BENCH_SELFTEST_FAIL_RANK="1"), capture_output=True,
                          text=True, timeout=300)
    assert proc.returncode == 7
    assert "rank 1 exited with code 7" in proc.stderr


def test_more_ranks_than_gpus_is_refused():
    import torch
    n = torch.cuda.device_count() + 1 if torch.cuda.device_count() else 2
    proc = subprocess.run([sys.executable, BENCH, "--gpus", str(n)], env=_clean_env(),
                          capture_output=True, text=True, timeout=300)
    assert proc.returncode != 0
    assert f"--gpus {n} needs {n} visible GPUs" in proc.stderr
    assert not [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]


def test_launcher_given_ranks_must_match_the_flag():
    proc = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-selftest"],
                          env=_clean_env(WORLD_SIZE="1", RANK="0"), capture_output=True,
                          text=True, timeout=300)
    # a launcher is in charge (WORLD_SIZE set): nothing is spawned, one rank runs
    assert proc.returncode == 0 and json.loads(proc.stdout.strip().splitlines()[-1])["n_gpus"] == 1
