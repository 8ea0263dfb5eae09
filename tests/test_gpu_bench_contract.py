"""`bench.py` on a small graph: one JSON line with the fields the round driver reads (metric,
value, unit, n_gpus, steps, warmup, ms_per_step, dtype, config.workload), the roofline object
(bound, achieved, peak, unit, frac, traffic) and the CPU baseline timed in the same run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contract_fields():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["BENCH_PRIME_SECONDS"] = "0.2"
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--scale", "0.05", "--parts",
                           "12", "--steps", "4", "--warmup", "2", "--cpu-seconds", "1"],
                          env=env, capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout[-2000:]
    rec = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in rec, key
    assert rec["n_gpus"] == 1 and rec["steps"] == 4 and rec["warmup"] == 2
    assert rec["unit"] == "edges/s" and rec["dtype"] == "f32" and rec["vs_baseline"] is None
    assert rec["value"] > 0 and rec["ms_per_step"] > 0 and "workload" in rec["config"]
    roof = rec["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert roof["traffic"] is None      # (counters are only quoted for the workload they were taken on)
    assert 0 < roof["avg_launch_ms"] <= roof["avg_interval_ms"]
    cpu = rec["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cpu, key
    assert cpu["value"] > 0 and cpu["kind"] in ("reference", "port")
