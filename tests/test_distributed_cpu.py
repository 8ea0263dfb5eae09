"""Partition-parallel path with world_size 2 and 4 (uneven shares, padding rounds) on the CPU (gloo): the row-sharded
History tables after a refresh sweep must equal the single-process tables, the push/pull
forward of a rank's partition must equal the single-process forward of that partition, and
the all-reduced gradient must be the mean of the per-rank gradients."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import model_cases as mc
from cpu_shim import cpu_kernels

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["gcn2", "sage", "appnp"]
TOL = dict(rtol=1e-4, atol=1e-4)


def check_against_single_process(ranks, world, dev=None):
    """The per-rank results of tests/dist_worker.py against one process working on the whole
    graph: on the CPU through the oracle kernels, or (``dev``) on a GPU through the HIP path."""
    import contextlib
    from torch_geometric_autoscale.loader import partition_batch
    import torch_geometric_autoscale as tga
    data, ptr = mc.make_graph()
    n_parts = ptr.numel() - 1
    on_gpu = dev is not None

    def to_dev(t):
        return t.to(dev) if on_gpu else t

    def host(t):
        return t.detach().cpu().numpy()

    with (contextlib.nullcontext() if on_gpu else cpu_kernels()):
        for name in NAMES:
            model = mc.make_model(name, data, data.x.size(1), 5, emb_device=dev if on_gpu else "cpu")
            if on_gpu:
                model = model.to(dev)
            model.eval()
            loader = tga.EvalSubgraphLoader(data, ptr, batch_size=1, log=False)
            out = model.mini_inference_vr(loader)
            grads, cover = [], []
            for r, res in enumerate(ranks):
                lo, hi = res["row_lo"], res["row_hi"]
                cover.append((lo, hi))
                np.testing.assert_allclose(res[f"{name}/out"].numpy(), host(out[lo:hi]), **TOL)
                for i in range(model.num_layers):
                    np.testing.assert_allclose(res[f"{name}/hist{i}"].numpy(),
                                               host(model.histories[i].emb[lo:hi]), **TOL)
                    np.testing.assert_allclose(res[f"{name}/hist_ag{i}"].numpy(),
                                               host(model.histories_ag[i].emb[lo:hi]), **TOL)
                part = r * n_parts // world
                sub = partition_batch(data, ptr, [part])
                b = sub.batch_size
                ret = model(to_dev(sub.data.x), to_dev(sub.data.adj_t), b, sub.n_id, sub.offset,
                            sub.count)
                np.testing.assert_allclose(res[f"{name}/gas_out"].numpy(), host(ret["out"]), **TOL)
                loss = torch.nn.functional.cross_entropy(ret["out"], to_dev(sub.data.y[:b]))
                model.zero_grad()
                loss.backward()
                grads.append(torch.cat([p.grad.reshape(-1) for p in model.parameters()
                                        if p.grad is not None]).cpu())
                sub = partition_batch(data, ptr, [part], in_batch_only=True)
                ret = model.VR_call(to_dev(sub.data.x), to_dev(sub.data.adj_t), sub.batch_size,
                                    sub.n_id, sub.offset, sub.count)
                np.testing.assert_allclose(res[f"{name}/vr_out"].numpy(), host(ret["out"]), **TOL)
            assert cover[0][0] == 0 and cover[-1][1] == data.num_nodes
            assert all(a[1] == b_[0] for a, b_ in zip(cover, cover[1:]))
            mean_grad = torch.stack(grads).mean(0).numpy()
            for res in ranks:                      # identical on every rank after the all-reduce
                np.testing.assert_allclose(res[f"{name}/grad"].numpy(), mean_grad, rtol=1e-4,
                                           atol=1e-6 if not on_gpu else 1e-5)


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("mode", ["buffer", "in_place"])
@pytest.mark.parametrize("world", [2, 4])
def test_sharded_run_matches_single_process(world, mode, tmp_path):
    """``mode``: out-of-batch rows gathered into a halo buffer, or read in place -- own rows
    where they are in the table shard, rows of peers received into the shard's tail."""
    env = dict(os.environ, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path), ",".join(NAMES), mode]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-3000:]
    ranks = [torch.load(tmp_path / f"rank{r}.pt") for r in range(world)]
    if mode == "in_place":      # the in-place exchange is what ran (where the model allows it)
        for res in ranks:
            assert res["gcn2/calls"]["in_place"] > 0 and res["gcn2/calls"]["buffer"] == 0, res["gcn2/calls"]

    check_against_single_process(ranks, world)


def check_pipeline_against_single_process(ranks, world, dev=None):
    """The steps of tests/dist_worker.py's pipeline case against one process that holds whole
    tables and hands every step its out-of-batch rows as the exchange running one step ahead
    delivers them: rows of peers as the pushes of the steps before the previous one left them,
    rows of the rank itself (read in place) as the previous step left them.  Without the
    look-ahead (one batch per rank) all of them as the previous step left them."""
    import contextlib
    import dist_worker as dw
    from torch_geometric_autoscale.loader import partition_batch
    data, ptr = mc.make_graph(parts=dw.PIPELINE_PARTS)
    on_gpu = dev is not None
    ahead = ranks[0]["ahead"]
    assert all(r["ahead"] == ahead for r in ranks)
    for r in ranks:     # consecutive batches of a rank's cycle never share a region
        reg = r["regions"]
        if ahead:
            assert all(reg[k] != reg[(k + 1) % len(reg)] for k in range(len(reg))), reg
    with (contextlib.nullcontext() if on_gpu else cpu_kernels()):
        model = mc.make_model(dw.PIPELINE_MODEL, data, data.x.size(1), 5,
                              emb_device=dev if on_gpu else "cpu")
        if on_gpu:
            model = model.to(dev)
        model.eval()
        tables = [h.emb for h in model.histories]
        live = [t.to(tables[0].device) for t in dw.pipeline_tables(model, data.num_nodes)]
        before = [[t.clone() for t in live]]        # before[i]: tables before step i's pushes
        n_steps = dw.pipeline_steps(world)
        for i in range(n_steps + 2):
            if i:
                before.append([t.clone() for t in live])
            extra = i >= n_steps        # the two steps on batch 0 after the drain: no look-ahead
            for res in ranks:
                part = res["parts"][0 if extra else i % len(res["parts"])]
                sub = partition_batch(data, ptr, [part])
                b = sub.batch_size
                lo, hi = res["row_lo"], res["row_hi"]
                for t, old, new in zip(tables, before[max(i - 1, 0)], before[i]):
                    t.copy_(old if ahead and not extra else new)
                    t[lo:hi] = new[lo:hi]
                with torch.no_grad():
                    ret = model(sub.data.x.to(dev) if on_gpu else sub.data.x,
                                sub.data.adj_t.to(dev) if on_gpu else sub.data.adj_t, b, sub.n_id,
                                sub.offset, sub.count)
                got = res[f"extra{i - n_steps}"] if extra else res[f"step{i}"]
                np.testing.assert_allclose(got.numpy(), ret["out"].cpu().numpy(), **TOL,
                                           err_msg=f"step {i}, partition {part}")
                rows = sub.n_id[:b].to(tables[0].device)
                for t, l_ in zip(tables, live):     # this step's pushes
                    l_[rows] = t[rows]
        for res in ranks:
            lo, hi = res["row_lo"], res["row_hi"]
            for k in range(1, model.num_layers):
                np.testing.assert_allclose(res[f"hist{k}"].numpy(), live[k][lo:hi].cpu().numpy(), **TOL)


def _run_pipeline_ranks(world, tmp_path, extra_env=None):
    env = dict(os.environ, OMP_NUM_THREADS="2", **(extra_env or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path), "-", "pipeline"]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-3000:]
    return [torch.load(tmp_path / f"rank{r}.pt") for r in range(world)]


@pytest.mark.parametrize("world", [2, 3, 4])
def test_exchange_one_step_ahead_matches_single_process(world, tmp_path):
    """Shares of 3 + 4, 2 + 2 + 3 and 1 + 2 + 1 + 2 partitions, of which every rank uses as many
    as the smallest share: cycles of 3 (odd: three tail regions), 2 and 1 batches (nothing to
    look ahead to: exchanges at the top of every step)."""
    ranks = _run_pipeline_ranks(world, tmp_path)
    assert ranks[0]["ahead"] == (world < 4)
    check_pipeline_against_single_process(ranks, world)


def test_halo_plan_orders_own_rows_first():
    from torch_geometric_autoscale import distributed as D
    ptr = torch.tensor([0, 10, 20, 30, 40])
    ctx = D.PartitionContext(ptr, rank=1, world=2)
    assert (ctx.row_lo, ctx.row_hi, ctx.local_parts) == (20, 40, [2, 3])
    assert ctx.owner_of(torch.tensor([0, 19, 20, 39])).tolist() == [0, 0, 1, 1]
    one = D.PartitionContext(ptr, rank=0, world=1)
    plan, order = D.build_halo_plan(one, torch.tensor([33, 2, 17]))
    assert order.tolist() == [0, 1, 2] and plan.own_local.tolist() == [33, 2, 17]
    assert plan.n_halo == 3 and plan.recv_counts == [0]
