"""The partition-parallel code path on one MI355X over RCCL (backend "nccl", world size 1, the
collectives issued although they move nothing): process-group set-up on the GPU, halo plans,
`all_to_all_single` on device tensors, stream ordering of the pending exchange, the flat gradient
all-reduce -- against the unsharded run of the same model.  Tables in HBM and in pinned host
memory (the placement of BASELINE.json's host-offload configuration)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist

import model_cases as mc

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-4, atol=1e-4)


@pytest.fixture(scope="module")
def rccl():
    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    from torch_geometric_autoscale import distributed as D
    D.enable_collective_records()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=dev)
    yield dev
    dist.destroy_process_group()


@pytest.mark.parametrize("tables", ["hbm", "hbm_in_place", "host"])
@pytest.mark.parametrize("name", ["gcn2", "sage", "appnp"])
def test_sharded_path_over_rccl_matches_unsharded(name, tables, rccl):
    import torch_geometric_autoscale as tga
    from torch_geometric_autoscale import _lib, distributed as D
    from torch_geometric_autoscale.loader import partition_batch
    dev = rccl
    data, ptr = mc.make_graph()
    f_in = data.x.size(1)

    ref = mc.make_model(name, data, f_in, 5, emb_device=dev).to(dev)
    ref.eval()
    out_ref = ref.mini_inference_vr(tga.EvalSubgraphLoader(data, ptr, batch_size=1, log=False))

    ctx = D.PartitionContext(ptr, exercise_collectives=True)
    assert (ctx.rank, ctx.world, ctx.local_rows) == (0, 1, data.num_nodes)
    in_place = tables == "hbm_in_place"
    tables = "hbm" if in_place else tables
    emb_device = dev if tables == "hbm" else None
    model = mc.make_model(name, data, f_in, 5, emb_device=emb_device, pool_size=2,
                          buffer_size=4 * data.num_nodes if tables == "host" else None).to(dev)
    model.load_state_dict(ref.state_dict())
    model.set_partition_context(ctx)
    if tables == "host":
        assert model.histories[0].emb.is_pinned() and not model.histories[0].emb.is_cuda
    model.eval()
    batches = D.sharded_eval_batches(data, ptr, ctx, batch_size=1)
    calls = {"in_place": 0}
    if in_place:    # out-of-batch rows read in place from the shard (+ its tail for peers' rows)
        D.reserve_halo_tails(model, batches)
        assert all(h._base is not None for h in model.histories)
        original = D.HaloPlan.start_exchange_in_place

        def counted(self, *a, **k):
            calls["in_place"] += 1
            return original(self, *a, **k)
        D.HaloPlan.start_exchange_in_place = counted
    out = model.mini_inference_vr(batches)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), out_ref.cpu().numpy(), **TOL)
    for i in range(model.num_layers):
        # layer inputs are copies of computed rows; aggregates are sums in another order only
        # where the halo rows arrive in plan order instead of first-seen order
        np.testing.assert_allclose(model.histories[i].emb.cpu().numpy(),
                                   ref.histories[i].emb.cpu().numpy(), **TOL)
        np.testing.assert_allclose(model.histories_ag[i].emb.cpu().numpy(),
                                   ref.histories_ag[i].emb.cpu().numpy(), **TOL)

    # one push/pull training step and one incremental-aggregation step on a partition
    model.train()
    ref.train()
    for vr in (False, True):
        sub = D.sharded_partition_batch(data, ptr, [3], ctx, in_batch_only=vr)
        sub_ref = partition_batch(data, ptr, [3], in_batch_only=vr)
        b = sub.batch_size
        outs, grads = [], []
        for m, s in ((model, sub), (ref, sub_ref)):
            call = m.VR_call if vr else m
            o = call(s.data.x.to(dev), s.data.adj_t.to(dev), b, s.n_id, s.offset, s.count)["out"]
            loss = torch.nn.functional.cross_entropy(o, s.data.y[:b].to(dev))
            m.zero_grad()
            loss.backward()
            if m is model:
                D.allreduce_gradients(m, ctx)
            outs.append(o.detach().cpu().numpy())
            grads.append(torch.cat([p.grad.reshape(-1) for p in m.parameters()
                                    if p.grad is not None]).cpu().numpy())
        np.testing.assert_allclose(outs[0], outs[1], **TOL)
        np.testing.assert_allclose(grads[0], grads[1], rtol=1e-3, atol=1e-5)
    torch.cuda.synchronize()
    _lib.check_device_errors()
    if in_place:
        D.HaloPlan.start_exchange_in_place = original
        if name == "gcn2":
            assert calls["in_place"] > 0


@pytest.mark.parametrize("ahead", [False, True])
def test_recorded_step_between_eager_collectives_trains_like_eager_steps(ahead, rccl):
    """A partition-parallel step split at its collectives: halo exchanges (into the table
    tails) and the flat gradient all-reduce are issued eagerly, everything between them --
    forward with ``halo_ready=True``, loss, backward into the flat gradient buffer -- is ONE
    HIP graph recorded next to the live RCCL process group.  Weights and tables after three
    such steps over two batches equal those of eagerly issued partition-parallel steps.
    ``ahead``: the exchanges of the next step's batch are issued before the current step's
    graph is replayed (with one rank there are no foreign rows, so both forms read the same
    values; what differs is which tail region a batch's row ids point into and that collectives
    are in flight during the replays)."""
    import copy
    import torch_geometric_autoscale as tga
    from torch_geometric_autoscale import _lib, distributed as D
    from torch_geometric_autoscale.graphs import StepGraphs
    dev = rccl
    data, ptr = mc.make_graph()
    f_in = data.x.size(1)
    ctx = D.PartitionContext(ptr, exercise_collectives=True)
    base = mc.make_model("gcn2", data, f_in, 5, emb_device=dev).to(dev)
    subs = [D.sharded_partition_batch(data, ptr, [p], ctx) for p in (2, 5)]
    steps = [0, 1, 0]

    def prepared():
        model = copy.deepcopy(base)
        for h_new, h_old in zip(list(model.histories) + list(model.histories_ag),
                                list(base.histories) + list(base.histories_ag)):
            h_new.emb = h_old.emb.clone()       # (tables are plain attributes: not deep-copied)
        model.set_partition_context(ctx)
        D.reserve_halo_tails(model, subs, regions=2)
        model.train()
        return model, torch.optim.SGD(model.parameters(), lr=0.05)

    def inputs(sub):
        return (sub.data.x.to(dev), sub.data.adj_t.to(dev), sub.batch_size, sub.n_id, sub.offset,
                sub.count), sub.data.y[:sub.batch_size].to(dev)

    # eager partition-parallel steps
    eager, opt = prepared()
    for k in steps:
        args, y = inputs(subs[k])
        loss = torch.nn.functional.cross_entropy(eager(*args)["out"], y)
        opt.zero_grad()
        loss.backward()
        D.allreduce_gradients(eager, ctx)
        opt.step()

    # the same steps, split at the collectives
    model, opt = prepared()
    flat = D.flatten_gradients(model)
    prepared_inputs = [inputs(s) for s in subs]

    def body(k):
        args, y = prepared_inputs[k]
        out = model(*args, halo_ready=True)["out"]
        opt.zero_grad(set_to_none=False)
        torch.nn.functional.cross_entropy(out, y).backward()

    pipe = D.HaloPipeline(model, [inp[0][1] for inp in prepared_inputs], ahead=ahead)
    assert pipe.ahead == ahead
    assert [s.data.adj_t._halo_plan.region for s in subs] == ([0, 1] if ahead else [0, 0])

    def step(k, replay):
        pipe.ready(k)
        graphs.replay(k) if replay else body(k)
        D.allreduce_flat_gradients(flat, ctx)
        opt.step()

    snapshot = copy.deepcopy(model.state_dict())
    tables = [h.emb_with_tail.clone() for h in model.histories]
    graphs = StepGraphs(dev)
    for k in range(len(subs)):                  # lazily built per-batch structures, then record
        graphs.warm(lambda: step(k, replay=False), times=1)
    # (a condition, not a pause: every collective issued so far is marked retired in
    # torch.distributed's own record before the recording starts)
    pipe.drain()
    assert D.quiesce_collectives() == "retired"
    for k in range(len(subs)):
        graphs.capture(k, lambda: body(k), capture_error_mode="thread_local")
    torch.cuda.synchronize()
    model.load_state_dict(snapshot)             # back to the start (in place: addresses stay)
    for h, t in zip(model.histories, tables):
        h.emb_with_tail.copy_(t)
    for k in steps:
        step(k, replay=True)
    pipe.drain()
    torch.cuda.synchronize()
    for (name, p), q in zip(model.named_parameters(), eager.parameters()):
        np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(),
                                   rtol=1e-4, atol=1e-5, err_msg=name)
    for h, g in zip(model.histories, eager.histories):
        np.testing.assert_allclose(h.emb.cpu().numpy(), g.emb.cpu().numpy(), **TOL)
    _lib.check_device_errors()


def _run_gpu_ranks(world, mode, tmp_path):
    """`world` fresh processes of tests/dist_worker.py, rank r on GPU r, backend "nccl"."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, DIST_WORKER_DEVICE="cuda", RANK=str(r), LOCAL_RANK=str(r),
                   WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(root, "tests", "dist_worker.py"), str(tmp_path),
             "gcn2,sage,appnp", mode], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
            text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=300)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r}:\n" + outs[r][-3000:]
    return [torch.load(tmp_path / f"rank{r}.pt") for r in range(world)]


@pytest.mark.parametrize("mode", ["buffer", "in_place"])
def test_one_rank_worker_over_rccl_matches_single_process(mode, tmp_path):
    """The multi-rank worker itself, started as ONE fresh process on GPU 0 (collectives forced
    on): what the two-rank test below runs, on a machine with a single GPU."""
    from test_distributed_cpu import check_against_single_process
    ranks = _run_gpu_ranks(1, mode, tmp_path)
    check_against_single_process(ranks, 1, dev=torch.device("cuda", 0))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
@pytest.mark.parametrize("mode", ["buffer", "in_place"])
def test_two_ranks_over_rccl_match_single_process(mode, tmp_path):
    """Two RCCL ranks, one per GPU: row-sharded tables after the refresh sweep, the push/pull
    forward of each rank's first partition, the incremental-aggregation forward and the
    all-reduced gradient equal what one process computes on the whole graph."""
    from test_distributed_cpu import check_against_single_process
    ranks = _run_gpu_ranks(2, mode, tmp_path)
    for res in ranks:
        if mode == "in_place":
            assert res["gcn2/calls"]["in_place"] > 0 and res["gcn2/calls"]["buffer"] == 0
    check_against_single_process(ranks, 2, dev=torch.device("cuda", 0))


def test_one_rank_exchange_one_step_ahead_over_rccl(tmp_path):
    """The look-ahead exchange (distributed.HaloPipeline: seven batches, three tail regions) as
    ONE fresh process on GPU 0 with the collectives forced on: every step's all-to-all calls
    are issued while the step before still computes."""
    from test_distributed_cpu import check_pipeline_against_single_process
    ranks = _run_gpu_ranks(1, "pipeline", tmp_path)
    assert ranks[0]["ahead"] and sorted(set(ranks[0]["regions"])) == [0, 1, 2]
    check_pipeline_against_single_process(ranks, 1, dev=torch.device("cuda", 0))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_two_ranks_exchange_one_step_ahead_over_rccl(tmp_path):
    from test_distributed_cpu import check_pipeline_against_single_process
    ranks = _run_gpu_ranks(2, "pipeline", tmp_path)
    check_pipeline_against_single_process(ranks, 2, dev=torch.device("cuda", 0))
