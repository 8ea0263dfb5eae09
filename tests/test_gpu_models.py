"""GPU parity at model level: the four model families through the HIP path (History tables
in HBM, and in pinned host memory behind the staging pools) against the numpy full-graph
restatement.  Tolerance 2e-4 (fp32 chains of up to 4 layers; the task's bar is 1e-4 per
embedding, met by the per-kernel tests)."""
import numpy as np
import pytest
import torch

import model_cases as mc
from oracle import cpu_path as cp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


@pytest.mark.parametrize("name", list(mc.MODEL_CASES))
def test_model_paths_tables_in_hbm(name, dev):
    from torch_geometric_autoscale import _lib
    data, ptr, n_cls = mc.case_graph(name)
    model = mc.make_model(name, data, data.x.size(1), n_cls, emb_device=dev).to(dev)
    assert model.pool is None
    loader, want = mc.check_refresh_sweep(name, model, data, ptr)
    mc.check_push_pull_forward(name, model, data, ptr, want, dev)
    if name in mc.SMALL_CASES:
        mc.check_plain_sweep(name, model, data, ptr)
        mc.check_ablation_modes(name, model, data, ptr, dev)
    # whole-model gradients of both step kinds against float64 autograd (tables as the sweep
    # left them for the push/pull step, weights moved away from them for the incremental one)
    mc.check_step_gradients(name, model, data, ptr, dev, vr=False)
    mc.check_incagg_forward(name, model, data, ptr, dev)
    mc.check_step_gradients(name, model, data, ptr, dev, vr=True)
    # a batch that is ONE range of table rows: dense update, activation and push (push/pull
    # step) or the next layer's relu(x) - M_in (incremental step) come from one fused launch
    mc.check_step_gradients(name, model, data, ptr, dev, vr=False, parts=(2,))
    mc.check_step_gradients(name, model, data, ptr, dev, vr=True, parts=(2,))
    mc.check_training_step(name, model, data, ptr, dev, vr=False)
    mc.check_training_step(name, model, data, ptr, dev, vr=True)
    _lib.check_device_errors()


@pytest.mark.parametrize("name", ["gcn2", "sage", "appnp", "gcn"])
@pytest.mark.parametrize("pool_size", [1, 3])
def test_model_paths_tables_on_host_with_staging_pools(name, pool_size, dev):
    from torch_geometric_autoscale import _lib
    import torch_geometric_autoscale as tga
    data, ptr = mc.make_graph()
    loader = tga.EvalSubgraphLoader(data, ptr, batch_size=2, log=False)
    buffer_size = max(s.n_id.numel() for s in loader) * 2          # main.py:177
    model = mc.make_model(name, data, data.x.size(1), 5, emb_device=None, pool_size=pool_size,
                          buffer_size=buffer_size).to(dev)
    assert model.pool is not None and model.pool_ag is not None
    assert model.histories[0].emb.is_pinned() and not model.histories[0].emb.is_cuda
    _, want = mc.check_refresh_sweep(name, model, data, ptr)
    mc.check_push_pull_forward(name, model, data, ptr, want, dev)
    mc.check_incagg_forward(name, model, data, ptr, dev)
    mc.check_training_step(name, model, data, ptr, dev, vr=False)
    mc.check_training_step(name, model, data, ptr, dev, vr=True)
    assert len(model.pool.pending) == 0 and len(model.pool_ag.pending) == 0
    _lib.check_device_errors()


def test_staged_pull_mode_matches_zero_copy(dev):
    """The worker-thread path (host gather into the pinned bounce buffer, then a copy) and the
    zero-copy gather kernel deliver identical rows."""
    import torch_geometric_autoscale as tga
    data, ptr = mc.make_graph()
    loader = tga.EvalSubgraphLoader(data, ptr, batch_size=2, log=False)
    buffer_size = max(s.n_id.numel() for s in loader) * 2
    outs = []
    for zero_copy in (True, False):
        model = mc.make_model("gcn2", data, data.x.size(1), 5, emb_device=None, pool_size=2,
                              buffer_size=buffer_size).to(dev)
        model.pool.zero_copy = model.pool_ag.zero_copy = zero_copy
        _, want = mc.check_refresh_sweep("gcn2", model, data, ptr)
        outs.append([h.emb.clone() for h in model.histories])
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("name", ["gcn2", "sage", "appnp", "gcn", "gcn2_headline"])
def test_sampled_and_edge_dropped_batches_through_the_hip_step(name, dev):
    """Neighbour sampling (loader.py:217-246) and random edge dropout (utils.py:62-73, applied
    to the device adjacency at every step, main.py:68) in front of the HIP training step:
    output, loss and every weight gradient against the float64 step oracle on the adjacency
    that came out of the sampling / the dropout."""
    from torch_geometric_autoscale import _lib
    data, ptr, n_cls = mc.case_graph(name)
    model = mc.make_model(name, data, data.x.size(1), n_cls, emb_device=dev).to(dev)
    mc.check_refresh_sweep(name, model, data, ptr)
    torch.manual_seed(11)
    mc.check_step_gradients(name, model, data, ptr, dev, vr=False, num_neighbors=4)
    mc.check_step_gradients(name, model, data, ptr, dev, vr=False, edge_dropout=0.3)
    mc.check_step_gradients(name, model, data, ptr, dev, vr=True, edge_dropout=0.3)
    mc.check_step_gradients(name, model, data, ptr, dev, vr=False, num_neighbors=6,
                            edge_dropout=0.2)
    _lib.check_device_errors()


@pytest.mark.parametrize("built_on", ["host", "device"])
def test_weighted_edge_dropout_keeps_the_cached_structure(built_on, dev):
    """Dropout on a weighted adjacency only changes edge values (utils.py:66-68): the copy
    shares the work plans, the in-place column ids and the index arrays of the transposed CSR
    with its source, and its transposed VALUES -- carried over by the cached permutation --
    equal those of a transposition from scratch, element for element."""
    import torch_geometric_autoscale as tga
    from torch_geometric_autoscale import ops
    from torch_geometric_autoscale.loader import device_partition_batch, partition_batch
    from torch_geometric_autoscale.utils import dropout
    data, ptr = mc.make_graph()
    if built_on == "host":
        adj = partition_batch(data, ptr, [1, 4]).data.adj_t.to(dev)
    else:
        adj = device_partition_batch(data, ptr, [1, 4], dev).data.adj_t
    n_r, n_c = adj.sparse_sizes()
    t_rowptr, t_col, t_val = adj.device_csr_t("sum")
    plan, plan_t = adj.plan(), adj.plan_t("sum", 0, n_r)
    torch.manual_seed(3)
    for _ in range(2):
        dropped = dropout(adj, p=0.4)
        assert dropped is not adj and dropped.nnz() == adj.nnz()
        value = dropped.storage.value()
        assert float((value == 0).float().mean()) > 0.2
        d_rowptr, d_col, d_val = dropped.device_csr_t("sum")
        assert d_rowptr is t_rowptr and d_col is t_col
        assert dropped.plan().order is plan.order
        assert dropped.plan_t("sum", 0, n_r).order is plan_t.order
        rp32, c32, _ = dropped.device_csr()
        f_rowptr, f_col, f_val = ops.csr_transpose_device(rp32, c32, value, n_r, n_c, "sum")
        assert torch.equal(f_rowptr, d_rowptr) and torch.equal(f_col, d_col)
        assert torch.equal(f_val, d_val)
        # and the aggregation over the dropped copy, forward and backward, follows its values
        x = torch.randn(n_c, 24, device=dev, requires_grad=True)
        dy = torch.randn(n_r, 24, device=dev)
        tga.spmm(dropped, x).backward(dy)
        rp, c, v = (t.cpu().numpy() for t in dropped.csr())
        want = cp.spmm_t(rp, c, v, dy.cpu().numpy(), n_c)
        np.testing.assert_allclose(x.grad.cpu().numpy(), want, rtol=1e-4, atol=1e-4)
    # the source is untouched
    assert torch.equal(adj.device_csr_t("sum")[2], t_val)


@pytest.mark.parametrize("reduce", ["sum", "mean"])
def test_aggregation_gradients(reduce, dev):
    """dX of spmm and of the incremental aggregate against the oracle's A^T dY."""
    import torch_geometric_autoscale as tga
    from helpers import random_csr
    rng = np.random.default_rng(5)
    b, m, w = 300, 800, 64
    rowptr, col, val = random_csr(rng, b, m, 14, hubs={2: 700}, empty_rows=(5,),
                                  weighted=(reduce == "sum"))
    adj = tga.SparseTensor(rowptr=torch.from_numpy(rowptr), col=torch.from_numpy(col),
                           value=None if val is None else torch.from_numpy(val),
                           sparse_sizes=(b, m), is_sorted=True).to(dev)
    x = torch.randn(m, w, device=dev, requires_grad=True)
    add = torch.randn(b, w, device=dev, requires_grad=True)
    dy = torch.randn(b, w, device=dev)
    y = tga.spmm(adj, x, reduce, scale=0.7, add=add, add_scale=0.3)
    y.backward(dy)
    want = 0.7 * cp.spmm_t(rowptr, col, val, dy.cpu().numpy(), m, reduce)
    np.testing.assert_allclose(x.grad.cpu().numpy(), want, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(add.grad.cpu().numpy(), 0.3 * dy.cpu().numpy(), rtol=1e-6, atol=1e-6)

    rowptr, col, val = random_csr(rng, b, b, 14, weighted=(reduce == "sum"))
    adj = tga.SparseTensor(rowptr=torch.from_numpy(rowptr), col=torch.from_numpy(col),
                           value=None if val is None else torch.from_numpy(val),
                           sparse_sizes=(b, b), is_sorted=True).to(dev)
    x = torch.randn(b, w, device=dev, requires_grad=True)
    m_in, m_ag = torch.randn(b, w, device=dev), torch.randn(b, w, device=dev)
    for fused in (True, False):
        x.grad = None
        h = tga.incagg_aggregate(adj, x, m_in, m_ag, reduce, fused=fused)
        h.backward(dy)
        want = cp.spmm_t(rowptr, col, val, dy.cpu().numpy(), b, reduce)
        np.testing.assert_allclose(x.grad.cpu().numpy(), want, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("kind", ["gcn2", "gcn"])
@pytest.mark.parametrize("vr", [False, True])
def test_deferred_activation_gives_the_same_outputs_and_gradients(vr, kind, dev):
    """Push/pull step and incremental-aggregation step with the relu between two layers folded
    into the kernels that consume it, against the same steps with the relu as a pass of its
    own."""
    mc.check_deferred_activation(dev, vr, kind)


@pytest.mark.parametrize("vr", [False, True])
def test_appnp_shared_term_gradients(vr, dev, monkeypatch):
    """APPNP hands alpha * x_0 to its propagation steps as one pre-scaled matrix whose gradient is
    summed over the steps by one launch (and fuses bias + relu of its first linear layer into
    the GEMM): outputs and every gradient equal those of the plain formulation, in which every
    step scales x_0 itself and autograd adds the gradients up (models/appnp.py:84-89,122-130)."""
    from torch_geometric_autoscale.loader import partition_batch
    from torch_geometric_autoscale.models import APPNP, appnp as appnp_module
    data, ptr = mc.make_graph()
    results = []
    for plain in (False, True):
        if plain:
            monkeypatch.setattr(appnp_module, "scaled_fan_out", lambda x, scale, n: None)
            monkeypatch.setattr(appnp_module, "embed_relu",
                                lambda lin, x, n: (lin(x).relu(), None))
        torch.manual_seed(5)
        model = APPNP(data.num_nodes, data.x.size(1), 32, 5, 4, alpha=0.1, dropout=0.0,
                      device=dev).to(dev)
        gen = torch.Generator().manual_seed(6)
        for h in list(model.histories) + list(model.histories_ag):
            h.emb.copy_(torch.randn(h.emb.shape, generator=gen))
        model.train()
        sub = partition_batch(data, ptr, [0, 1, 3], in_batch_only=vr)
        b = sub.batch_size
        call = model.VR_call if vr else model
        out = call(sub.data.x.to(dev), sub.data.adj_t.to(dev), b, sub.n_id, sub.offset,
                   sub.count)["out"]
        torch.nn.functional.cross_entropy(out, sub.data.y[:b].to(dev)).backward()
        results.append((out.detach().clone(),
                        {k: p.grad.clone() for k, p in model.named_parameters()}))
    (o1, g1), (o2, g2) = results
    assert torch.allclose(o1, o2, rtol=1e-5, atol=1e-6)
    assert g1.keys() == g2.keys() and len(g1) == 4
    for k in g1:
        assert torch.allclose(g1[k], g2[k], rtol=1e-4, atol=1e-6), k


def test_sage_fused_update_gradients(dev):
    """GraphSAGE's update as two GEMMs into one matrix (nn._SageUpdate) against the two linear
    layers and an addition (models/graphsage.py:623-644): same outputs, same gradients."""
    from torch_geometric_autoscale.nn import SAGEConv
    torch.manual_seed(11)
    conv = SAGEConv(48, 33).to(dev)
    agg = torch.randn(5000, 48, device=dev, requires_grad=True)
    x = torch.randn(6000, 48, device=dev, requires_grad=True)
    dy = torch.randn(5000, 33, device=dev)
    got = conv.update(agg, x)
    got.backward(dy)
    grads = [t.grad.clone() for t in (agg, x, *conv.parameters())]
    for t in (agg, x, *conv.parameters()):
        t.grad = None
    want = conv.lin_l(agg) + conv.lin_r(x[:5000])
    want.backward(dy)
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-5)
    for g, t in zip(grads, (agg, x, *conv.parameters())):
        assert torch.allclose(g, t.grad, rtol=1e-4, atol=1e-4)


def test_graph_replayed_steps_train_like_eager_steps():
    """Training steps recorded into HIP graphs (one per batch) and replayed leave the model
    where the same steps issued eagerly leave it."""
    import copy
    import torch_geometric_autoscale as tga
    from torch_geometric_autoscale.graphs import StepGraphs
    from torch_geometric_autoscale.loader import partition_batch
    from torch_geometric_autoscale.models import GCN2
    dev = torch.device("cuda", 0)
    adj, ptr = tga.synthetic_graph(6000, 6000 * 14, 6, cross_fraction=0.2, seed=9)
    adj = tga.gcn_norm(adj.set_diag())
    g = torch.Generator().manual_seed(2)
    data = tga.Data(adj_t=adj, num_nodes=6000, x=torch.randn(6000, 20, generator=g),
                    y=torch.randint(0, 4, (6000,), generator=g))
    torch.manual_seed(1)
    base = GCN2(6000, 20, 64, 4, 3, alpha=0.1, theta=0.5, shared_weights=False, dropout=0.0,
                device=dev).to(dev)
    for h in base.histories:
        h.emb.normal_(0, 0.1)
    batches = []
    for p in range(4):
        sub = partition_batch(data, ptr, [p])
        batches.append((sub.data.x.to(dev), sub.data.adj_t.to(dev), sub.batch_size,
                        sub.n_id.to(dev), sub.offset, sub.count, sub.data.y[:sub.batch_size].to(dev)))

    tables0 = [h.emb.clone() for h in base.histories]

    def run(graphed):
        model = copy.deepcopy(base)
        # plain SGD: rounding-level differences stay rounding-level (Adam would turn them
        # into +-lr steps wherever a gradient is ~0)
        opt = torch.optim.SGD(model.parameters(), lr=0.05)

        def step(k):
            x, a, b, nid, off, cnt, y = batches[k]
            out = model(x, a, b, nid, off, cnt)["out"]
            opt.zero_grad(set_to_none=True)
            torch.nn.functional.cross_entropy(out, y).backward()
            opt.step()

        order = [0, 1, 2, 3, 2, 0, 1, 3]
        if not graphed:
            for k in order:
                step(k)
        else:
            # warm-up and recording run real steps too: put weights, tables and optimizer
            # statistics back (in place: the graphs hold their addresses) before replaying
            state = copy.deepcopy(model.state_dict())
            graphs = StepGraphs(dev)
            for k in range(4):
                graphs.warm(lambda: step(k), times=1)
            for k in range(4):
                graphs.capture(k, lambda: step(k))
            torch.cuda.synchronize()
            model.load_state_dict(state)
            for h, t in zip(model.histories, tables0):
                h.emb.copy_(t)
            for k in order:
                graphs.replay(k)
        torch.cuda.synchronize()
        return model

    eager, replayed = run(False), run(True)
    moved = False
    for (n1, p1), (_, p2), (_, p0) in zip(eager.named_parameters(), replayed.named_parameters(),
                                          base.named_parameters()):
        assert torch.allclose(p1, p2, rtol=1e-3, atol=1e-5), n1
        moved = moved or not torch.equal(p1, p0)
    assert moved
    for h1, h2 in zip(eager.histories, replayed.histories):
        assert torch.allclose(h1.emb, h2.emb, rtol=1e-3, atol=1e-5)
