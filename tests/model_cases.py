"""Shared model-level checks, run on the CPU (kernels swapped for the oracle: host
orchestration only) and on the GPU (real HIP kernels) by test_models_cpu.py /
test_gpu_models.py."""
import numpy as np
import torch

from oracle import grad_oracle as go
from oracle import model_oracle as mo

TOL = dict(rtol=1e-4, atol=1e-4)   # the fp32 bar per embedding, held over chains of up to 8 layers


def make_graph(n=1400, parts=6, f_in=24, n_cls=5, seed=0):
    import torch_geometric_autoscale as tga
    adj, ptr = tga.synthetic_graph(n, n * 10, parts, cross_fraction=0.3, seed=seed)
    adj = tga.gcn_norm(adj.set_diag())
    g = torch.Generator().manual_seed(seed + 1)
    data = tga.Data(adj_t=adj, num_nodes=n, x=torch.randn(n, f_in, generator=g),
                    y=torch.randint(0, n_cls, (n,), generator=g))
    return data, ptr


MODEL_CASES = {
    "gcn": dict(kind="gcn", cfg=dict(num_layers=3, hidden_channels=32, batch_norm=True)),
    "gcn_linear": dict(kind="gcn", cfg=dict(num_layers=2, hidden_channels=32, linear=True,
                                            residual=True)),
    "sage": dict(kind="sage", cfg=dict(num_layers=3, hidden_channels=32)),
    "gcn2_shared": dict(kind="gcn2", cfg=dict(num_layers=4, hidden_channels=32, alpha=0.1,
                                              theta=0.5, shared_weights=True, batch_norm=True)),
    "gcn2": dict(kind="gcn2", cfg=dict(num_layers=3, hidden_channels=32, alpha=0.1, theta=0.5,
                                       shared_weights=False, residual=True)),
    "appnp": dict(kind="appnp", cfg=dict(num_layers=4, hidden_channels=32, alpha=0.1)),
}

# The shape the headline metric is quoted on (GCNII, 8 layers x 256, a weight pair per layer, no
# dropout: conf/model/gcn2.yaml:51-71 widened to the 8 layers x 256 of BASELINE.json's headline configuration) on a 20k-node graph with the
# input width and class count of ogbn-products: the [B,256]x[256,8*256] x_0 GEMM, the 8-slab
# weight gradients and the side-by-side gradient blocks only exist at this depth and width.
HEADLINE_CASES = {
    "gcn2_headline": dict(kind="gcn2", cfg=dict(num_layers=8, hidden_channels=256, alpha=0.1,
                                                theta=0.5, shared_weights=False),
                          graph=dict(n=20000, parts=10, f_in=100, n_cls=47)),
}
MODEL_CASES.update(HEADLINE_CASES)
SMALL_CASES = [k for k in MODEL_CASES if k not in HEADLINE_CASES]


def case_graph(name):
    """``(data, ptr, n_cls)`` of a case: the shared 1400-node graph unless the case names its own."""
    g = MODEL_CASES[name].get("graph")
    if g is None:
        return make_graph() + (5,)
    data, ptr = make_graph(g["n"], g["parts"], g["f_in"], g["n_cls"])
    return data, ptr, g["n_cls"]


def make_model(name, data, f_in, n_cls, emb_device, pool_size=None, buffer_size=None,
               num_nodes=None):
    from torch_geometric_autoscale import models
    case = MODEL_CASES[name]
    cls = dict(gcn=models.GCN, sage=models.GraphSAGE, gcn2=models.GCN2, appnp=models.APPNP)[case["kind"]]
    torch.manual_seed(7)
    model = cls(num_nodes=data.num_nodes if num_nodes is None else num_nodes, in_channels=f_in, out_channels=n_cls,
                pool_size=pool_size, buffer_size=buffer_size, device=emb_device, **case["cfg"])
    # non-trivial batch-norm statistics so that eval-mode bn is not the identity
    for bn in getattr(model, "bns", []):
        with torch.no_grad():
            bn.running_mean.normal_(0, 0.1)
            bn.running_var.uniform_(0.5, 1.5)
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.normal_(0, 0.1)
    return model


def numpy_state(model):
    return {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()
            if not k.startswith(("reg_modules", "nonreg_modules"))}


def check_refresh_sweep(name, model, data, ptr, batch_parts=2):
    """After mini_inference_vr the tables hold the exact full-graph layer inputs and their
    aggregates, and the returned matrix is the full-batch output."""
    import torch_geometric_autoscale as tga
    case = MODEL_CASES[name]
    model.eval()
    loader = tga.EvalSubgraphLoader(data, ptr, batch_size=batch_parts, log=False)
    out = model.mini_inference_vr(loader=loader)
    if out.is_cuda:
        torch.cuda.synchronize()
    rowptr, col, val = (t.numpy() for t in data.adj_t.csr())
    inputs, aggs, want = mo.full_graph(case["kind"], numpy_state(model), case["cfg"], rowptr, col,
                                       val, data.x.numpy())
    width = model.histories[0].embedding_dim
    for i in range(model.num_layers):
        np.testing.assert_allclose(model.histories[i].emb.cpu().numpy(),
                                   mo.padded(inputs[i], width), **TOL,
                                   err_msg=f"{name}: M_in of layer {i}")
        np.testing.assert_allclose(model.histories_ag[i].emb.cpu().numpy(),
                                   mo.padded(aggs[i], width), **TOL,
                                   err_msg=f"{name}: M_ag of layer {i}")
    np.testing.assert_allclose(out.cpu().numpy(), want, **TOL, err_msg=f"{name}: output")
    # the sweep with out-of-batch rows gathered into a buffer first writes the same values
    first = out.clone()
    model.halo_in_place = False
    try:
        second = model.mini_inference_vr(loader=loader)
    finally:
        model.halo_in_place = True
    assert torch.allclose(first, second, rtol=1e-5, atol=1e-6), \
        f"{name}: gathered vs in-place sweep differ"
    return loader, want


def check_plain_sweep(name, model, data, ptr):
    """mini_inference (no aggregate tables) returns the same output and leaves
    histories[1:] identical to what the VR sweep wrote."""
    import torch_geometric_autoscale as tga
    model.eval()
    loader = tga.EvalSubgraphLoader(data, ptr, batch_size=1, log=False)
    before = [h.emb.clone() for h in model.histories]
    for h in model.histories[1:]:
        h.emb.zero_()
    out = model(loader=loader)
    for i in range(1, model.num_layers):
        np.testing.assert_allclose(model.histories[i].emb.cpu().numpy(),
                                   before[i].cpu().numpy(), **TOL)
    return out


def check_push_pull_forward(name, model, data, ptr, want, device):
    """With fresh tables the push/pull forward of a batch reproduces the full-batch rows."""
    from torch_geometric_autoscale.loader import partition_batch
    model.eval()
    for parts in ([1], [0, 3]):
        sub = partition_batch(data, ptr, parts)
        b = sub.batch_size
        ret = model(sub.data.x.to(device), sub.data.adj_t.to(device), b, sub.n_id, sub.offset,
                    sub.count)
        out = ret["out"]
        assert out.size(0) == b
        np.testing.assert_allclose(out.detach().cpu().numpy(), want[sub.n_id[:b].numpy()], **TOL,
                                   err_msg=f"{name}: push/pull forward, parts {parts}")
        # out-of-batch rows pulled into a buffer first, or read in place from the table by the
        # aggregation: same values through the same kernel
        model.halo_in_place = not model.halo_in_place
        try:
            again = model(sub.data.x.to(device), sub.data.adj_t.to(device), b, sub.n_id,
                          sub.offset, sub.count)["out"]
        finally:
            model.halo_in_place = not model.halo_in_place
        # (rows cut into slices are summed with atomics: equal up to fp32 summation order)
        assert torch.allclose(again, out, rtol=1e-5, atol=1e-6), \
            f"{name}: pulled vs in-place halo rows differ"


def check_ablation_modes(name, model, data, ptr, device):
    """The two ablation switches of the push/pull forward (models/gcn.py:115-141,167-193):
    ``aggregate_combined=False`` aggregates over in-batch edges only -- the model evaluated on
    the subgraph induced by the batch -- and ``use_aggregation=False`` replaces every
    aggregation by the identity (the layers degrade to an MLP over the batch rows)."""
    from torch_geometric_autoscale.loader import partition_batch
    case = MODEL_CASES[name]
    model.eval()
    sd = numpy_state(model)
    for parts in ([2], [5, 0]):
        sub = partition_batch(data, ptr, parts)
        b = sub.batch_size
        rowptr, col, val = (t.numpy() for t in sub.data.adj_t.csr())
        keep = col < b
        row = np.repeat(np.arange(b), np.diff(rowptr))
        ib_rowptr = np.zeros(b + 1, dtype=np.int64)
        np.cumsum(np.bincount(row[keep], minlength=b), out=ib_rowptr[1:])
        x_b = sub.data.x[:b].numpy()
        _, _, want_ib = mo.full_graph(case["kind"], sd, case["cfg"], ib_rowptr, col[keep],
                                      val[keep], x_b)
        eye = np.arange(b + 1, dtype=np.int64)
        _, _, want_mlp = mo.full_graph(case["kind"], sd, case["cfg"], eye, eye[:b],
                                       np.ones(b, dtype=np.float32), x_b)
        args = (sub.data.x.to(device), sub.data.adj_t.to(device), b, sub.n_id, sub.offset, sub.count)
        with torch.no_grad():
            got_ib = model(*args, aggregate_combined=False)["out"]
            got_mlp = model(*args, use_aggregation=False)["out"]
        assert got_ib.size(0) == b and got_mlp.size(0) == b
        np.testing.assert_allclose(got_ib.cpu().numpy(), want_ib, **TOL,
                                   err_msg=f"{name}: in-batch edges only, parts {parts}")
        np.testing.assert_allclose(got_mlp.cpu().numpy(), want_mlp, **TOL,
                                   err_msg=f"{name}: no aggregation, parts {parts}")


def check_incagg_forward(name, model, data, ptr, device):
    """Incremental-aggregation forward against the explicit formula with the table contents,
    after the weights moved away from the ones the tables were computed with."""
    from torch_geometric_autoscale.loader import partition_batch
    case = MODEL_CASES[name]
    model.eval()
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.05 * torch.randn(p.shape, generator=torch.Generator().manual_seed(3)).to(p.device))
    sd = numpy_state(model)
    t_in = [h.emb.cpu().numpy() for h in model.histories]
    t_ag = [h.emb.cpu().numpy() for h in model.histories_ag]
    for parts in ([2], [4, 1]):
        sub = partition_batch(data, ptr, parts, in_batch_only=True)
        b = sub.batch_size
        rp, c, v = (t.numpy() for t in sub.data.adj_t.csr())
        want = mo.vr_forward(case["kind"], sd, case["cfg"], rp, c, v, sub.data.x.numpy(), t_in, t_ag,
                             sub.n_id.numpy())
        for fused in (True, False):
            model.fused_incagg = fused
            ret = model.VR_call(sub.data.x.to(device), sub.data.adj_t.to(device), b, sub.n_id,
                                sub.offset, sub.count)
            np.testing.assert_allclose(ret["out"].detach().cpu().numpy(), want, **TOL,
                                       err_msg=f"{name}: VR forward fused={fused} parts {parts}")
        model.fused_incagg = False


def check_training_step(name, model, data, ptr, device, vr):
    """One optimisation step runs end to end and moves the weights; gradients are finite."""
    from torch_geometric_autoscale.loader import partition_batch
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    sub = partition_batch(data, ptr, [0, 1], in_batch_only=vr)
    b = sub.batch_size
    call = model.VR_call if vr else model
    out = call(sub.data.x.to(device), sub.data.adj_t.to(device), b, sub.n_id, sub.offset,
               sub.count)["out"]
    loss = torch.nn.functional.cross_entropy(out, sub.data.y[:b].to(device))
    opt.zero_grad()
    loss.backward()
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    assert grads and all(torch.isfinite(g).all() for g in grads)
    assert any(float(g.abs().sum()) > 0 for g in grads)
    opt.step()
    return float(loss.detach())


def check_step_gradients(name, model, data, ptr, device, vr, parts=(1, 3), num_neighbors=-1,
                         edge_dropout=0.0):
    """Loss, output and EVERY weight gradient of one training step (push/pull, or incremental
    aggregation with ``vr``) against float64 autograd over the layer formulas
    (oracle/grad_oracle.py) with the table contents the step reads; tolerance 1e-4 relative to
    the largest entry of each gradient.  Dropout off, batch norm (where the case has it) on batch
    statistics.

    ``num_neighbors`` / ``edge_dropout``: the batch adjacency is first sampled per row
    (loader.py:217-246) and / or put through random edge dropout (utils.py:62-73, main.py:68),
    on ``device``, exactly as the training loop does; the oracle then works on the adjacency
    that came out."""
    from torch_geometric_autoscale.loader import partition_batch, sample_neighbors
    from torch_geometric_autoscale.utils import dropout
    case = MODEL_CASES[name]
    model.train()
    sub = partition_batch(data, ptr, list(parts), in_batch_only=vr)
    b = sub.batch_size
    n_id = sub.n_id
    sd = numpy_state(model)
    adj = sub.data.adj_t.to(device)
    if num_neighbors >= 0:
        adj, n_id = sample_neighbors(adj, num_neighbors, n_id)
        assert adj.sparse_sizes() == sub.data.adj_t.sparse_sizes()
        assert int(adj.storage.rowcount().max()) <= num_neighbors
    if edge_dropout:
        before = adj
        adj = dropout(adj, p=edge_dropout)
        assert adj is not before and adj.nnz() <= before.nnz()
    rp, c, v = (None if t is None else t.cpu().numpy() for t in adj.csr())
    y = sub.data.y[:b]
    extra = {}
    if vr:
        extra["m_in"] = [h.emb[n_id[:b].to(h.emb.device)].cpu().numpy() for h in model.histories]
        extra["m_ag"] = [h.emb[n_id[:b].to(h.emb.device)].cpu().numpy() for h in model.histories_ag]
    else:
        extra["halo"] = [h.emb[n_id[b:].to(h.emb.device)].cpu().numpy() for h in model.histories]
    want_loss, want_out, want = go.step_gradients(case["kind"], sd, case["cfg"], rp, c, v,
                                                  sub.data.x.numpy(), b, y.numpy(), **extra)
    model.zero_grad(set_to_none=True)
    call = model.VR_call if vr else model
    out = call(sub.data.x.to(device), adj, b, n_id, sub.offset, sub.count)["out"]
    loss = torch.nn.functional.cross_entropy(out, y.to(device))
    loss.backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), want_out, **TOL,
                               err_msg=f"{name}: training-mode output, vr={vr}")
    assert abs(float(loss.detach()) - want_loss) <= 1e-4 * max(1.0, abs(want_loss))
    got = {k: p.grad.detach().cpu().numpy() for k, p in model.named_parameters()
           if p.grad is not None and not k.startswith(("reg_modules", "nonreg_modules"))}
    # (a gradient that is zero by construction -- a bias in front of batch norm -- comes out as
    # fp32 rounding noise: its allowance is tied to the largest gradient of the model)
    top = max(float(np.abs(g).max()) for g in want.values())
    # parameters the step does not reach have no gradient on either side (or a zero one)
    for k in sorted(set(want) | set(got)):
        g_want = want.get(k)
        g_got = got.get(k)
        if g_want is None or g_got is None:
            other = g_got if g_want is None else g_want
            assert not np.any(other), f"{name}: gradient of {k} on one side only"
            continue
        scale = float(np.abs(g_want).max())
        np.testing.assert_allclose(g_got, g_want, rtol=1e-4, atol=1e-4 * scale + 1e-6 * top,
                                   err_msg=f"{name}: gradient of {k}, vr={vr}")
    assert len(want) >= 3
    model.zero_grad(set_to_none=True)


def check_deferred_activation(device, vr, kind="gcn2"):
    """GCNII leaves the relu between two layers to its consumers (the next aggregation -- or the
    subtraction in front of the incremental aggregate -- reads max(0, x), the History push stores
    max(0, x), the aggregation's backward returns the gradient w.r.t. the pre-activations, the
    first aggregation's backward also masks the embedding's gradient): outputs, table contents
    and every weight gradient must equal the plain formulation's (models/gcn2.py:131-135 for the
    push/pull step, models/gcn2.py:246-262 for the incremental one)."""
    from torch_geometric_autoscale.loader import partition_batch
    from torch_geometric_autoscale.models import GCN, GCN2
    data, ptr = make_graph()
    device = torch.device(device)
    results = {}
    # ("shared" for GCN: the width of the hidden layers -- 32 makes the last layer, 32 -> 5,
    # aggregate a transformed copy of its input, where the relu cannot be folded in; with 4 no
    # layer does)
    for shared in (False, True):
        for defer in (True, False):
            torch.manual_seed(3)
            if kind == "gcn2":
                model = GCN2(data.num_nodes, data.x.size(1), 32, 5, 4, alpha=0.1, theta=0.5,
                             shared_weights=shared, dropout=0.0, device=device).to(device)
            else:
                model = GCN(data.num_nodes, data.x.size(1), 4 if shared else 32, 5, 3,
                            dropout=0.0, drop_input=False, batch_norm=False, residual=False,
                            device=device).to(device)
            gen = torch.Generator().manual_seed(4)
            for h in list(model.histories) + list(model.histories_ag):
                h.emb.copy_(torch.randn(h.emb.shape, generator=gen))
            if not defer:
                model._defers_relu = lambda: False
            assert model._defers_relu() == defer
            model.train()
            # (one partition: its table rows are read in place by the incremental step on the
            # GPU; two: they are copied out chunk by chunk)
            for parts in ([1, 4], [2]):
                sub = partition_batch(data, ptr, parts, in_batch_only=vr)
                b = sub.batch_size
                call = model.VR_call if vr else model
                model.zero_grad(set_to_none=True)
                out = call(sub.data.x.to(device), sub.data.adj_t.to(device), b, sub.n_id, sub.offset,
                           sub.count)["out"]
                torch.nn.functional.cross_entropy(out, sub.data.y[:b].to(device)).backward()
                results[(shared, defer, len(parts))] = (
                    out.detach().cpu().clone(), [h.emb.cpu().clone() for h in model.histories],
                    {k: p.grad.cpu().clone() for k, p in model.named_parameters()
                     if p.grad is not None})
    for key, (o1, t1, g1) in results.items():
        if not key[1]:
            continue
        o2, t2, g2 = results[(key[0], False, key[2])]
        assert torch.allclose(o1, o2, rtol=1e-5, atol=1e-6), key
        # (equal up to the summation order of the dense update: with the activation deferred
        # the update and the push of a single-range batch are one hand-written launch, without
        # it a library GEMM and a copy)
        for a, b_ in zip(t1, t2):
            assert torch.allclose(a, b_, rtol=1e-5, atol=1e-6), key
        assert g1.keys() == g2.keys() and len(g1) >= (6 if kind == "gcn2" else 5)
        for k in g1:
            assert torch.allclose(g1[k], g2[k], rtol=1e-4, atol=1e-6), (key, k)
