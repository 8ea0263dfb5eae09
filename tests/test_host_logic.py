"""Host-side pieces that need no GPU: the CSR container, the transposed CSR built for the
backward pass, loaders / batch tuples, the partition stand-in, and the staging pool's queue
discipline against the pure-Python model of pool.py:64-123."""
import numpy as np
import pytest
import torch

from helpers import random_csr
from oracle import cpu_path as cp
from oracle.pool_model import PoolModel


def test_sparse_tensor_surface():
    import torch_geometric_autoscale as tga
    row = torch.tensor([2, 0, 1, 0])
    col = torch.tensor([1, 2, 0, 0])
    val = torch.tensor([4.0, 2.0, 3.0, 1.0])
    a = tga.SparseTensor(row=row, col=col, value=val, sparse_sizes=(3, 3))
    rowptr, c, v = a.csr()
    assert rowptr.tolist() == [0, 2, 3, 4] and c.tolist() == [0, 2, 0, 1] and v.tolist() == [1, 2, 3, 4]
    assert a.storage.row().tolist() == [0, 0, 1, 2] and a.storage.rowcount().tolist() == [2, 1, 1]
    assert a.size(0) == 3 and a.sparse_sizes() == (3, 3) and a.nnz() == 4
    assert a.set_value(None).storage.value() is None and a.storage.value() is not None
    d = a.set_diag()
    assert d.nnz() == 6 and d.csr()[1].tolist() == [0, 2, 0, 1, 1, 2]
    p = a.permute(torch.tensor([2, 0, 1]))            # out[i, j] = a[perm[i], perm[j]]
    dense = torch.zeros(3, 3); dense[a.storage.row(), a.storage.col()] = a.storage.value()
    pd = torch.zeros(3, 3); pd[p.storage.row(), p.storage.col()] = p.storage.value()
    assert torch.equal(pd, dense[[2, 0, 1]][:, [2, 0, 1]])
    s = tga.SparseTensor(row=torch.tensor([0]), col=torch.tensor([1]), sparse_sizes=(2, 2)).to_symmetric()
    assert s.nnz() == 2
    m = a.masked_select_nnz(torch.tensor([True, False, True, True]))
    assert m.nnz() == 3 and m.sparse_sizes() == (3, 3)
    assert a.max_row_nnz() == 2


@pytest.mark.parametrize("reduce", ["sum", "mean"])
def test_csr_transpose_matches_oracle(reduce):
    from torch_geometric_autoscale import ops
    rng = np.random.default_rng(2)
    rowptr, col, val = random_csr(rng, 70, 90, 6, hubs={1: 80}, empty_rows=(0,))
    t_rowptr, t_col, t_val = ops.csr_transpose(torch.from_numpy(rowptr), torch.from_numpy(col),
                                               torch.from_numpy(val), 70, 90, reduce)
    assert t_rowptr.dtype == t_col.dtype == torch.int32
    dy = rng.standard_normal((70, 3))
    got = cp.spmm_f64(t_rowptr.numpy().astype(np.int64), t_col.numpy().astype(np.int64),
                      t_val.numpy(), dy.astype(np.float32))
    want = cp.spmm_t(rowptr, col, val, dy.astype(np.float32), 90, reduce)
    assert np.allclose(got, want, rtol=1e-5, atol=1e-5)
    for r in range(90):                                # source rows ascending in every row
        seg = t_col[t_rowptr[r]:t_rowptr[r + 1]].tolist()
        assert seg == sorted(seg)
    with pytest.raises(RuntimeError):
        ops.csr_transpose(torch.tensor([0, 1]), torch.tensor([5]), None, 1, 3, "sum")


def test_loaders_and_batch_tuples():
    import torch_geometric_autoscale as tga
    data, f_in, n_cls = tga.get_data(None, "arxiv", num_parts=6, scale=0.01, seed=3)
    data.adj_t = tga.gcn_norm(data.adj_t.set_diag())
    ptr = data.ptr
    ev = tga.EvalSubgraphLoader(data, ptr, batch_size=2, log=False)
    seen = []
    for sub in ev:
        data_b, b, n_id, offset, count = sub
        assert data_b.adj_t.sparse_sizes() == (b, n_id.numel())
        assert int(count.sum()) == b and torch.equal(n_id[:b], torch.arange(int(offset[0]), int(offset[0]) + b))
        assert torch.equal(data_b.x, data.x[n_id]) and torch.equal(data_b.y, data.y[n_id])
        assert len(set(n_id.tolist())) == n_id.numel()
        seen.append(n_id[:b])
    assert torch.equal(torch.cat(seen), torch.arange(data.num_nodes))
    tr = tga.SubgraphLoader(data, ptr, batch_size=3, shuffle=True, type='train', IB=True)
    for data_b, b, n_id, offset, count in tr:
        assert data_b.adj_t.sparse_sizes() == (b, b) and n_id.numel() == b and offset.numel() == 3
    ns_loader = tga.SubgraphLoader(data, ptr, batch_size=2, shuffle=False, type='train', IB=False,
                                   num_neighbors=3)
    for data_b, b, n_id, offset, count in ns_loader:
        assert int(data_b.adj_t.storage.rowcount().max()) <= 3


def test_partition_stand_in_and_permute():
    import torch_geometric_autoscale as tga
    adj, _ = tga.synthetic_graph(500, 4000, 5, seed=1)
    perm, ptr = tga.metis(adj, 4, log=False)
    assert sorted(perm.tolist()) == list(range(500)) and ptr[0] == 0 and ptr[-1] == 500
    assert ptr.numel() == 5 and int((ptr[1:] - ptr[:-1]).min()) > 0
    data = tga.Data(adj_t=adj, x=torch.arange(500.0).view(-1, 1), num_nodes=500)
    out = tga.permute(data, perm, log=False)
    assert torch.equal(out.x.view(-1), perm.to(torch.float32))
    r, c, _ = out.adj_t.coo()
    orig = set(zip(*[t.tolist() for t in adj.coo()[:2]]))
    assert all((int(perm[i]), int(perm[j])) in orig for i, j in zip(r[:200].tolist(), c[:200].tolist()))
    cl = torch.tensor([1, 0, 1, 0])
    perm2, ptr2 = tga.metis(tga.SparseTensor(rowptr=torch.tensor([0, 0, 0, 0, 0]), col=torch.zeros(0, dtype=torch.long),
                                             sparse_sizes=(4, 4)), 2, cluster=cl)
    assert perm2.tolist() == [1, 3, 0, 2] and ptr2.tolist() == [0, 2, 4]


def test_pool_queue_matches_reference_model(monkeypatch):
    """Same launches, same slots, same order as the model of pool.py -- with the copies and
    streams replaced by a recording backend (no GPU involved)."""
    from torch_geometric_autoscale import pool as pool_mod
    log = []

    class FakeStream:
        def wait_stream(self, other): pass
        def synchronize(self): pass

    p = pool_mod.AsyncIOPool(pool_size=2, buffer_size=8, embedding_dim=4)
    monkeypatch.setattr(p, "_start_pull", lambda req: log.append(("launch_pull", req.slot, req.src)))
    monkeypatch.setattr(p, "synchronize_pull", lambda: log.append(("sync_pull", p.pending[0].slot, p.pending[0].src)) or p.pending[0].slot)
    m = PoolModel(2)
    script = ["pull a", "pull b", "pull c", "sync", "free", "pull d", "sync", "free", "sync", "free",
              "pull e", "sync", "free", "sync", "free"]
    for cmd in script:
        if cmd.startswith("pull"):
            tag = cmd.split()[1]
            p.async_pull(tag, None, None, torch.zeros(0, dtype=torch.long)); m.async_pull(tag)
        elif cmd == "sync":
            assert p.synchronize_pull() == m.synchronize_pull()
        else:
            p.free_pull(); m.free_pull()
        assert [q.slot for q in p.pending] == [q[0] for q in m.queue]
        assert p.last_pull_slot == m.pull_index
    assert log == m.log


def test_history_interface_on_host():
    import torch_geometric_autoscale as tga
    h = tga.History(10, 3)
    assert h.emb.shape == (10, 3) and float(h.emb.abs().sum()) == 0 and not h.emb.is_cuda
    assert list(h.state_dict().keys()) == []                 # emb is not a buffer
    h.to(torch.float64)                                      # must not move / convert emb
    assert h.emb.dtype == torch.float32
    x = torch.arange(12.0).view(4, 3)
    h.push(x, torch.tensor([7, 1, 3, 9]))
    assert torch.equal(h.pull(torch.tensor([1, 9])), x[[1, 3]])
    h.push(x, torch.arange(4), torch.tensor([5, 0]), torch.tensor([3, 1]))
    assert torch.equal(h.emb[5:8], x[:3]) and torch.equal(h.emb[0], x[3])
    with pytest.raises(ValueError):
        h.push(x)
    h.push(torch.ones(10, 3))
    assert float(h.pull().sum()) == 30
    with pytest.raises(NotImplementedError):
        h(x)


def test_neighbor_sampling_and_edge_dropout():
    """Per-row neighbour sampling of the train loader (loader.py:32-93, :217-246) and edge
    dropout (utils.py:62-73)."""
    import torch_geometric_autoscale as tga
    from torch_geometric_autoscale.loader import sample_neighbors
    adj, ptr = tga.synthetic_graph(600, 600 * 12, 4, seed=2)
    rowptr, col, _ = adj.csr()
    torch.manual_seed(0)
    out, _ = sample_neighbors(adj, 5)
    rp2, col2, _ = out.csr()
    deg, deg2 = rowptr[1:] - rowptr[:-1], rp2[1:] - rp2[:-1]
    assert torch.equal(deg2, torch.clamp(deg, max=5))
    for r in (0, 17, 300, 599):   # survivors are a subset of the row, original order kept
        full = col[rowptr[r]:rowptr[r + 1]].tolist()
        kept = col2[rp2[r]:rp2[r + 1]].tolist()
        assert kept == sorted(kept) and set(kept) <= set(full)
    assert sample_neighbors(adj, -1)[0] is adj
    # the train loader with a neighbour budget: batches keep their node lists, rows are capped
    data = tga.Data(adj_t=adj, num_nodes=600, x=torch.randn(600, 4))
    loader = tga.SubgraphLoader(data, ptr, batch_size=2, shuffle=False, type='train',
                                num_neighbors=3, log=False)
    for sub in loader:
        rp, c, _ = sub.data.adj_t.csr()
        assert int((rp[1:] - rp[:-1]).max()) <= 3
        assert sub.data.x.size(0) == sub.n_id.numel() and int(c.max()) < sub.n_id.numel()
    # edge dropout: structure thinned without weights, weights dropped and rescaled with them
    torch.manual_seed(1)
    thin = tga.dropout(adj, 0.5)
    assert 0.3 * adj.nnz() < thin.nnz() < 0.7 * adj.nnz()
    assert tga.dropout(adj, 0.5, training=False) is adj and tga.dropout(adj, 0.0) is adj
    weighted = adj.fill_value(1.0)
    w = tga.dropout(weighted, 0.5).storage.value()
    assert w.numel() == adj.nnz() and set(w.unique().tolist()) == {0.0, 2.0}
    # an adjacency with dropped weights does not inherit value-carrying views of its source:
    # the in-batch view (models/gcn.py:117-126) of the dropped matrix holds the dropped weights
    from torch_geometric_autoscale.models.base import in_batch_columns
    first = in_batch_columns(weighted, 300)           # cached on the source
    dropped = tga.dropout(weighted, 0.5)
    view = in_batch_columns(dropped, 300)
    assert view is not first
    _, c, v = dropped.coo()[0], dropped.coo()[1], dropped.coo()[2]
    assert torch.equal(view.storage.value(), v[c < 300])
    assert set(first.storage.value().unique().tolist()) == {1.0}


def test_batched_gcn2_terms_and_fused_embed_have_the_plain_gradients():
    """The all-layers-at-once x_0 GEMM and the fused embed/relu backward are re-arrangements:
    outputs and gradients must equal the layer-by-layer composition."""
    from torch_geometric_autoscale.nn import GCN2Conv, RowsLinear, embed_relu, gcn2_batched_terms
    torch.manual_seed(0)
    rows, b, f_in, c, n_layers = 50, 30, 7, 16, 3
    lin = RowsLinear(f_in, c).double()
    convs = [GCN2Conv(c, alpha=0.1, theta=0.5, layer=i + 1, shared_weights=False).double()
             for i in range(n_layers)]
    x = torch.randn(rows, f_in, dtype=torch.float64)
    aggs = [torch.randn(b, c, dtype=torch.float64) for _ in range(n_layers)]
    params = list(lin.parameters()) + [p for conv in convs for p in conv.parameters()]

    def run(fast):
        if fast:
            full, x0 = embed_relu(lin, x, b)
            w1f, terms, dw1 = gcn2_batched_terms(convs, x0)
            outs = [conv.forward_after_propagate(a, x0, w1f[i], terms[i], dw1[i])
                    for i, (conv, a) in enumerate(zip(convs, aggs))]
        else:
            full = lin(x).relu()
            x0 = full[:b]
            outs = [conv.forward_after_propagate(a, x0) for conv, a in zip(convs, aggs)]
        loss = sum((o * o).sum() for o in outs) + (full * torch.arange(c, dtype=torch.float64)).sum()
        grads = torch.autograd.grad(loss, params)
        return outs, grads

    outs_f, grads_f = run(True)
    outs_p, grads_p = run(False)
    for a, b_ in zip(outs_f + list(grads_f), outs_p + list(grads_p)):
        assert torch.allclose(a, b_, rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("kind", ["gcn2", "gcn"])
@pytest.mark.parametrize("vr", [False, True])
def test_deferred_activation_gives_the_same_outputs_and_gradients(vr, kind):
    """Host orchestration of model_cases.check_deferred_activation (kernels swapped for the
    oracle); the GPU run of the same check is in test_gpu_models.py."""
    import model_cases as mc
    from cpu_shim import cpu_kernels
    with cpu_kernels():
        mc.check_deferred_activation("cpu", vr, kind)


def test_merge_row_chunks_joins_neighbours_and_keeps_the_row_order():
    from torch_geometric_autoscale import ops
    t = lambda *v: torch.tensor(v, dtype=torch.int64)
    cases = [
        ((t(0, 10, 30), t(10, 20, 5)), ([0], [35])),               # all adjoin
        ((t(0, 12, 30), t(10, 18, 5)), ([0, 12], [10, 23])),       # a gap in front of the second
        ((t(30, 0, 10), t(5, 10, 20)), ([30, 0], [5, 30])),        # table order != packed order
        ((t(0, 10, 10, 25), t(10, 0, 15, 5)), ([0], [30])),        # an empty chunk in between
        ((t(7), t(3)), ([7], [3])),
        ((t(), t()), ([], [])),
        ((t(4, 9), t(0, 0)), ([], [])),
    ]
    for (off, cnt), (want_o, want_c) in cases:
        o, c = ops.merge_row_chunks(off, cnt)
        assert o.dtype == torch.int64 and c.dtype == torch.int64
        assert o.tolist() == want_o and c.tolist() == want_c
        # the rows addressed, in order, are unchanged
        rows = lambda oo, cc: [r for a, n in zip(oo.tolist(), cc.tolist()) for r in range(a, a + n)]
        assert rows(o, c) == rows(off, cnt)
    with pytest.raises(RuntimeError, match="Size mismatch"):
        ops.merge_row_chunks(t(1, 2), t(1))
