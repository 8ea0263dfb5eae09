"""GPU parity of the device-side batch builder (relabel + transposed CSR), called through the C
ABI: bit-exact against the reference-made fixtures (tests/golden/relabel_golden.npz), against
the CPU oracle on partition batches of a larger graph, and -- at ogbn-products batch size --
against the host builder of the product (itself pinned by the same fixtures)."""
import numpy as np
import pytest
import torch

from oracle import cpu_path as cp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the GPU"
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def ops():
    from torch_geometric_autoscale import ops as _ops
    return _ops


def _square_rowptr(rp32, n_pad):
    rp = rp32.cpu().numpy().astype(np.int64)
    return np.concatenate([rp, np.full(n_pad, rp[-1], dtype=np.int64)])


def test_device_relabel_matches_reference_fixtures(ops, dev, golden_relabel):
    g = golden_relabel
    tags = sorted({k.split("/")[0] for k in g.files if k.startswith("c")})
    seen = 0
    for tag in tags:
        gname, iname, vname, bip, op = g[f"{tag}/meta"]
        if vname not in ("none", "f32"):
            continue                      # the device format carries fp32 values only
        rowptr = torch.from_numpy(g[f"graph/{gname}/rowptr"]).to(dev)
        col = torch.from_numpy(g[f"graph/{gname}/col"]).to(dev)
        val = None if vname == "none" else torch.from_numpy(g[f"graph/{gname}/value_{vname}"]).to(dev)
        idx = torch.from_numpy(g[f"{tag}/idx"]).to(dev)
        within = op != "one_hop"
        rp, c, v, nid, distinct = ops.relabel_one_hop_device(rowptr, col, val, idx, within)
        n_idx = idx.numel()
        want_rp = g[f"{tag}/out_rowptr"]
        if bip == "1":
            got_rp = rp.cpu().numpy().astype(np.int64)
        else:   # square form: the caller pads (relabel_cpu.cpp:98-106, 204-212)
            got_rp = _square_rowptr(rp, distinct if within else nid.numel() - n_idx)
        assert np.array_equal(got_rp, want_rp), tag
        assert np.array_equal(c.cpu().numpy().astype(np.int64), g[f"{tag}/out_col"]), tag
        assert np.array_equal(nid.cpu().numpy(), g[f"{tag}/out_n_id"]), tag
        if val is None:
            assert v is None
        else:
            assert np.array_equal(v.cpu().numpy(), g[f"{tag}/out_value"]), tag
        seen += 1
    assert seen >= 100
    # the work tables are all -1 again
    from torch_geometric_autoscale.ops import _build_ws
    for m in _build_ws[dev]["maps"]:
        assert int((m != -1).sum()) == 0


def _graph(n, deg, parts, seed):
    import torch_geometric_autoscale as tga
    adj, ptr = tga.synthetic_graph(n, n * deg, parts, seed=seed)
    rowptr, col, _ = adj.csr()
    val = torch.arange(col.numel(), dtype=torch.float32) * 0.5
    return rowptr, col, val, ptr


def test_device_relabel_matches_oracle_on_partition_batches(ops, dev):
    rowptr, col, val, ptr = _graph(60000, 16, 12, seed=1)
    d = [t.to(dev) for t in (rowptr, col, val)]
    for parts in ([0], [3, 7], [11, 2, 5]):
        idx = torch.cat([torch.arange(int(ptr[p]), int(ptr[p + 1])) for p in parts])
        for within, ofn in ((False, cp.relabel_one_hop), (True, cp.relabel_one_hop_within_batch)):
            rp, c, v, nid, _ = ops.relabel_one_hop_device(d[0], d[1], d[2], idx.to(dev), within)
            w_rp, w_c, w_v, w_nid = ofn(rowptr.numpy(), col.numpy(), val.numpy(), idx.numpy(), True)
            assert np.array_equal(rp.cpu().numpy(), w_rp)
            assert np.array_equal(c.cpu().numpy(), w_c)
            assert np.array_equal(v.cpu().numpy(), w_v)
            assert np.array_equal(nid.cpu().numpy(), w_nid)
    # duplicates in idx: the later position owns the id, rows are repeated
    idx = torch.tensor([5, 9, 5, 100, 9, 7], dtype=torch.int64)
    for within, ofn in ((False, cp.relabel_one_hop), (True, cp.relabel_one_hop_within_batch)):
        got = ops.relabel_one_hop_device(d[0], d[1], None, idx.to(dev), within)
        want = ofn(rowptr.numpy(), col.numpy(), None, idx.numpy(), True)
        assert np.array_equal(got[0].cpu().numpy(), want[0])
        assert np.array_equal(got[1].cpu().numpy(), want[1])
        assert np.array_equal(got[3].cpu().numpy(), want[3])
        assert got[4] == 4


def test_device_relabel_errors_and_edges(ops, dev):
    rowptr = torch.tensor([0, 1, 2], device=dev)
    col = torch.tensor([1, 0], device=dev)
    with pytest.raises(RuntimeError):
        ops.relabel_one_hop_device(rowptr, col, None, torch.tensor([2], device=dev))
    with pytest.raises(RuntimeError):
        ops.relabel_one_hop_device(rowptr, torch.tensor([5, 0], device=dev), None,
                                   torch.tensor([0], device=dev))
    with pytest.raises(RuntimeError):       # capacity too small for the batch
        ops.relabel_one_hop_device(rowptr, col, None, torch.tensor([0, 1], device=dev),
                                   edge_capacity=1)
    with pytest.raises(RuntimeError):       # ... also for the within-batch variant, whose work
        ops.relabel_one_hop_device(rowptr, col, None, torch.tensor([0, 1], device=dev),   # arrays
                                   within_batch=True, edge_capacity=1)     # hold every edge of the rows
    # after failed calls the tables are clean: a good call still gives the right answer
    rp, c, v, nid, _ = ops.relabel_one_hop_device(rowptr, col, None, torch.tensor([1], device=dev))
    assert rp.tolist() == [0, 1] and c.tolist() == [1] and nid.tolist() == [1, 0] and v is None
    rp, c, v, nid, _ = ops.relabel_one_hop_device(rowptr, col, None,
                                                  torch.zeros(0, dtype=torch.int64, device=dev))
    assert rp.tolist() == [0] and c.numel() == 0 and nid.numel() == 0
    # a batch of isolated nodes: no edges at all
    rp0 = torch.zeros(4, dtype=torch.int64, device=dev)
    rp, c, v, nid, _ = ops.relabel_one_hop_device(rp0, torch.zeros(0, dtype=torch.int64, device=dev),
                                                  None, torch.tensor([2, 0], device=dev))
    assert rp.tolist() == [0, 0, 0] and c.numel() == 0 and nid.tolist() == [2, 0]


@pytest.mark.parametrize("reduce", ["sum", "mean"])
def test_device_transpose_matches_host(ops, dev, reduce):
    from helpers import random_csr
    rng = np.random.default_rng(5)
    for n_rows, n_cols, avg, hubs in ((300, 700, 9, {5: 650}), (1, 1, 1, None), (50, 3, 2, None),
                                      (4000, 9000, 20, {7: 5000})):
        rowptr, col, val = random_csr(rng, n_rows, n_cols, avg, hubs=hubs or (), empty_rows=(0,))
        for v in (val, None):
            want = ops.csr_transpose(torch.from_numpy(rowptr), torch.from_numpy(col),
                                     None if v is None else torch.from_numpy(v), n_rows, n_cols, reduce)
            got = ops.csr_transpose_device(
                torch.from_numpy(rowptr.astype(np.int32)).to(dev),
                torch.from_numpy(col.astype(np.int32)).to(dev),
                None if v is None else torch.from_numpy(v).to(dev), n_rows, n_cols, reduce)
            for a, b in zip(got, want):
                assert torch.equal(a.cpu(), b)
    # no edges
    z = torch.zeros(4, dtype=torch.int32, device=dev)
    t_rp, t_c, t_v = ops.csr_transpose_device(z, torch.zeros(0, dtype=torch.int32, device=dev), None,
                                              3, 5, reduce)
    assert t_rp.tolist() == [0] * 6 and t_c.numel() == 0


def test_device_builder_full_size_equals_host_builder(ops, dev):
    """One ogbn-products-sized partition batch (B ~ 16k rows, ~0.9M edges) out of a 2.4M-node
    graph would need the whole graph; a 1/8-scale graph with the same batch shape is used."""
    import torch_geometric_autoscale as tga
    adj, ptr = tga.synthetic_graph(300000, 300000 * 50, 18, cross_fraction=0.15, seed=3)
    rowptr, col, _ = adj.csr()
    val = torch.rand(col.numel(), generator=torch.Generator().manual_seed(0))
    idx = torch.arange(int(ptr[4]), int(ptr[5]))
    ns = torch.ops.torch_geometric_autoscale
    for within, fn in ((False, ns.relabel_one_hop), (True, ns.relabel_one_hop_within_batch)):
        w_rp, w_c, w_v, w_nid = fn(rowptr, col, val, idx, True)
        rp, c, v, nid, _ = ops.relabel_one_hop_device(rowptr.to(dev), col.to(dev), val.to(dev),
                                                      idx.to(dev), within)
        assert torch.equal(rp.cpu().to(torch.int64), w_rp)
        assert torch.equal(c.cpu().to(torch.int64), w_c)
        assert torch.equal(v.cpu(), w_v)
        assert torch.equal(nid.cpu(), w_nid)
    # sortedness / permutation properties of the transposed CSR at this size
    t_rp, t_c, t_v = ops.csr_transpose_device(rp, c, v, rp.numel() - 1, nid.numel(), "sum")
    assert int(t_rp[-1]) == c.numel()
    counts = torch.bincount(c.to(torch.int64), minlength=nid.numel())
    assert torch.equal((t_rp[1:] - t_rp[:-1]).to(torch.int64), counts)
    assert abs(float(t_v.double().sum()) - float(v.double().sum())) < 1e-6 * c.numel()


def test_device_partition_batch_drives_the_model_like_the_host_batch(ops, dev):
    """A batch built on the GPU is interchangeable with the host-built one: same adjacency,
    same transposed CSR, same model output and gradients."""
    import torch_geometric_autoscale as tga
    from torch_geometric_autoscale.loader import device_partition_batch, partition_batch
    from torch_geometric_autoscale.models import GCN2
    adj, ptr = tga.synthetic_graph(20000, 20000 * 12, 8, cross_fraction=0.2, seed=5)
    adj = tga.gcn_norm(adj.set_diag())
    g = torch.Generator().manual_seed(1)
    data = tga.Data(adj_t=adj, num_nodes=20000, x=torch.randn(20000, 24, generator=g),
                    y=torch.randint(0, 5, (20000,), generator=g))
    for in_batch_only in (False, True):
        host = partition_batch(data, ptr, [2, 5], in_batch_only=in_batch_only)
        devb = device_partition_batch(data, ptr, [2, 5], dev, in_batch_only=in_batch_only)
        assert devb.batch_size == host.batch_size
        assert torch.equal(devb.n_id.cpu(), host.n_id)
        assert torch.equal(devb.offset, host.offset) and torch.equal(devb.count, host.count)
        assert torch.equal(devb.data.x, host.data.x) and torch.equal(devb.data.y, host.data.y)
        h_adj = host.data.adj_t.to(dev)
        for a, b in zip(devb.data.adj_t.device_csr(), h_adj.device_csr()):
            assert torch.equal(a, b)
        for a, b in zip(devb.data.adj_t.device_csr_t("sum")[:3], h_adj.device_csr_t("sum")[:3]):
            assert torch.equal(a, b)
        assert devb.data.adj_t.max_row_nnz() == h_adj.max_row_nnz()
        # the aggregation's work plan: built on the device for one, on the host for the other
        pd, ph = devb.data.adj_t.plan(), h_adj.plan()
        assert (pd.n_hub, pd.n_items, pd.n_multi) == (ph.n_hub, ph.n_items, ph.n_multi)
        assert torch.equal(pd.order, ph.order)
        assert torch.equal(pd.items[:pd.n_items], ph.items[:ph.n_items])

    torch.manual_seed(0)
    model = GCN2(20000, 24, 32, 5, 3, alpha=0.1, theta=0.5, shared_weights=False, dropout=0.0,
                 device=dev).to(dev)
    for h in model.histories:
        h.emb.normal_(0, 0.1)
    outs, grads = [], []
    host = partition_batch(data, ptr, [1, 6])
    devb = device_partition_batch(data, ptr, [1, 6], dev)
    for sub in (host, devb):
        model.zero_grad()
        out = model(sub.data.x.to(dev), sub.data.adj_t.to(dev), sub.batch_size, sub.n_id,
                    sub.offset, sub.count)["out"]
        torch.nn.functional.cross_entropy(out, sub.data.y[:sub.batch_size].to(dev)).backward()
        outs.append(out.detach())
        grads.append([p.grad.clone() for p in model.parameters() if p.grad is not None])
    assert torch.allclose(outs[0], outs[1], rtol=1e-5, atol=1e-6)
    for a, b in zip(*grads):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6)


import os  # noqa: E402
import subprocess  # noqa: E402
import sys  # noqa: E402

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isfile(os.path.join(_ROOT, "oracle", "_ref", "ref_relabel.so")),
                    reason="reference build not present in this tree")
def test_device_relabel_matches_live_reference(ops, dev, tmp_path):
    """The device batch builder against the reference's own relabel operators (compiled from
    the reference's sources by oracle/build_ref.py; the built object travels with the tree) on
    seeds no committed fixture holds.  The reference runs in a subprocess on the CPU: it
    registers the same operator names as the product."""
    import torch_geometric_autoscale as tga
    adj, ptr = tga.synthetic_graph(30000, 30000 * 14, 12, seed=23)
    rowptr, col, _ = adj.csr()
    val = torch.rand(col.numel(), generator=torch.Generator().manual_seed(5))
    idx = torch.cat([torch.arange(int(ptr[1]), int(ptr[2])), torch.arange(int(ptr[7]), int(ptr[9]))])
    inp, outp = tmp_path / "in.npz", tmp_path / "out.npz"
    np.savez(inp, rowptr=rowptr.numpy(), col=col.numpy(), val=val.numpy(), idx=idx.numpy())
    code = f"""
import sys, numpy as np, torch
sys.path.insert(0, {_ROOT!r})
from oracle import build_ref
one_hop, within = build_ref.load()
d = np.load({str(inp)!r})
t = lambda k: torch.from_numpy(d[k])
out = {{}}
for name, fn in (("one_hop", one_hop), ("within", within)):
    rp, c, v, nid = fn(t("rowptr"), t("col"), t("val"), t("idx"), True)
    for k, a in zip(("rp", "c", "v", "nid"), (rp, c, v, nid)):
        out[f"{{name}}_{{k}}"] = a.numpy()
np.savez({str(outp)!r}, **out)
"""
    subprocess.check_call([sys.executable, "-c", code])
    ref = np.load(outp)
    for name, within in (("one_hop", False), ("within", True)):
        rp, c, v, nid, _ = ops.relabel_one_hop_device(rowptr.to(dev), col.to(dev), val.to(dev),
                                                      idx.to(dev), within)
        assert np.array_equal(rp.cpu().numpy().astype(np.int64), ref[f"{name}_rp"]), name
        assert np.array_equal(c.cpu().numpy().astype(np.int64), ref[f"{name}_c"]), name
        assert np.array_equal(nid.cpu().numpy(), ref[f"{name}_nid"]), name
        assert np.array_equal(v.cpu().numpy(), ref[f"{name}_v"]), name
