"""GPU parity: the HIP aggregation and History-row kernels, called through the C ABI,
against the CPU oracle on identical seeded inputs.

Tolerance for floating point: |gpu - fp64 oracle| <= 1e-4 + 1e-4 * |oracle| (the 1e-4 fp32
bar the task states), and the fp32 oracle itself must sit inside the same bound.
Copies (gather / scatter / chunk moves) must be bit-exact.
"""
import numpy as np
import pytest
import torch

from helpers import random_csr, tolerance
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


@pytest.fixture(params=["rows", "sliced", "grouped", "grouped_sorted", "auto"])
def variant(request):
    """Forces one aggregation kernel: whole rows per wave, feature slices with one row per wave,
    feature slices with 8 rows per wave (identity order / the degree-sorted order that the
    shape heuristic keys on), or the heuristic without an order."""
    from torch_geometric_autoscale import _lib
    _lib.set_tuning("spmm_sliced", {"rows": 0, "sliced": 1, "grouped": 2, "grouped_sorted": -1,
                                    "auto": -1}[request.param])
    yield request.param
    _lib.set_tuning("spmm_sliced", -1)


def _ro(variant, rowptr32):
    from torch_geometric_autoscale import ops as _ops
    return _ops.degree_order(rowptr32) if variant == "grouped_sorted" else None


def _dev_csr(rowptr, col, val, dev):
    return (torch.from_numpy(rowptr.astype(np.int32)).to(dev),
            torch.from_numpy(col.astype(np.int32)).to(dev),
            None if val is None else torch.from_numpy(val).to(dev))


def _check(y_gpu, y64, y32=None):
    y = y_gpu.cpu().numpy()
    tol = tolerance(y64)
    assert np.all(np.abs(y - y64) <= tol), f"max err {np.abs(y - y64).max()}"
    if y32 is not None:
        assert np.all(np.abs(y32 - y64) <= tol)


WIDTHS = [256, 128, 100, 64, 41, 47, 4, 1, 3, 260, 512, 1024]


@pytest.mark.parametrize("width", WIDTHS)
@pytest.mark.parametrize("reduce", ["sum", "mean"])
def test_spmm_matches_oracle(ops, dev, width, reduce, variant):
    from torch_geometric_autoscale import _lib
    rng = np.random.default_rng(width * 7 + (reduce == "mean"))
    n_rows, n_cols = 300, 700
    rowptr, col, val = random_csr(rng, n_rows, n_cols, 9, hubs={5: 650, 17: 700},
                                  empty_rows=(0, 11, 299), weighted=(reduce == "sum"))
    x = rng.standard_normal((n_cols, width)).astype(np.float32)
    y64 = cp.spmm_f64(rowptr, col, val, x, reduce)
    y32 = cp.spmm(rowptr, col, val, x, reduce)
    rp, c, v = _dev_csr(rowptr, col, val, dev)
    xd = torch.from_numpy(x).to(dev)
    for hint in (-1, int(np.diff(rowptr).max())):
        y = ops.spmm_csr(rp, c, v, xd, n_cols, reduce, max_row_nnz=hint, row_order=_ro(variant, rp))
        _check(y, y64, y32)
    _lib.check_device_errors()


def test_spmm_epilogue_and_strides(ops, dev, variant):
    rng = np.random.default_rng(3)
    n_rows, n_cols, width = 257, 400, 256
    rowptr, col, val = random_csr(rng, n_rows, n_cols, 30)
    xfull = rng.standard_normal((n_cols, 2 * width)).astype(np.float32)
    add = rng.standard_normal((n_rows + 5, width)).astype(np.float32)
    x = np.ascontiguousarray(xfull[:, :width])
    y64 = 0.9 * cp.spmm_f64(rowptr, col, val, x) + 0.1 * add[:n_rows].astype(np.float64)
    rp, c, v = _dev_csr(rowptr, col, val, dev)
    xd = torch.from_numpy(xfull).to(dev)[:, :width]          # ld = 2 * width
    y = ops.spmm_csr(rp, c, v, xd, n_cols, "sum", 0.9, torch.from_numpy(add).to(dev), 0.1,
                     row_order=_ro(variant, rp))
    _check(y, y64)


def test_spmm_hub_rows_sliced(ops, dev, variant):
    """Rows far beyond the slicing threshold, incl. one that covers every column twice over."""
    rng = np.random.default_rng(11)
    n_rows, n_cols, width = 64, 6000, 256
    rowptr, col, val = random_csr(rng, n_rows, n_cols, 20, hubs={0: 513, 1: 6000, 7: 3000, 63: 1025})
    x = rng.standard_normal((n_cols, width)).astype(np.float32)
    for reduce in ("sum", "mean"):
        v_in = val if reduce == "sum" else None
        y64 = cp.spmm_f64(rowptr, col, v_in, x, reduce)
        rp, c, v = _dev_csr(rowptr, col, v_in, dev)
        y = ops.spmm_csr(rp, c, v, torch.from_numpy(x).to(dev), n_cols, reduce,
                         row_order=_ro(variant, rp))
        tol = 1e-4 + 1e-4 * np.abs(y64) + 1e-6 * np.sqrt(6000)  # long fp32 sums
        assert np.all(np.abs(y.cpu().numpy() - y64) <= tol)
    # called twice in a row: the slice queue must be clean for the second call
    y2 = ops.spmm_csr(rp, c, v, torch.from_numpy(x).to(dev), n_cols, "mean",
                      row_order=_ro(variant, rp))
    assert torch.allclose(y, y2, rtol=1e-5, atol=1e-5)


def test_spmm_empty_and_degenerate(ops, dev):
    z = torch.zeros(1, dtype=torch.int32, device=dev)
    e = torch.zeros(0, dtype=torch.int32, device=dev)
    y = ops.spmm_csr(z, e, None, torch.zeros(3, 8, device=dev), 3, "sum")
    assert y.shape == (0, 8)
    rp = torch.zeros(5, dtype=torch.int32, device=dev)          # 4 rows, no edges
    y = ops.spmm_csr(rp, e, None, torch.ones(3, 8, device=dev), 3, "mean")
    assert y.shape == (4, 8) and float(y.abs().sum()) == 0.0


def test_spmm_bad_column_is_skipped_and_reported(ops, dev):
    from torch_geometric_autoscale import _lib
    rp = torch.tensor([0, 2], dtype=torch.int32, device=dev)
    c = torch.tensor([1, 99], dtype=torch.int32, device=dev)
    y = ops.spmm_csr(rp, c, None, torch.ones(3, 4, device=dev), 3, "sum")
    assert torch.equal(y.cpu(), torch.ones(1, 4))
    with pytest.raises(RuntimeError):
        _lib.check_device_errors()
    _lib.check_device_errors()  # latch cleared


def test_spmm_row_order_is_validated(ops, dev):
    """A row_order entry outside [0, n_rows) is not followed: it is reported, and the call's
    other rows are still written."""
    from torch_geometric_autoscale import _lib
    rp = torch.tensor([0, 1, 2, 3], dtype=torch.int32, device=dev)
    c = torch.tensor([0, 1, 2], dtype=torch.int32, device=dev)
    x = torch.arange(12, dtype=torch.float32, device=dev).view(3, 4)
    with pytest.raises(ValueError):
        ops.spmm_csr(rp, c, None, x, 3, row_order=torch.tensor([0, 1], dtype=torch.int32, device=dev))
    bad = torch.tensor([2, 7, 0], dtype=torch.int32, device=dev)
    y = ops.spmm_csr(rp, c, None, x, 3, row_order=bad)
    assert torch.equal(y[[0, 2]].cpu(), x[[0, 2]].cpu())
    with pytest.raises(RuntimeError):
        _lib.check_device_errors()
    good = torch.tensor([2, 1, 0], dtype=torch.int32, device=dev)
    assert torch.equal(ops.spmm_csr(rp, c, None, x, 3, row_order=good).cpu(), x.cpu())
    _lib.check_device_errors()


@pytest.mark.parametrize("width", [256, 128, 41, 512])
@pytest.mark.parametrize("reduce", ["sum", "mean"])
def test_incagg_matches_oracle(ops, dev, width, reduce, variant):
    rng = np.random.default_rng(width + 1000)
    b = 500
    rowptr, col, val = random_csr(rng, b, b, 12, hubs={3: 500, 4: 499}, empty_rows=(9,),
                                  weighted=(reduce == "sum"))
    x = rng.standard_normal((b, width)).astype(np.float32)
    table_w = max(width, 256)
    m_in = rng.standard_normal((b + 7, table_w)).astype(np.float32)
    m_ag = rng.standard_normal((b + 7, table_w)).astype(np.float32)
    mi, ma = np.ascontiguousarray(m_in[:b, :width]), np.ascontiguousarray(m_ag[:b, :width])
    h32 = cp.incagg(rowptr, col, val, x, mi, ma, reduce)
    h64 = cp.spmm_f64(rowptr, col, val, (x.astype(np.float64) - mi).astype(np.float32), reduce) + ma
    rp, c, v = _dev_csr(rowptr, col, val, dev)
    args = (rp, c, v, torch.from_numpy(x).to(dev), torch.from_numpy(m_in).to(dev)[:b, :width],
            torch.from_numpy(m_ag).to(dev)[:b, :width], reduce)
    _check(ops.spmm_csr_incagg(*args, row_order=_ro(variant, rp)), h64, h32)
    # with the epilogue: scale * (agg + m_ag) + add_scale * add
    add = rng.standard_normal((b, width)).astype(np.float32)
    h = ops.spmm_csr_incagg(*args, 0.9, torch.from_numpy(add).to(dev), 0.1,
                            row_order=_ro(variant, rp))
    _check(h, 0.9 * h64 + 0.1 * add)
    # the two-pass form agrees
    d = ops.rows_sub(args[3], args[4])
    assert torch.equal(d.cpu(), torch.from_numpy(x - mi))


@pytest.mark.parametrize("width", [256, 41, 100, 1024, 6])
def test_gather_scatter_chunks_bit_exact(ops, dev, width):
    rng = np.random.default_rng(width)
    n, k = 5000, 1237
    emb = rng.standard_normal((n, width)).astype(np.float32)
    n_id = rng.permutation(n)[:k].astype(np.int64)
    embd = torch.from_numpy(emb).to(dev)
    out = ops.gather_rows(embd, torch.from_numpy(n_id).to(dev))
    assert np.array_equal(out.cpu().numpy(), cp.history_pull(emb, n_id))

    x = rng.standard_normal((k, width)).astype(np.float32)
    ref = emb.copy()
    cp.history_push(ref, x, n_id)
    ops.scatter_rows(embd, torch.from_numpy(n_id).to(dev), torch.from_numpy(x).to(dev))
    assert np.array_equal(embd.cpu().numpy(), ref)

    offset = np.array([4000, 10, 2500, 0, 777], dtype=np.int64)
    count = np.array([321, 100, 0, 10, 1], dtype=np.int64)
    xs = rng.standard_normal((int(count.sum()) + 3, width)).astype(np.float32)
    cp.history_push(ref, xs, n_id, offset, count)
    ops.copy_row_chunks(embd, torch.from_numpy(xs).to(dev), torch.from_numpy(offset),
                        torch.from_numpy(count), packed_is_src=True)
    assert np.array_equal(embd.cpu().numpy(), ref)

    pulled = torch.full((int(count.sum()) + 2, width), 7.0, device=dev)
    ops.copy_row_chunks(pulled, embd, torch.from_numpy(offset), torch.from_numpy(count),
                        packed_is_src=False)
    want = np.full((int(count.sum()) + 2, width), 7.0, dtype=np.float32)
    cp.read_async(ref, offset, count, np.zeros(0, dtype=np.int64), want)
    assert np.array_equal(pulled.cpu().numpy(), want)


def test_chunk_copy_pads_with_zeros_and_many_chunks(ops, dev):
    rng = np.random.default_rng(5)
    n, w_src, w_dst = 3000, 100, 256
    table = torch.full((n, w_dst), 3.0, device=dev)
    n_chunks = 75                                     # more than one kernel-argument table
    offset = (np.arange(n_chunks) * 40).astype(np.int64)
    count = rng.integers(0, 30, size=n_chunks).astype(np.int64)
    x = rng.standard_normal((int(count.sum()), w_src)).astype(np.float32)
    ops.copy_row_chunks(table, torch.from_numpy(x).to(dev), torch.from_numpy(offset),
                        torch.from_numpy(count), packed_is_src=True)
    ref = np.full((n, w_dst), 3.0, dtype=np.float32)
    s = 0
    for o, c in zip(offset, count):
        ref[o:o + c, :w_src] = x[s:s + c]
        ref[o:o + c, w_src:] = 0
        s += c
    assert np.array_equal(table.cpu().numpy(), ref)


def test_chunk_copy_rejects_bad_chunks(ops, dev):
    table = torch.zeros(10, 8, device=dev)
    x = torch.zeros(4, 8, device=dev)
    with pytest.raises(RuntimeError, match="Invalid index"):
        ops.copy_row_chunks(table, x, torch.tensor([8]), torch.tensor([3]), packed_is_src=True)
    with pytest.raises(RuntimeError, match="Invalid index"):
        ops.copy_row_chunks(table, x, torch.tensor([0]), torch.tensor([5]), packed_is_src=True)


def test_gather_bad_index_is_skipped_and_reported(ops, dev):
    from torch_geometric_autoscale import _lib
    emb = torch.ones(10, 4, device=dev)
    out = torch.zeros(2, 4, device=dev)
    ops.gather_rows(emb, torch.tensor([3, 10], device=dev), out=out)
    assert torch.equal(out.cpu(), torch.tensor([[1.0] * 4, [0.0] * 4]))
    with pytest.raises(RuntimeError):
        _lib.check_device_errors()


def test_gather_from_pinned_host_table(ops, dev):
    rng = np.random.default_rng(9)
    emb = torch.from_numpy(rng.standard_normal((4000, 256)).astype(np.float32)).pin_memory()
    n_id = torch.from_numpy(rng.permutation(4000)[:999].astype(np.int64))
    out = ops.gather_rows(emb, n_id.to(dev))
    assert torch.equal(out.cpu(), emb[n_id])
    x = torch.from_numpy(rng.standard_normal((999, 256)).astype(np.float32))
    ops.scatter_rows(emb, n_id.to(dev), x.to(dev))
    torch.cuda.synchronize()
    assert torch.equal(emb[n_id], x)


def test_full_size_properties(ops, dev, variant):
    """ogbn-products-like partition batch (B = 16k rows, ~0.84M edges, F = 256): properties
    that need no CPU oracle -- linearity in x, A @ 1 = row sums, mean of a constant."""
    rng = np.random.default_rng(2026)
    b, m, width = 16384, 60000, 256
    rowptr, col, val = random_csr(rng, b, m, 51, hubs={100: 9000, 2000: 17000})
    rp, c, v = _dev_csr(rowptr, col, val, dev)
    g = torch.Generator(device=dev).manual_seed(0)
    x1 = torch.randn(m, width, device=dev, generator=g)
    x2 = torch.randn(m, width, device=dev, generator=g)
    ro = _ro(variant, rp)
    y1 = ops.spmm_csr(rp, c, v, x1, m, row_order=ro)
    y2 = ops.spmm_csr(rp, c, v, x2, m, row_order=ro)
    y12 = ops.spmm_csr(rp, c, v, x1 + 2 * x2, m, row_order=ro)
    err = (y12 - (y1 + 2 * y2)).abs().max().item()
    assert err <= 2e-3, err                      # hub rows sum 17k terms in fp32
    ones = torch.ones(m, width, device=dev)
    rs = ops.spmm_csr(rp, c, v, ones, m, row_order=ro)
    want = np.add.reduceat(np.concatenate([val, [0]]).astype(np.float64),
                           np.minimum(rowptr[:-1], val.size))
    want[np.diff(rowptr) == 0] = 0
    assert np.allclose(rs[:, 0].cpu().numpy(), want, atol=2e-3)
    mean1 = ops.spmm_csr(rp, c, None, ones * 3, m, "mean", row_order=ro)
    nonempty = torch.from_numpy(np.diff(rowptr) > 0).to(dev)
    assert torch.allclose(mean1[nonempty], torch.full_like(mean1[nonempty], 3.0), atol=1e-4)
    # gather then scatter back is the identity on a History-sized table
    table = torch.randn(200000, width, device=dev, generator=g)
    n_id = torch.randperm(200000, device=dev)[:m]
    rows = ops.gather_rows(table, n_id)
    t2 = table.clone()
    ops.scatter_rows(t2, n_id, rows)
    assert torch.equal(t2, table)
