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


@pytest.fixture(params=["rows", "rows_plan", "sliced", "sliced_plan", "sliced_short_segments",
                        "sliced_short_segments_zeroed", "rows_short_segments",
                        "two_slices_plan", "two_slices_short_segments"])
def variant(request):
    """Forces one aggregation kernel (whole rows per wave / 128-byte feature slices) and one
    kind of work plan: none (rows in storage order, every row by one wave or lane group), the
    default plan (long rows as whole-workgroup items), or a plan with a low threshold and
    128-edge segments, which sends many rows through the zero + atomic-combine path.
    Rows of 33..48 floats on aligned pitches go to the one-pass narrow-row kernel in the
    "sliced" variants and stay with the feature slices in the "two_slices" ones."""
    from torch_geometric_autoscale import _lib
    _lib.set_tuning("spmm_variant", 0 if request.param.startswith("rows") else 1)
    _lib.set_tuning("narrow_max_width", 32 if request.param.startswith("two_slices") else 48)
    yield request.param
    _lib.set_tuning("spmm_variant", -1)
    _lib.set_tuning("narrow_max_width", 48)


def _plan(variant, rowptr32):
    from torch_geometric_autoscale import ops as _ops
    if variant.endswith("_plan"):
        return _ops.SpmmPlan.build(rowptr32)
    if variant.endswith("_short_segments"):
        return _ops.SpmmPlan.build(rowptr32, hub_threshold=24, segment_edges=128)
    if variant.endswith("_short_segments_zeroed"):   # no combine buffer: zeroing launch in front
        return _ops.SpmmPlan.build(rowptr32, hub_threshold=24, segment_edges=128).without_combine()
    return None


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


WIDTHS = [256, 128, 100, 64, 41, 47, 36, 48, 4, 1, 3, 260, 512, 1024]


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
    y = ops.spmm_csr(rp, c, v, xd, n_cols, reduce, plan=_plan(variant, rp))
    _check(y, y64, y32)
    _lib.check_device_errors()


@pytest.mark.parametrize("width", [256, 41])
@pytest.mark.parametrize("weighted", [True, False])
def test_spmm_packed_edge_list(ops, dev, width, weighted):
    """A plan carrying the packed (col, val) pairs: the feature-sliced kernel reads its edges
    from there -- same results, bit for bit where no atomics are involved, against the oracle
    too; a packed list of the wrong length is refused."""
    rng = np.random.default_rng(77 + width)
    n_rows, n_cols = 300, 900
    rowptr, col, val = random_csr(rng, n_rows, n_cols, 9, hubs={5: 650, 17: 880},
                                  empty_rows=(0, 299), weighted=weighted)
    x = rng.standard_normal((n_cols, width)).astype(np.float32)
    rp, c, v = _dev_csr(rowptr, col, val, dev)
    _, xd = _pitched(x, dev, (width + 15) // 16 * 16 + 32, fill=0.0)
    plan = ops.SpmmPlan.build(rp)
    packed = ops.pack_edges(c, v)
    assert packed.shape == (col.size, 2) and packed.dtype == torch.int32
    y0 = ops.spmm_csr(rp, c, v, xd, n_cols, "sum", plan=plan)
    # poisoned col / val prove that the packed list is what the kernel reads
    y1 = ops.spmm_csr(rp, torch.zeros_like(c), None if v is None else torch.zeros_like(v), xd,
                      n_cols, "sum", plan=plan.with_edges(packed))
    assert torch.equal(y0, y1)
    _check(y1, cp.spmm_f64(rowptr, col, val, x, "sum"))
    m_in = rng.standard_normal((n_rows, width)).astype(np.float32)
    m_ag = rng.standard_normal((n_rows, width)).astype(np.float32)
    rowptr2, col2, val2 = random_csr(rng, n_rows, n_rows, 9, hubs={5: 290}, weighted=weighted)
    rp2, c2, v2 = _dev_csr(rowptr2, col2, val2, dev)
    plan2 = ops.SpmmPlan.build(rp2)
    (_, mind), (_, magd) = _pitched(m_in, dev, 1056, 0.0), _pitched(m_ag, dev, 1056, 0.0)
    h0 = ops.spmm_csr_incagg(rp2, c2, v2, xd, mind, magd, "mean", plan=plan2)
    h1 = ops.spmm_csr_incagg(rp2, c2, v2, xd, mind, magd, "mean",
                             plan=plan2.with_edges(ops.pack_edges(c2, v2)))
    assert torch.equal(h0, h1)
    with pytest.raises(ValueError):
        ops.spmm_csr(rp, c, v, xd, n_cols, "sum", plan=plan.with_edges(packed[:-1].contiguous()))
    from torch_geometric_autoscale import _lib
    _lib.check_device_errors()


def _pitched(a, dev, pitch, fill=float("nan")):
    """``a`` on the device as a [rows, width] view of a [rows, pitch] matrix whose other
    columns hold ``fill``."""
    buf = torch.full((a.shape[0], pitch), fill, dtype=torch.float32, device=dev)
    buf[:, :a.shape[1]] = torch.from_numpy(a).to(dev)
    return buf, buf[:, :a.shape[1]]


@pytest.mark.parametrize("width", [41, 47, 3, 6, 9, 33, 35, 37, 40, 44, 45, 48, 129, 1023])
def test_spmm_odd_widths_on_aligned_rows(ops, dev, width, variant):
    """Widths that are not a multiple of 4 on rows that start 16-byte aligned (the pitch the
    package allocates, ops.row_pitch): the feature-sliced kernel moves a row's last, partial
    piece float by float -- NaNs parked in the padding of every operand never reach a result,
    the padding of the output is not written, with every epilogue, rows cut into several
    segments, and the incremental aggregate."""
    rng = np.random.default_rng(1000 + width)
    n_rows, n_cols = 200, 200
    pitch = (width + 15) // 16 * 16 + 16
    rowptr, col, val = random_csr(rng, n_rows, n_cols, 7, hubs={3: 190, 100: 140},
                                  empty_rows=(0, 199))
    x = rng.standard_normal((n_cols, width)).astype(np.float32)
    add = rng.standard_normal((n_rows, width)).astype(np.float32)
    gate = rng.standard_normal((n_rows, width)).astype(np.float32)
    rp, c, v = _dev_csr(rowptr, col, val, dev)
    plan = _plan(variant, rp)
    (_, xd), (_, addd), (_, gated) = (_pitched(t, dev, pitch) for t in (x, add, gate))
    ybuf = torch.full((n_rows, pitch), 7.0, device=dev)
    y = ops.spmm_csr(rp, c, v, xd, n_cols, "sum", 0.5, addd, 2.0, out=ybuf[:, :width], plan=plan,
                     relu_x=True, gate=gated)
    want = (0.5 * cp.spmm_f64(rowptr, col, val, np.maximum(x, 0), "sum") + 2.0 * add) * (gate > 0)
    _check(y, want)
    assert bool((ybuf[:, width:] == 7.0).all())
    # split source + mean
    xb, xh = x[:120], x[120:]
    (_, xbd), (_, xhd) = _pitched(xb, dev, pitch), _pitched(xh, dev, pitch + 16)
    y = ops.spmm_csr(rp, c, None, xbd, n_cols, "mean", x_halo=xhd, out=ybuf[:, :width], plan=plan)
    _check(y, cp.spmm_f64(rowptr, col, None, x, "mean"))
    assert bool((ybuf[:, width:] == 7.0).all())
    # incremental aggregate on the square part
    rowptr2, col2, val2 = random_csr(rng, n_rows, n_rows, 7, hubs={3: 190, 100: 140})
    m_in = rng.standard_normal((n_rows, width)).astype(np.float32)
    m_ag = rng.standard_normal((n_rows, width)).astype(np.float32)
    rp2, c2, v2 = _dev_csr(rowptr2, col2, val2, dev)
    (_, mind), (_, magd) = _pitched(m_in, dev, pitch), _pitched(m_ag, dev, pitch)
    h = ops.spmm_csr_incagg(rp2, c2, v2, xd, mind, magd, "sum", out=ybuf[:, :width],
                            plan=_plan(variant, rp2))
    _check(h, cp.spmm_f64(rowptr2, col2, val2, (x[:n_rows].astype(np.float64) - m_in)
                          .astype(np.float32), "sum") + m_ag)
    assert bool((ybuf[:, width:] == 7.0).all())
    from torch_geometric_autoscale import _lib
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
                     plan=_plan(variant, rp))
    _check(y, y64)


@pytest.mark.parametrize("width", [256, 41])
def test_spmm_second_epilogue_term(ops, dev, width, variant):
    """y = scale * A x + add_scale * add + add2_scale * add2, then the gate: short rows, rows
    of one segment and rows cut into several (the terms ride with the first segment only), on
    padded row pitches; add2 alone; a mis-shaped add2 is refused."""
    rng = np.random.default_rng(31 + width)
    n_rows, n_cols = 300, 9000
    rowptr, col, val = random_csr(rng, n_rows, n_cols, 12, hubs={3: 700, 9: 8800},
                                  empty_rows=(0, 299))
    x = rng.standard_normal((n_cols, width)).astype(np.float32)
    add = rng.standard_normal((n_rows, width)).astype(np.float32)
    add2 = rng.standard_normal((n_rows + 3, width)).astype(np.float32)
    gate = rng.standard_normal((n_rows, width)).astype(np.float32)
    pitch = (width + 15) // 16 * 16 + 32
    rp, c, v = _dev_csr(rowptr, col, val, dev)
    (_, xd), (_, ad), (_, a2d), (_, gd) = (_pitched(t, dev, pitch) for t in (x, add, add2, gate))
    agg = cp.spmm_f64(rowptr, col, val, x)
    tol = lambda ref: 1e-4 + 1e-4 * np.abs(ref) + 1e-6 * np.sqrt(8800)
    y = ops.spmm_csr(rp, c, v, xd, n_cols, "sum", 0.9, ad, 0.9, plan=_plan(variant, rp),
                     add2=a2d, add2_scale=0.1)
    want = 0.9 * agg + 0.9 * add + 0.1 * add2[:n_rows]
    assert np.all(np.abs(y.cpu().numpy() - want) <= tol(want))
    y = ops.spmm_csr(rp, c, v, xd, n_cols, "sum", 0.9, None, 0.0, plan=_plan(variant, rp),
                     add2=a2d, add2_scale=-2.0, gate=gd)
    want = np.where(gate > 0, 0.9 * agg - 2.0 * add2[:n_rows], 0.0)
    assert np.all(np.abs(y.cpu().numpy() - want) <= tol(want))
    with pytest.raises(ValueError):
        ops.spmm_csr(rp, c, v, xd, n_cols, "sum", add2=a2d[:n_rows - 1], add2_scale=1.0)
    from torch_geometric_autoscale import _lib
    _lib.check_device_errors()


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
                         plan=_plan(variant, rp))
        tol = 1e-4 + 1e-4 * np.abs(y64) + 1e-6 * np.sqrt(6000)  # long fp32 sums
        assert np.all(np.abs(y.cpu().numpy() - y64) <= tol)
    # called twice in a row: rows combined from several segments start from zero again
    y2 = ops.spmm_csr(rp, c, v, torch.from_numpy(x).to(dev), n_cols, "mean",
                      plan=_plan(variant, rp))
    assert torch.allclose(y, y2, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("width", [256, 128, 36, 41, 48, 1024])
def test_spmm_combine_buffer_is_left_clean(ops, dev, width):
    """Rows cut into several segments meet in the plan's combine buffer: many launches in a row
    over ONE plan (different widths use different parts of it) give the result of the zeroing
    path, into an output full of garbage, and every launch leaves the buffer all zero."""
    rng = np.random.default_rng(12)
    n_rows, n_cols = 40, 3000
    rowptr, col, val = random_csr(rng, n_rows, n_cols, 10, hubs={3: 700, 11: 2999, 39: 130})
    x = rng.standard_normal((n_cols, width)).astype(np.float32)
    add = rng.standard_normal((n_rows, width)).astype(np.float32)
    rp, c, v = _dev_csr(rowptr, col, val, dev)
    plan = ops.SpmmPlan.build(rp, hub_threshold=24, segment_edges=128)
    assert plan.n_multi == 3 and plan.combine is not None
    xd, addd = torch.from_numpy(x).to(dev), torch.from_numpy(add).to(dev)
    y64 = cp.spmm_f64(rowptr, col, val, x, "sum") * 0.5 + 2.0 * add
    ref = ops.spmm_csr(rp, c, v, xd, n_cols, "sum", 0.5, addd, 2.0, plan=plan.without_combine())
    for _ in range(5):
        out = torch.full((n_rows, width), float("nan"), device=dev)
        y = ops.spmm_csr(rp, c, v, xd, n_cols, "sum", 0.5, addd, 2.0, out=out, plan=plan)
        tol = 1e-4 + 1e-4 * np.abs(y64) + 1e-6 * np.sqrt(3000)
        assert np.all(np.abs(y.cpu().numpy() - y64) <= tol)
        assert torch.allclose(y, ref, rtol=1e-5, atol=1e-5)
        assert int(torch.count_nonzero(plan.combine.view(torch.int32))) == 0
    from torch_geometric_autoscale import _lib
    _lib.check_device_errors()


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


def test_spmm_plan_is_validated(ops, dev, variant):
    """Plan entries outside the matrix are not followed: they are reported, and the rows the
    plan does name correctly are still written.  Row pointers outside the edge arrays are
    treated the same way."""
    from torch_geometric_autoscale import _lib
    rp = torch.tensor([0, 1, 2, 3], dtype=torch.int32, device=dev)
    c = torch.tensor([0, 1, 2], dtype=torch.int32, device=dev)
    x = torch.arange(12, dtype=torch.float32, device=dev).view(3, 4)
    good = ops.SpmmPlan.build(rp)
    assert good.order.tolist() == [0, 1, 2] and good.n_hub == 0 and good.n_items == 0
    assert torch.equal(ops.spmm_csr(rp, c, None, x, 3, plan=good).cpu(), x.cpu())
    with pytest.raises(ValueError):
        ops.spmm_csr(rp[:3].contiguous(), c, None, x, 3, plan=good)       # built for 3 rows
    bad = ops.SpmmPlan(torch.tensor([2, 7, 0], dtype=torch.int32, device=dev), good.items,
                       good.multi_rows, 0, 0, 0, good.hub_threshold, good.segment_edges)
    y = ops.spmm_csr(rp, c, None, x, 3, plan=bad)
    assert torch.equal(y[[0, 2]].cpu(), x[[0, 2]].cpu())
    with pytest.raises(RuntimeError):
        _lib.check_device_errors()
    # an item naming a row / a segment the matrix does not have
    items = torch.tensor([[9, 0], [1, 5]], dtype=torch.int32, device=dev)
    bad = ops.SpmmPlan(good.order, items, good.multi_rows, 2, 2, 0, 1, 64)
    y = ops.spmm_csr(rp, c, None, x, 3, plan=bad)
    assert torch.equal(y[2].cpu(), x[2].cpu())
    with pytest.raises(RuntimeError):
        _lib.check_device_errors()
    # row pointers that leave the edge arrays
    wild = torch.tensor([0, 1, 900000, 3], dtype=torch.int32, device=dev)
    y = ops.spmm_csr(wild, c, None, x, 3)
    assert torch.equal(y[0].cpu(), x[0].cpu())
    with pytest.raises(RuntimeError):
        _lib.check_device_errors()
    _lib.check_device_errors()


def test_plan_host_and_device_builders_agree(ops, dev):
    """The plan is integer work: host and device builders give identical arrays, for a whole
    row-pointer array and for a window of one."""
    rng = np.random.default_rng(5)
    rowptr, _, _ = random_csr(rng, 3000, 9000, 14, hubs={0: 129, 5: 128, 17: 4097, 18: 1024,
                                                         19: 1025, 2999: 8000},
                              empty_rows=(1, 2, 2998), weighted=False)
    rp = torch.from_numpy(rowptr.astype(np.int32))
    for lo, hi in ((0, 3000), (10, 2500), (17, 20)):
        for kw in ({}, {"hub_threshold": 24, "segment_edges": 128}):
            win = rp[lo:hi + 1]
            h = ops.SpmmPlan.build(win, **kw)
            d = ops.SpmmPlan.build(win.to(dev), **kw)
            assert (h.n_hub, h.n_items, h.n_multi, h.hub_threshold, h.segment_edges) == \
                   (d.n_hub, d.n_items, d.n_multi, d.hub_threshold, d.segment_edges)
            assert torch.equal(h.order, d.order.cpu())
            assert torch.equal(h.items[:h.n_items], d.items[:d.n_items].cpu())
            assert torch.equal(h.multi_rows[:h.n_multi], d.multi_rows[:d.n_multi].cpu())
            deg = np.diff(rowptr[lo:hi + 1])
            assert h.n_hub == int((deg > h.hub_threshold).sum())
            assert h.n_items == int(sum(-(-x // h.segment_edges) for x in deg[deg > h.hub_threshold]))
            _check_plan_order(h, deg)


def _check_plan_order(plan, deg):
    """Long rows first; each part by (chunk of the row, descending length, row id)."""
    from torch_geometric_autoscale import _lib
    n = deg.size
    chunk = int(_lib.lib().incagg_spmm_plan_chunk_rows(n))
    rows = np.arange(n)
    key = np.stack([deg <= plan.hub_threshold, rows // chunk if chunk else 0 * rows, -deg, rows])
    want = np.lexsort(key[::-1])
    assert np.array_equal(plan.order.numpy(), want)


@pytest.fixture
def chunked_plans():
    """Plans of matrices of more than 300 rows ordered in chunks of 64 rows."""
    from torch_geometric_autoscale import _lib
    _lib.set_tuning("plan_local_rows", 300)
    _lib.set_tuning("plan_chunk_rows", 64)
    yield
    _lib.set_tuning("plan_local_rows", 32768)
    _lib.set_tuning("plan_chunk_rows", 1024)


def test_chunked_plan_order(ops, dev, chunked_plans):
    """Large matrices: rows ordered chunk by chunk, by both builders alike; the default
    thresholds on a matrix of 40 000 rows, small ones forced on a matrix of 3 000."""
    from torch_geometric_autoscale import _lib
    rng = np.random.default_rng(6)
    rowptr, _, _ = random_csr(rng, 3000, 9000, 14, hubs={0: 129, 5: 128, 17: 4097, 70: 1025,
                                                         2999: 8000}, empty_rows=(1, 2, 2998),
                              weighted=False)
    cases = [(rowptr, 64)]
    _lib.set_tuning("plan_local_rows", 32768)
    _lib.set_tuning("plan_chunk_rows", 1024)
    big, _, _ = random_csr(rng, 40000, 9000, 5, hubs={7: 300, 39000: 5000}, weighted=False)
    assert int(_lib.lib().incagg_spmm_plan_chunk_rows(40000)) == 1024
    assert int(_lib.lib().incagg_spmm_plan_chunk_rows(32768)) == 0
    for rp_np, chunk in ((big, 1024), (rowptr, 64)):
        if chunk == 64:
            _lib.set_tuning("plan_local_rows", 300)
            _lib.set_tuning("plan_chunk_rows", 64)
        rp = torch.from_numpy(rp_np.astype(np.int32))
        h, d = ops.SpmmPlan.build(rp), ops.SpmmPlan.build(rp.to(dev))
        assert (h.n_hub, h.n_items, h.n_multi) == (d.n_hub, d.n_items, d.n_multi)
        assert torch.equal(h.order, d.order.cpu())
        assert torch.equal(h.items[:h.n_items], d.items[:d.n_items].cpu())
        assert torch.equal(h.multi_rows[:h.n_multi], d.multi_rows[:d.n_multi].cpu())
        deg = np.diff(rp_np)
        assert int(_lib.lib().incagg_spmm_plan_chunk_rows(deg.size)) == chunk
        _check_plan_order(h, deg)
        first_short = h.order[h.n_hub:h.n_hub + 8].numpy()
        assert np.all(first_short < chunk)          # chunk 0 leads the short rows


@pytest.mark.parametrize("width", [8, 41, 47, 36, 64, 128, 256])
def test_spmm_with_chunked_plan_matches_oracle(ops, dev, width, chunked_plans):
    """Narrow matrices under a chunk-ordered plan: the workgroups of an XCD walk one contiguous
    range of the plan (1, 2 and 4 slices; 8 slices as before)."""
    from torch_geometric_autoscale import _lib
    rng = np.random.default_rng(width)
    for b, m in ((1900, 5000), (333, 333)):
        rowptr, col, val = random_csr(rng, b, m, 9, hubs={3: 700, 64: 4200, b - 1: 130},
                                      empty_rows=(9, 65), weighted=True)
        x = rng.standard_normal((m, width)).astype(np.float32)
        h64 = cp.spmm_f64(rowptr, col, val, x, "sum")
        rp, c, v = _dev_csr(rowptr, col, val, dev)
        plan = ops.SpmmPlan.build(rp)
        xd = ops.rows_empty(m, width, dev)
        xd.copy_(torch.from_numpy(x))
        _check(ops.spmm_csr(rp, c, v, xd, m, plan=plan), h64)
        _check(ops.spmm_csr(rp, c, v, xd, m, plan=plan.with_edges(ops.pack_edges(c, v))), h64)
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
    _check(ops.spmm_csr_incagg(*args, plan=_plan(variant, rp)), h64, h32)
    # with the epilogue: scale * (agg + m_ag) + add_scale * add
    add = rng.standard_normal((b, width)).astype(np.float32)
    h = ops.spmm_csr_incagg(*args, 0.9, torch.from_numpy(add).to(dev), 0.1,
                            plan=_plan(variant, rp))
    _check(h, 0.9 * h64 + 0.1 * add)
    # the two-pass form agrees
    d = ops.rows_sub(args[3], args[4])
    assert torch.equal(d.cpu(), torch.from_numpy(x - mi))
    # ... also with the relu of the layer below applied to the first operand as it is read
    d = ops.rows_sub(args[3], args[4], relu_a=True)
    assert torch.equal(d.cpu(), torch.from_numpy(np.maximum(x, 0) - mi))


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


@pytest.mark.parametrize("w_src,w_dst", [(41, 41), (41, 47), (37, 48), (3, 9), (47, 47), (5, 6)])
def test_chunk_copy_odd_widths_on_aligned_rows(ops, dev, w_src, w_dst):
    """Rows that start 16-byte aligned on both sides but whose widths are not multiples of 4: the
    vector path with element-wise handling of the quad across each row end; columns past the
    destination width (the row's padding) stay untouched."""
    rng = np.random.default_rng(w_src * 64 + w_dst)
    n, pitch = 700, 48
    base = torch.full((n, pitch), 7.0, device=dev)
    table = base[:, :w_dst]
    offset = np.array([600, 0, 130, 300], dtype=np.int64)
    count = np.array([100, 57, 0, 33], dtype=np.int64)
    xb = torch.full((int(count.sum()), pitch), -5.0, device=dev)
    x = rng.standard_normal((int(count.sum()), w_src)).astype(np.float32)
    xb[:, :w_src] = torch.from_numpy(x).to(dev)
    for relu in (False, True):
        base.fill_(7.0)
        ops.copy_row_chunks(table, xb[:, :w_src], torch.from_numpy(offset), torch.from_numpy(count),
                            packed_is_src=True, relu=relu)
        ref = np.full((n, pitch), 7.0, dtype=np.float32)
        s = 0
        for o, c in zip(offset, count):
            ref[o:o + c, :w_src] = np.maximum(x[s:s + c], 0) if relu else x[s:s + c]
            ref[o:o + c, w_src:w_dst] = 0
            s += c
        assert np.array_equal(base.cpu().numpy(), ref)
    # and back out of the table (packed side is the destination)
    out = torch.full((int(count.sum()), pitch), 9.0, device=dev)
    ops.copy_row_chunks(out[:, :w_dst], table, torch.from_numpy(offset), torch.from_numpy(count),
                        packed_is_src=False)
    got = out.cpu().numpy()
    assert np.array_equal(got[:, :w_src], np.maximum(x, 0))
    assert np.all(got[:, w_dst:] == 9.0)


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
    ro = _plan(variant, rp)
    y1 = ops.spmm_csr(rp, c, v, x1, m, plan=ro)
    y2 = ops.spmm_csr(rp, c, v, x2, m, plan=ro)
    y12 = ops.spmm_csr(rp, c, v, x1 + 2 * x2, m, plan=ro)
    err = (y12 - (y1 + 2 * y2)).abs().max().item()
    # hub rows sum 17k terms in fp32: in pieces of <= 64 with a plan, as one serial chain without
    tol = 2e-3 if ro is not None else 2e-2
    assert err <= tol, err
    ones = torch.ones(m, width, device=dev)
    rs = ops.spmm_csr(rp, c, v, ones, m, plan=ro)
    want = np.add.reduceat(np.concatenate([val, [0]]).astype(np.float64),
                           np.minimum(rowptr[:-1], val.size))
    want[np.diff(rowptr) == 0] = 0
    assert np.allclose(rs[:, 0].cpu().numpy(), want, atol=tol)
    mean1 = ops.spmm_csr(rp, c, None, ones * 3, m, "mean", plan=ro)
    nonempty = torch.from_numpy(np.diff(rowptr) > 0).to(dev)
    assert torch.allclose(mean1[nonempty], torch.full_like(mean1[nonempty], 3.0), atol=1e-4)
    # gather then scatter back is the identity on a History-sized table
    table = torch.randn(200000, width, device=dev, generator=g)
    n_id = torch.randperm(200000, device=dev)[:m]
    rows = ops.gather_rows(table, n_id)
    t2 = table.clone()
    ops.scatter_rows(t2, n_id, rows)
    assert torch.equal(t2, table)


# ---------------------------------------------------------------- split source / in-place rows
def _straddling_batch(rng, b, m, hubs):
    """[b, m] adjacency whose hub rows reference columns on both sides of column b."""
    rowptr, col, val = random_csr(rng, b, m, 11, hubs=hubs, empty_rows=(1, b - 2))
    for r in dict(hubs):
        cs = col[rowptr[r]:rowptr[r + 1]]
        assert cs.min() < b <= cs.max(), "hub row must straddle the split"
    return rowptr, col, val


@pytest.fixture(params=["offsets24", "offsets64"])
def addressing(request):
    """Split-source launches compute row offsets with a 24-bit multiply-add when row ids and
    pitches allow it (the default) or in 64 bits (forced here)."""
    from torch_geometric_autoscale import _lib
    _lib.set_tuning("wide_addressing", 1 if request.param == "offsets64" else -1)
    yield request.param
    _lib.set_tuning("wide_addressing", -1)


@pytest.mark.parametrize("width", [256, 128, 41])
@pytest.mark.parametrize("form", ["pulled", "table"])
def test_spmm_split_source_matches_oracle(ops, dev, width, form, variant, addressing):
    """The aggregation as the push/pull step launches it: source rows < B from the fresh batch
    matrix, the rest either from a pulled matrix or in place from a whole History table
    through rewritten column ids (B + global id); hub rows (> 128 and > 1 024 edges) on both
    sides of the split.  Oracle: the fp64 row loop on the concatenated source."""
    from torch_geometric_autoscale import _lib
    rng = np.random.default_rng(width * 3 + (form == "table"))
    b, m, n_table, table_w = 400, 3000, 5000, 256
    rowptr, col, val = _straddling_batch(rng, b, m, {3: 140, 50: 700, 51: 1500, 399: 2900})
    halo_ids = rng.permutation(n_table)[:m - b].astype(np.int64)
    x = rng.standard_normal((b, width)).astype(np.float32)
    table = rng.standard_normal((n_table, table_w)).astype(np.float32)
    src = np.concatenate([x, table[halo_ids, :width]])
    rp, c, v = _dev_csr(rowptr, col, val, dev)
    xd = torch.from_numpy(x).to(dev)
    td = torch.from_numpy(table).to(dev)
    for reduce in ("sum", "mean"):
        y64 = cp.spmm_f64(rowptr, col, val if reduce == "sum" else None, src, reduce)
        vv = v if reduce == "sum" else None
        if form == "pulled":
            halo = ops.gather_rows(td, torch.from_numpy(halo_ids).to(dev))[:, :width]
            y = ops.spmm_csr(rp, c, vv, xd, m, reduce, x_halo=halo, plan=_plan(variant, rp))
        else:
            cip = np.where(col < b, col, b + halo_ids[np.maximum(col - b, 0)]).astype(np.int32)
            y = ops.spmm_csr(rp, torch.from_numpy(cip).to(dev), vv, xd, b + n_table, reduce,
                             x_halo=td[:, :width], plan=_plan(variant, rp))
        tol = tolerance(y64) + 1e-6 * np.sqrt(2900)
        err = np.abs(y.cpu().numpy() - y64)
        assert np.all(err <= tol), f"{form} {reduce}: max err {err.max()}"
    _lib.check_device_errors()


def test_spmm_in_place_through_sparse_tensor(dev):
    """spmm(adj, x, x_halo=table, halo_ids=...) builds the rewritten column ids itself and
    agrees with the pulled form, also after the same adjacency is called with another (equal)
    id tensor: the id array is refilled in place, never replaced."""
    import torch_geometric_autoscale as tga
    rng = np.random.default_rng(77)
    b, m, n_table, w = 300, 1500, 4000, 256
    rowptr, col, val = _straddling_batch(rng, b, m, {7: 600, 100: 1400})
    adj = tga.SparseTensor(rowptr=torch.from_numpy(rowptr), col=torch.from_numpy(col),
                           value=torch.from_numpy(val), sparse_sizes=(b, m), is_sorted=True).to(dev)
    halo_ids = torch.from_numpy(rng.permutation(n_table)[:m - b].astype(np.int64)).to(dev)
    x = torch.randn(b, w, device=dev)
    table = torch.randn(n_table, w, device=dev)
    want = tga.spmm(adj, x, x_halo=table[halo_ids])
    got = tga.spmm(adj, x, x_halo=table, halo_ids=halo_ids)
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-5)
    first = adj.col_in_place(halo_ids, b)
    again = adj.col_in_place(halo_ids.clone(), b)
    assert again.data_ptr() == first.data_ptr()
    other = halo_ids.flip(0).contiguous()
    got2 = tga.spmm(adj, x, x_halo=table, halo_ids=other)
    assert adj.col_in_place(other, b).data_ptr() == first.data_ptr()
    assert torch.allclose(got2, tga.spmm(adj, x, x_halo=table[other]), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("reduce", ["sum", "mean"])
def test_unaligned_operands_are_realigned(dev, reduce):
    """A contiguous [rows, 41] source and a contiguous [rows, 41] output gradient (rows that do
    not start 16-byte aligned) are copied to the padded pitch before they are gathered from:
    same results as the oracle, forward and backward, and the caller's tensors are untouched."""
    import torch_geometric_autoscale as tga
    rng = np.random.default_rng(41)
    b, m, w = 4500, 6000, 41
    rowptr, col, val = random_csr(rng, b, m, 9, hubs={3: 900}, empty_rows=(0,),
                                  weighted=(reduce == "sum"))
    adj = tga.SparseTensor(rowptr=torch.from_numpy(rowptr), col=torch.from_numpy(col),
                           value=None if val is None else torch.from_numpy(val),
                           sparse_sizes=(b, m), is_sorted=True).to(dev)
    x_np = rng.standard_normal((m, w)).astype(np.float32)
    dy_np = rng.standard_normal((b, w)).astype(np.float32)
    x = torch.from_numpy(x_np).to(dev).requires_grad_()
    assert x.stride(0) % 4 != 0
    y = tga.spmm(adj, x, reduce)
    _check(y.detach(), cp.spmm_f64(rowptr, col, val, x_np, reduce))
    dy = torch.from_numpy(dy_np).to(dev)
    y.backward(dy)
    want = cp.spmm_t(rowptr, col, val, dy_np, m, reduce)
    np.testing.assert_allclose(x.grad.cpu().numpy(), want, rtol=1e-4, atol=1e-4)
    assert torch.equal(x.detach().cpu(), torch.from_numpy(x_np))
    assert torch.equal(dy.cpu(), torch.from_numpy(dy_np))
    from torch_geometric_autoscale import _lib
    _lib.check_device_errors()


def test_spmm_single_source_beyond_4gib(ops, dev, variant):
    """One source matrix of more than 4 GiB: row offsets need 64 bits in every kernel variant.
    The oracle runs on the referenced rows only."""
    width = 256
    n_cols = (1 << 32) // (width * 4) + 70000         # ~4.37 GiB of fp32 rows
    rng = np.random.default_rng(4)
    n_rows = 200
    rowptr, col, val = random_csr(rng, n_rows, n_cols, 15, hubs={2: 300, 9: 1100})
    col[-1] = n_cols - 1                               # the very last row is referenced
    col[rowptr[2]:rowptr[3]] = np.sort(rng.choice(np.arange(n_cols - 5000, n_cols - 1),
                                                  size=300, replace=False))
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(n_cols, width, device=dev, generator=g)
    uniq, inv = np.unique(col, return_inverse=True)
    xs = x[torch.from_numpy(uniq).to(dev)].cpu().numpy()
    y64 = cp.spmm_f64(rowptr, inv.astype(np.int64), val, xs)
    rp, c, v = _dev_csr(rowptr, col, val, dev)
    y = ops.spmm_csr(rp, c, v, x, n_cols, plan=_plan(variant, rp))
    tol = tolerance(y64) + 1e-6 * np.sqrt(1100)
    assert np.all(np.abs(y.cpu().numpy() - y64) <= tol)
    del x
    torch.cuda.empty_cache()


@pytest.mark.parametrize("reduce", ["sum", "mean"])
def test_transposed_windows_width_256(dev, reduce, variant):
    """The two backward launches of a push/pull step at the headline width: the [M, B]
    transposed adjacency (layer 0) and its [B, B] window of in-batch columns (other layers,
    source split in fresh + pulled rows), each with its own degree order."""
    import torch_geometric_autoscale as tga
    rng = np.random.default_rng(8 + (reduce == "mean"))
    b, m, w = 500, 2600, 256
    rowptr, col, val = _straddling_batch(rng, b, m, {0: 900, 250: 2000})
    # columns referenced by hundreds of rows: long rows of the transposed matrix
    for r in range(0, b, 2):
        if rowptr[r + 1] > rowptr[r]:
            col[rowptr[r]] = 0 if 0 not in col[rowptr[r]:rowptr[r + 1]] else col[rowptr[r]]
    for r in range(b):
        col[rowptr[r]:rowptr[r + 1]].sort()
    weighted = reduce == "sum"
    adj = tga.SparseTensor(rowptr=torch.from_numpy(rowptr), col=torch.from_numpy(col),
                           value=torch.from_numpy(val) if weighted else None,
                           sparse_sizes=(b, m), is_sorted=True).to(dev)
    v_or = val if weighted else None
    dy = torch.randn(b, w, device=dev)
    want = cp.spmm_t(rowptr, col, v_or, dy.cpu().numpy(), m, reduce)
    # [M, B]: gradient for every source row
    x = torch.randn(m, w, device=dev, requires_grad=True)
    tga.spmm(adj, x, reduce).backward(dy)
    np.testing.assert_allclose(x.grad.cpu().numpy(), want, rtol=1e-4, atol=1e-4)
    # [B, B] window: only the fresh rows get a gradient
    xb = torch.randn(b, w, device=dev, requires_grad=True)
    halo = torch.randn(m - b, w, device=dev)
    tga.spmm(adj, xb, reduce, x_halo=halo).backward(dy)
    np.testing.assert_allclose(xb.grad.cpu().numpy(), want[:b], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("width", [256, 41])
def test_spmm_fused_activation_and_gate(ops, dev, width, variant):
    """relu applied to the fresh source rows as they are gathered (rows of x_halo untouched)
    and the relu backward applied to the result (zero where gate <= 0), against the oracle on
    the explicitly activated source and the explicitly masked result."""
    from torch_geometric_autoscale import _lib
    rng = np.random.default_rng(width + 5)
    b, m = 300, 1500
    rowptr, col, val = _straddling_batch(rng, b, m, {2: 200, 40: 1400})
    x = rng.standard_normal((b, width)).astype(np.float32)
    halo = rng.standard_normal((m - b, width)).astype(np.float32)
    gate = rng.standard_normal((b + 3, width)).astype(np.float32)
    gate[5] = 0.0                                      # zero counts as "not positive"
    rp, c, v = _dev_csr(rowptr, col, val, dev)
    xd, hd, gd = (torch.from_numpy(a).to(dev) for a in (x, halo, gate))
    plan = _plan(variant, rp)
    # split source: only the rows of x are activated
    y64 = cp.spmm_f64(rowptr, col, val, np.concatenate([np.maximum(x, 0), halo]))
    y = ops.spmm_csr(rp, c, v, xd, m, x_halo=hd, plan=plan, relu_x=True)
    tol = tolerance(y64) + 1e-6 * np.sqrt(1400)
    assert np.all(np.abs(y.cpu().numpy() - y64) <= tol)
    # one source, activation and gate together
    src = np.concatenate([x, halo])
    y64 = cp.spmm_f64(rowptr, col, val, np.maximum(src, 0))
    y64 = np.where(gate[:b] > 0, y64, 0.0)
    y = ops.spmm_csr(rp, c, v, torch.from_numpy(src).to(dev), m, plan=plan, relu_x=True, gate=gd)
    got = y.cpu().numpy()
    assert np.all(np.abs(got - y64) <= tol)
    assert np.all(got[gate[:b] <= 0] == 0.0)
    _lib.check_device_errors()


def test_chunk_push_with_activation(ops, dev):
    rng = np.random.default_rng(12)
    table = rng.standard_normal((900, 256)).astype(np.float32)
    x = rng.standard_normal((300, 200)).astype(np.float32)
    offset, count = np.array([600, 10], dtype=np.int64), np.array([100, 200], dtype=np.int64)
    ref = table.copy()
    cp.history_push(ref, np.pad(np.maximum(x, 0), ((0, 0), (0, 56))), np.arange(300), offset, count)
    td = torch.from_numpy(table).to(dev)
    ops.copy_row_chunks(td, torch.from_numpy(x).to(dev), torch.from_numpy(offset),
                        torch.from_numpy(count), packed_is_src=True, relu=True)
    assert np.array_equal(td.cpu().numpy(), ref)


@pytest.mark.parametrize("shape", [(8, 256, 256, 7), (9, 256, 256, 0), (32, 256, 100, 26), (6, 47, 256, 5)])
def test_tn_finish_matches_torch(ops, dev, shape):
    """Slab sum + remainder rows of a weight-gradient product in one launch."""
    s_, m, n, r = shape
    g = torch.Generator(device=dev).manual_seed(3)
    parts = torch.randn(s_, m, n, device=dev, generator=g)
    a = torch.randn(40, m + 8, device=dev, generator=g)[:r, :m]      # strided remainder rows
    b = torch.randn(40, n + 32, device=dev, generator=g)[:r, :n]
    got = ops.tn_finish(parts, a if r else None, b if r else None)
    want = parts.double().sum(0) + (a.double().t() @ b.double() if r else 0)
    assert torch.allclose(got.double(), want, rtol=1e-5, atol=1e-5)
    # the product it finishes: a^T b over tall operands, against torch in fp64
    from torch_geometric_autoscale.nn import _tn_product
    rows = 16327
    x = torch.randn(rows, m, device=dev, generator=g)
    y = torch.randn(rows, n, device=dev, generator=g)
    ref = (x.double().t() @ y.double())
    assert torch.allclose(_tn_product(x, y).double(), ref, rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize("r", [0, 6])
def test_tn_finish_batched_matches_torch(ops, dev, r):
    """The last step of several weight-gradient products at once, and the deferred form of
    nn._tn_product that uses it (one finishing launch for all layers)."""
    from torch_geometric_autoscale.nn import _DeferredProducts, _tn_product
    g = torch.Generator(device=dev).manual_seed(4 + r)
    n_l, s_, m, n = 5, 8, 48, 36
    parts = [torch.randn(s_, m, n, device=dev, generator=g) for _ in range(n_l)]
    tails = [(torch.randn(9, m + 4, device=dev, generator=g)[:r, :m],
              torch.randn(9, n + 12, device=dev, generator=g)[:r, :n]) if r else None
             for _ in range(n_l)]
    out = torch.full((n_l, m, n), float("nan"), device=dev)
    ops.tn_finish_batched(parts, tails, out)
    for l in range(n_l):
        want = parts[l].double().sum(0)
        if r:
            want = want + tails[l][0].double().t() @ tails[l][1].double()
        assert torch.allclose(out[l].double(), want, rtol=1e-5, atol=1e-5)
    with pytest.raises(ValueError):
        ops.tn_finish_batched(parts, tails[:-1], out)
    # deferred products: rows that do (8 * 2000) / do not (16327) fill the slabs
    rows = 16000 if r == 0 else 16327
    xs = [torch.randn(rows, m, device=dev, generator=g) for _ in range(3)]
    ys = [torch.randn(rows, n + 4, device=dev, generator=g)[:, :n] for _ in range(3)]
    for which in ((0, 1, 2), (2, 0)):          # all layers (one launch) / a subset (one each)
        stack = torch.full((3, m, n), float("nan"), device=dev)
        deferred = _DeferredProducts(stack)
        for l in which:
            got = _tn_product(xs[l], ys[l], deferred.slot(l))
            assert got.data_ptr() == stack[l].data_ptr()
        deferred.finish()
        for l in which:
            ref = xs[l].double().t() @ ys[l].double()
            assert torch.allclose(stack[l].double(), ref, rtol=1e-4, atol=2e-3)
