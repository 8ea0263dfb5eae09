"""Pins the CPU oracle: the relabel restatement against outputs of the reference's own
C++ (tests/golden/relabel_golden.npz, made by tests/golden/make_relabel_fixtures.py), the
C aggregation against pure-Python float64 loops on tiny graphs, the copy restatements
against hand-written expectations."""
import numpy as np
import pytest

from helpers import random_csr
from oracle import cpu_path as cp
from oracle.pool_model import PoolModel


def _cases(g):
    return sorted({k.split("/")[0] for k in g.files if k.startswith("c")})


def _run_case(g, tag, one_hop, within):
    gname, iname, vname, bip, op = g[f"{tag}/meta"]
    rowptr, col = g[f"graph/{gname}/rowptr"], g[f"graph/{gname}/col"]
    val = None if vname == "none" else g[f"graph/{gname}/value_{vname}"]
    fn = one_hop if op == "one_hop" else within
    return fn(rowptr, col, val, g[f"{tag}/idx"], bip == "1"), val


def test_relabel_oracle_matches_reference_fixtures(golden_relabel):
    g = golden_relabel
    tags = _cases(g)
    assert len(tags) >= 200
    for tag in tags:
        (rp, c, v, nid), val = _run_case(g, tag, cp.relabel_one_hop, cp.relabel_one_hop_within_batch)
        assert np.array_equal(rp, g[f"{tag}/out_rowptr"]), tag
        assert np.array_equal(c, g[f"{tag}/out_col"]), tag
        assert np.array_equal(nid, g[f"{tag}/out_n_id"]), tag
        if val is not None:
            assert v.dtype == g[f"{tag}/out_value"].dtype
            assert np.array_equal(v, g[f"{tag}/out_value"]), tag
        else:
            assert v is None and f"{tag}/out_value" not in g.files


def test_relabel_known_answer():
    """8-node ring plus the chord (1,5), idx = [0,1,5,6,7]."""
    rowptr = np.array([0, 2, 5, 7, 9, 11, 14, 16, 18])
    col = np.array([1, 7, 0, 2, 5, 1, 3, 2, 4, 3, 5, 1, 4, 6, 5, 7, 0, 6])
    idx = np.array([0, 1, 5, 6, 7])
    rp, c, v, nid = cp.relabel_one_hop(rowptr, col, None, idx, True)
    assert rp.tolist() == [0, 2, 5, 8, 10, 12]
    assert c.tolist() == [1, 4, 0, 5, 2, 1, 6, 3, 2, 4, 0, 3]
    assert nid.tolist() == [0, 1, 5, 6, 7, 2, 4] and v is None
    rp2, *_ = cp.relabel_one_hop(rowptr, col, None, idx, False)
    assert rp2.tolist() == [0, 2, 5, 8, 10, 12, 12, 12]
    rp, c, _, nid = cp.relabel_one_hop_within_batch(rowptr, col, None, idx, True)
    assert rp.tolist() == [0, 2, 4, 6, 8, 10] and c.tolist() == [1, 4, 0, 2, 1, 3, 2, 4, 0, 3]
    assert nid.tolist() == idx.tolist()
    py = cp.relabel_one_hop_py(rowptr, col, None, idx, True)
    assert py[1].tolist() == [1, 4, 0, 5, 2, 1, 6, 3, 2, 4, 0, 3]


@pytest.mark.parametrize("reduce", ["sum", "mean"])
@pytest.mark.parametrize("width", [1, 5, 32])
def test_aggregation_oracle_against_naive_loops(reduce, width):
    rng = np.random.default_rng(width)
    rowptr, col, val = random_csr(rng, 40, 60, 5, hubs={3: 55}, empty_rows=(0, 39),
                                  weighted=(reduce == "sum"))
    x = rng.standard_normal((60, width)).astype(np.float32)
    naive = cp.spmm_naive_f64(rowptr, col, val, x, reduce)
    assert np.allclose(cp.spmm_f64(rowptr, col, val, x, reduce), naive, rtol=1e-12, atol=1e-12)
    assert np.allclose(cp.spmm(rowptr, col, val, x, reduce), naive, rtol=1e-5, atol=1e-5)
    # transposed form: <A x, dy> == <x, A^T dy>
    dy = rng.standard_normal((40, width))
    lhs = float((naive * dy).sum())
    rhs = float((x.astype(np.float64) * cp.spmm_t(rowptr, col, val, dy, 60, reduce)).sum())
    assert abs(lhs - rhs) <= 1e-9 * max(1.0, abs(lhs))


@pytest.mark.parametrize("reduce", ["sum", "mean"])
def test_aggregation_oracle_against_torch_sparse_csr(reduce):
    """The CPU path the reference's arithmetic reduces to without torch-sparse: a
    ``torch.sparse_csr_tensor @ x`` (medium graph with long rows and empty rows), forward and
    transposed, against the C restatement."""
    import torch
    rng = np.random.default_rng(21)
    n_rows, n_cols, width = 3000, 5000, 96
    rowptr, col, val = random_csr(rng, n_rows, n_cols, 12, hubs={5: 2500, 77: 900},
                                  empty_rows=(0, 1234, 2999))
    x = rng.standard_normal((n_cols, width)).astype(np.float32)
    v = val if reduce == "sum" else None
    a = torch.sparse_csr_tensor(torch.from_numpy(rowptr), torch.from_numpy(col),
                                torch.from_numpy(val if v is not None else
                                                 np.ones(col.size, np.float32)),
                                size=(n_rows, n_cols))
    want = (a @ torch.from_numpy(x)).numpy()
    if reduce == "mean":
        want = want / np.maximum(np.diff(rowptr), 1)[:, None]
    got = cp.spmm(rowptr, col, v, x, reduce)
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-4)
    # transposed: A^T dY as the backward pass computes it
    dy = rng.standard_normal((n_rows, width)).astype(np.float32)
    want_t = (a.to_dense().t() @ torch.from_numpy(dy)).numpy()
    if reduce == "sum":
        np.testing.assert_allclose(cp.spmm_t(rowptr, col, v, dy, n_cols, reduce), want_t,
                                   rtol=1e-4, atol=1e-4)


def test_incagg_oracle_is_the_three_step_formula():
    rng = np.random.default_rng(4)
    b, w = 50, 8
    rowptr, col, val = random_csr(rng, b, b, 6)
    x, m_in, m_ag = (rng.standard_normal((b, w)).astype(np.float32) for _ in range(3))
    want = cp.spmm(rowptr, col, val, x - m_in) + m_ag
    assert np.array_equal(cp.incagg(rowptr, col, val, x, m_in, m_ag), want)


def test_copy_oracles():
    emb = np.zeros((10, 2), dtype=np.float32)
    x = np.arange(8, dtype=np.float32).reshape(4, 2)
    cp.history_push(emb, x, np.arange(4), np.array([6, 1]), np.array([3, 1]))
    assert emb[6:9].tolist() == x[:3].tolist() and emb[1].tolist() == x[3].tolist()
    assert cp.history_pull(emb, np.array([8, 1])).tolist() == [x[2].tolist(), x[3].tolist()]
    with pytest.raises(ValueError):
        cp.history_push(emb, x)
    dst = np.full((6, 2), -1, dtype=np.float32)
    cp.read_async(emb, np.array([6]), np.array([2]), np.array([1, 0]), dst)
    assert dst[:4].tolist() == [x[0].tolist(), x[1].tolist(), x[3].tolist(), [0, 0]]
    assert dst[4:].tolist() == [[-1, -1], [-1, -1]]
    back = np.zeros((10, 2), dtype=np.float32)
    cp.write_async(dst, np.array([3, 0]), np.array([2, 1]), back)
    assert back[3:5].tolist() == dst[:2].tolist() and back[0].tolist() == dst[2].tolist()


def test_gcn_norm_oracle_rows():
    rowptr = np.array([0, 1, 3, 4])
    col = np.array([1, 0, 2, 1])
    rp, c, v = cp.set_diag_and_gcn_norm(rowptr, col, 3)
    assert rp.tolist() == [0, 2, 5, 7] and c.tolist() == [0, 1, 0, 1, 2, 1, 2]
    deg = np.array([2.0, 3.0, 2.0])
    want = [1 / 2, 1 / np.sqrt(6), 1 / np.sqrt(6), 1 / 3, 1 / np.sqrt(6), 1 / np.sqrt(6), 1 / 2]
    assert np.allclose(v, want, rtol=1e-6) and deg is not None


def test_pool_model_trace():
    m = PoolModel(2)
    for t in "abcd":
        m.async_pull(t)
    assert [e for e in m.log if e[0] == "launch_pull"] == [("launch_pull", 0, "a"), ("launch_pull", 1, "b")]
    assert m.synchronize_pull() == 0
    m.free_pull()
    assert m.log[-1] == ("launch_pull", 0, "c")
    assert m.synchronize_pull() == 1
    m.free_pull()
    assert m.log[-1] == ("launch_pull", 1, "d")
    m.synchronize_pull(); m.free_pull(); m.synchronize_pull(); m.free_pull()
    assert m.queue == [] and m.pull_index == -1
