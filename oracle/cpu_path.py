"""ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU restatement (numpy + the plain-C objects in this directory) of the reference's
hot path.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it, and only as the checker / the timed baseline;
the shipped package never does.

Parity status, per function:
  * relabel_*            PINNED   (reference C++ built by oracle/build_ref.py ->
                                   tests/golden/relabel_*.npz)
  * history_* / *_async  restate plain copies spelled out in the reference itself
                         (history.py:33-65, csrc/cuda/async_cuda.cu:62-112,139-163)
  * spmm / incagg / layer formulas  PARITY UNPINNED: the arithmetic lives in
                         torch-sparse / torch-geometric, un-vendored and un-pinned
                         in the reference, with no test or golden vector there.

Every array argument is a numpy array; indices are int64 as in the reference.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

_i64p = ctypes.POINTER(ctypes.c_int64)
_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)


def build(force=False):
    srcs = [os.path.join(_HERE, s) for s in ("relabel_oracle.c", "aggregate_oracle.c")]
    if (force or not os.path.isfile(_SO)
            or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.isfile(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_get_threads.restype = ctypes.c_int
    return _lib


def _p(a, typ):
    return None if a is None else a.ctypes.data_as(typ)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def set_threads(n):
    lib().oracle_set_threads(ctypes.c_int(int(n)))


def get_threads():
    return int(lib().oracle_get_threads())


# --------------------------------------------------------------------------- relabel
def _relabel(fn_name, rowptr, col, value, idx, bipartite):
    rowptr, col, idx = _i64(rowptr), _i64(col), _i64(idx)
    n = idx.size
    ub = int((rowptr[idx + 1] - rowptr[idx]).sum()) if n else 0
    out_rowptr = np.empty(2 * n + 1 + ub, dtype=np.int64)
    out_col = np.empty(max(ub, 1), dtype=np.int64)
    out_nid = np.empty(n + ub, dtype=np.int64)
    out_val, vsize, vptr, ovptr = None, 0, None, None
    if value is not None:
        value = np.ascontiguousarray(value)
        vsize = value.dtype.itemsize
        out_val = np.empty(max(ub, 1), dtype=value.dtype)
        vptr = value.ctypes.data_as(ctypes.c_void_p)
        ovptr = out_val.ctypes.data_as(ctypes.c_void_p)
    counts = np.zeros(3, dtype=np.int64)
    rc = getattr(lib(), fn_name)(
        _p(rowptr, _i64p), _p(col, _i64p), vptr, ctypes.c_int64(vsize), _p(idx, _i64p),
        ctypes.c_int64(n), ctypes.c_int(1 if bipartite else 0), _p(out_rowptr, _i64p),
        _p(out_col, _i64p), ovptr, _p(out_nid, _i64p), _p(counts, _i64p))
    if rc != 0:
        raise MemoryError("oracle relabel failed")
    nnz, n_total, rp_len = (int(c) for c in counts)
    return (out_rowptr[:rp_len].copy(), out_col[:nnz].copy(),
            None if out_val is None else out_val[:nnz].copy(), out_nid[:n_total].copy())


def relabel_one_hop(rowptr, col, value, idx, bipartite=True):
    """csrc/cpu/relabel_cpu.cpp:3-108."""
    return _relabel("oracle_relabel_one_hop", rowptr, col, value, idx, bipartite)


def relabel_one_hop_within_batch(rowptr, col, value, idx, bipartite=True):
    """csrc/cpu/relabel_cpu.cpp:111-214."""
    return _relabel("oracle_relabel_one_hop_within_batch", rowptr, col, value, idx, bipartite)


def relabel_one_hop_py(rowptr, col, value, idx, bipartite=True):
    """Pure-Python dict form of relabel_cpu.cpp:3-108 for tiny cases (cross-check of the C)."""
    n = len(idx)
    m = {}
    out_rowptr = [0]
    for i, v in enumerate(idx):
        m[int(v)] = i
        out_rowptr.append(out_rowptr[-1] + int(rowptr[v + 1] - rowptr[v]))
    new, out_col, out_val = [], [], []
    for v in idx:
        for j in range(int(rowptr[v]), int(rowptr[v + 1])):
            w = int(col[j])
            if w not in m:
                m[w] = n + len(new)
                new.append(w)
            out_col.append(m[w])
            if value is not None:
                out_val.append(value[j])
    if not bipartite:
        out_rowptr += [len(out_col)] * len(new)
    return (np.array(out_rowptr, dtype=np.int64), np.array(out_col, dtype=np.int64),
            None if value is None else np.array(out_val, dtype=np.asarray(value).dtype),
            np.array(list(map(int, idx)) + new, dtype=np.int64))


# --------------------------------------------------------------------------- aggregate
_REDUCE = {"sum": 0, "add": 0, "mean": 1}


def spmm(rowptr, col, val, x, reduce="sum", n_rows=None):
    """`adj_t @ x` / `spmm(adj_t, x, reduce)`: fp32 row loop in storage order."""
    rowptr, col = _i64(rowptr), _i64(col)
    x = np.ascontiguousarray(x, dtype=np.float32)
    val = None if val is None else np.ascontiguousarray(val, dtype=np.float32)
    n_rows = rowptr.size - 1 if n_rows is None else n_rows
    y = np.empty((n_rows, x.shape[1]), dtype=np.float32)
    lib().oracle_spmm_csr_f32(_p(rowptr, _i64p), _p(col, _i64p), _p(val, _f32p), _p(x, _f32p),
                              ctypes.c_int64(x.shape[1]), _p(y, _f32p),
                              ctypes.c_int64(x.shape[1]), ctypes.c_int64(n_rows),
                              ctypes.c_int64(x.shape[1]), ctypes.c_int(_REDUCE[reduce]))
    return y


def spmm_f64(rowptr, col, val, x, reduce="sum", n_rows=None):
    """Same aggregation with a float64 accumulator (tolerance yardstick)."""
    rowptr, col = _i64(rowptr), _i64(col)
    x = np.ascontiguousarray(x, dtype=np.float32)
    val = None if val is None else np.ascontiguousarray(val, dtype=np.float32)
    n_rows = rowptr.size - 1 if n_rows is None else n_rows
    y = np.empty((n_rows, x.shape[1]), dtype=np.float64)
    lib().oracle_spmm_csr_f64acc(_p(rowptr, _i64p), _p(col, _i64p), _p(val, _f32p),
                                 _p(x, _f32p), ctypes.c_int64(x.shape[1]), _p(y, _f64p),
                                 ctypes.c_int64(x.shape[1]), ctypes.c_int64(n_rows),
                                 ctypes.c_int64(x.shape[1]), ctypes.c_int(_REDUCE[reduce]))
    return y


def spmm_naive_f64(rowptr, col, val, x, reduce="sum"):
    """Pure-Python float64 loops, tiny cases only: pins the C restatement."""
    n = len(rowptr) - 1
    x = np.asarray(x, dtype=np.float64)
    y = np.zeros((n, x.shape[1]), dtype=np.float64)
    for i in range(n):
        for e in range(int(rowptr[i]), int(rowptr[i + 1])):
            y[i] += (1.0 if val is None else float(val[e])) * x[int(col[e])]
        if reduce == "mean":
            y[i] /= max(int(rowptr[i + 1] - rowptr[i]), 1)
    return y


def incagg(rowptr, col, val, x, m_in, m_ag, reduce="sum"):
    """`adj_t @ (x - M_in) + M_ag` in the reference's operation order (models/gcn2.py:255)."""
    rowptr, col = _i64(rowptr), _i64(col)
    x = np.ascontiguousarray(x, dtype=np.float32)
    m_in = np.ascontiguousarray(m_in, dtype=np.float32)
    m_ag = np.ascontiguousarray(m_ag, dtype=np.float32)
    val = None if val is None else np.ascontiguousarray(val, dtype=np.float32)
    n_rows, n_cols, w = rowptr.size - 1, x.shape[0], x.shape[1]
    h = np.empty((n_rows, w), dtype=np.float32)
    scratch = np.empty((n_cols, w), dtype=np.float32)
    lib().oracle_incagg_f32(_p(rowptr, _i64p), _p(col, _i64p), _p(val, _f32p), _p(x, _f32p),
                            ctypes.c_int64(w), _p(m_in, _f32p), ctypes.c_int64(m_in.shape[1]),
                            _p(m_ag, _f32p), ctypes.c_int64(m_ag.shape[1]), _p(h, _f32p),
                            ctypes.c_int64(w), ctypes.c_int64(n_rows), ctypes.c_int64(n_cols),
                            ctypes.c_int64(w), ctypes.c_int(_REDUCE[reduce]), _p(scratch, _f32p))
    return h


def spmm_t(rowptr, col, val, dy, n_cols, reduce="sum"):
    """Backward of spmm w.r.t. x: dX = A^T dY (autograd through the call sites above)."""
    rowptr, col = _i64(rowptr), _i64(col)
    dy = np.asarray(dy, dtype=np.float64)
    dx = np.zeros((n_cols, dy.shape[1]), dtype=np.float64)
    deg = np.diff(rowptr)
    rows = np.repeat(np.arange(rowptr.size - 1), deg)
    w = np.ones(col.size) if val is None else np.asarray(val, dtype=np.float64)
    if reduce == "mean":
        w = w / np.maximum(deg, 1)[rows]
    np.add.at(dx, col, w[:, None] * dy[rows])
    return dx


# --------------------------------------------------------------------------- history
def history_pull(emb, n_id=None):
    """history.py:33-39."""
    return emb.copy() if n_id is None else emb[_i64(n_id)].copy()


def history_push(emb, x, n_id=None, offset=None, count=None):
    """history.py:41-65 (in place on ``emb``)."""
    if n_id is None and x.shape[0] != emb.shape[0]:
        raise ValueError
    if n_id is None:
        emb[...] = x
    elif offset is None or count is None:
        emb[_i64(n_id)] = x
    else:
        src = 0
        for o, c in zip(np.asarray(offset).tolist(), np.asarray(count).tolist()):
            emb[o:o + c] = x[src:src + c]
            src += c


def read_async(src, offset, count, index, dst):
    """csrc/cuda/async_cuda.cu:62-112: chunks first, then gathered rows, into dst (in place)."""
    numel = 0
    if offset is not None:
        for o, c in zip(np.asarray(offset).tolist(), np.asarray(count).tolist()):
            dst[numel:numel + c] = src[o:o + c]
            numel += c
    index = _i64(index)
    if index.size:
        dst[numel:numel + index.size] = src[index]


def write_async(src, offset, count, dst):
    """csrc/cuda/async_cuda.cu:139-163 (in place on dst)."""
    s = 0
    for o, c in zip(np.asarray(offset).tolist(), np.asarray(count).tolist()):
        dst[o:o + c] = src[s:s + c]
        s += c


# --------------------------------------------------------------------------- graph prep
def set_diag_and_gcn_norm(rowptr, col, n):
    """main.py:147-151: adj.set_diag() then gcn_norm(add_self_loops=False).

    Restated [3P]: self loops replace any existing diagonal with weight 1; deg = row
    sums; A_hat = D^-1/2 A D^-1/2 with inf -> 0.  Input must have sorted columns per row.
    """
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    keep = rows != col
    r = np.concatenate([rows[keep], np.arange(n)])
    c = np.concatenate([col[keep], np.arange(n)])
    order = np.lexsort((c, r))
    r, c = r[order], c[order]
    v = np.ones(r.size, dtype=np.float32)
    deg = np.bincount(r, weights=v, minlength=n).astype(np.float32)
    with np.errstate(divide="ignore"):
        dis = deg ** np.float32(-0.5)
    dis[np.isinf(dis)] = 0
    v = (dis[r] * v * dis[c]).astype(np.float32)
    new_rowptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(r, minlength=n), out=new_rowptr[1:])
    return new_rowptr, c.astype(np.int64), v


# --------------------------------------------------------------------------- layer formulas
def gcn_layer(rowptr, col, val, x, weight, bias):
    """GCNConv(normalize=False) [3P]: linear first, aggregate, then bias
    (split confirmed by the reference's MLP fallback, models/gcn.py:175-178)."""
    h = spmm(rowptr, col, val, x.astype(np.float32) @ weight.T.astype(np.float32))
    return h if bias is None else h + bias


def gcn2_layer(rowptr, col, val, x, x0, alpha, beta, w1, w2=None):
    """GCN2Conv(normalize=False) [3P]; layer index convention models/gcn2.py:46-48."""
    b = rowptr.size - 1
    ax = spmm(rowptr, col, val, x) * np.float32(1 - alpha)
    x0 = x0[:b] * np.float32(alpha)
    if w2 is None:
        p = ax + x0
        return np.float32(1 - beta) * p + np.float32(beta) * (p @ w1)
    out = np.float32(1 - beta) * ax + np.float32(beta) * (ax @ w1)
    return out + np.float32(1 - beta) * x0 + np.float32(beta) * (x0 @ w2)


def sage_layer(rowptr, col, x, wl, bl, wr):
    """models/graphsage.py:623-644: lin_l(mean_agg) + lin_r(x[:B])."""
    b = rowptr.size - 1
    h = spmm(rowptr, col, None, x, reduce="mean") @ wl.T
    if bl is not None:
        h = h + bl
    return h + x[:b] @ wr.T


def appnp_step(rowptr, col, val, x, x0, alpha):
    """models/appnp.py:85."""
    return np.float32(1 - alpha) * spmm(rowptr, col, val, x) + np.float32(alpha) * x0
