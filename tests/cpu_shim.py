"""TEST-ONLY stand-ins that let the host-side orchestration (models, sweeps, loaders, pool
bookkeeping) run on a machine without a GPU: the HIP entry points of
``torch_geometric_autoscale.ops`` are swapped for the CPU oracle for the duration of a
test.  Nothing in the shipped package imports this file; the `-m gpu` tests never use it.
"""
import contextlib

import numpy as np
import torch

from oracle import cpu_path as cp


def _np(t):
    return None if t is None else t.detach().cpu().numpy()


def _check_plan(plan, rowptr32):
    """The plan must describe the rows the call addresses: a permutation, long rows first, each
    part by (chunk of the row, descending length, row id), whose long prefix is covered segment
    by segment."""
    if plan is None:
        return
    n_rows = rowptr32.numel() - 1
    assert plan.n_rows == n_rows and plan.order.dtype == torch.int32
    order = plan.order.to(torch.int64).cpu()
    assert torch.equal(torch.sort(order).values, torch.arange(n_rows))
    deg = (rowptr32[1:] - rowptr32[:-1]).to(torch.int64).cpu()
    from torch_geometric_autoscale import _lib
    chunk = int(_lib.lib().incagg_spmm_plan_chunk_rows(n_rows))
    rows = torch.arange(n_rows)
    key = ((deg <= plan.hub_threshold).to(torch.int64) << 62) \
        | ((rows // chunk if chunk else torch.zeros_like(rows)) << 31) | (0x7fffffff - deg)
    assert torch.equal(order, torch.sort(key, stable=True).indices)
    assert plan.n_hub == int((deg > plan.hub_threshold).sum())
    want = [(int(r), s) for r in order[:plan.n_hub].tolist()
            for s in range(-(-int(deg[r]) // plan.segment_edges))]
    got = [tuple(p) for p in plan.items[:plan.n_items].cpu().tolist()]
    assert got == want
    multi = [int(r) for r in order[:plan.n_hub].tolist() if int(deg[r]) > plan.segment_edges]
    assert plan.multi_rows[:plan.n_multi].cpu().tolist() == multi


def _spmm_csr(rowptr32, col32, val, x, n_cols, reduce="sum", scale=1.0, add=None, add_scale=0.0,
              out=None, x_halo=None, plan=None, relu_x=False, gate=None, add2=None,
              add2_scale=0.0):
    _check_plan(plan, rowptr32)
    if relu_x:
        x = torch.relu(x)
    if x_halo is not None:
        x = torch.cat([x, x_halo], dim=0)
    y = cp.spmm(_np(rowptr32).astype(np.int64), _np(col32).astype(np.int64), _np(val),
                _np(x), reduce)
    y = torch.from_numpy(y) * scale
    if add is not None:
        y = y + add_scale * add[:y.size(0)]
    if add2 is not None:
        y = y + add2_scale * add2[:y.size(0)]
    if gate is not None:
        y = torch.where(gate[:y.size(0)] > 0, y, torch.zeros_like(y))
    if out is not None:
        out.copy_(y)
        return out
    return y


def _spmm_csr_incagg(rowptr32, col32, val, x, m_in, m_ag, reduce="sum", scale=1.0, add=None,
                     add_scale=0.0, out=None, plan=None):
    _check_plan(plan, rowptr32)
    b, w = rowptr32.numel() - 1, x.size(1)
    h = cp.incagg(_np(rowptr32).astype(np.int64), _np(col32).astype(np.int64), _np(val),
                  _np(x[:b]), _np(m_in[:b, :w]), _np(m_ag[:b, :w]), reduce)
    h = torch.from_numpy(h) * scale
    if add is not None:
        h = h + add_scale * add[:b]
    return h


def _rows_sub(a, b, relu_a=False):
    return (torch.relu(a) if relu_a else a) - b[:a.size(0), :a.size(1)]


def _gather_rows(emb, n_id, out=None, device=None):
    rows = emb.index_select(0, n_id.cpu())
    if out is not None:
        out[:rows.size(0), :rows.size(1)] = rows
        return out
    return rows


def _scatter_rows(emb, n_id, x):
    emb[n_id.cpu(), :x.size(1)] = x


def _copy_row_chunks(dst, src, offset, count, packed_is_src, device=None, relu=False):
    if relu:
        src = torch.relu(src)
    s = 0
    w = src.size(1)
    for o, c in zip(offset.tolist(), count.tolist()):
        if packed_is_src:
            dst[o:o + c, :w] = src[s:s + c]
            dst[o:o + c, w:] = 0
        else:
            dst[s:s + c, :w] = src[o:o + c]
            dst[s:s + c, w:] = 0
        s += c


@contextlib.contextmanager
def cpu_kernels():
    from torch_geometric_autoscale import ops, sparse, history
    saved = {k: getattr(ops, k) for k in ("spmm_csr", "spmm_csr_incagg", "rows_sub", "gather_rows",
                                          "scatter_rows", "copy_row_chunks")}
    saved_csr = sparse.SparseTensor.device_csr
    saved_csr_t = sparse.SparseTensor.device_csr_t
    saved_gpu_path = history.History._gpu_path

    def device_csr(self):
        v = self._value
        return self._rowptr32(), self._col32(), None if v is None else v.to(torch.float32)

    ops.spmm_csr, ops.spmm_csr_incagg, ops.rows_sub = _spmm_csr, _spmm_csr_incagg, _rows_sub
    ops.gather_rows, ops.scatter_rows, ops.copy_row_chunks = _gather_rows, _scatter_rows, _copy_row_chunks
    sparse.SparseTensor.device_csr = device_csr
    history.History._gpu_path = lambda self, other_is_cuda=False: False
    try:
        yield
    finally:
        for k, v in saved.items():
            setattr(ops, k, v)
        sparse.SparseTensor.device_csr = saved_csr
        sparse.SparseTensor.device_csr_t = saved_csr_t
        history.History._gpu_path = saved_gpu_path
