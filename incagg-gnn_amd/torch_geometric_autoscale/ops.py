"""Tensor-level entry points of the hot path.

Every function here validates its tensors, then calls the C ABI
(``include/incagg_gnn.h``) with raw pointers, sizes and the current HIP stream.
PyTorch only supplies device memory and streams.  The five operators of the
reference's native extension are registered under the same names in
``torch.ops.torch_geometric_autoscale`` (csrc/async.cpp:44-48, csrc/relabel.cpp:40-42).
"""
import collections
import ctypes
from typing import Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from . import _lib
from ._lib import REDUCE, check, lib


def _ptr(t: Optional[Tensor]):
    return None if t is None or t.numel() == 0 else t.data_ptr()


class _on_device:
    """``with torch.cuda.device(device)`` that costs nothing when ``device`` is already the
    current one (every call of a training step; the context manager's enter / exit pair is a
    measurable part of an entry point's host cost)."""
    __slots__ = ("_ctx",)

    def __init__(self, device):
        idx = device.index
        self._ctx = None if idx is None or idx == torch.cuda.current_device() \
            else torch.cuda.device(device)

    def __enter__(self):
        if self._ctx is not None:
            self._ctx.__enter__()

    def __exit__(self, *exc):
        if self._ctx is not None:
            self._ctx.__exit__(*exc)
        return False


def _stream_of(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _rows_f32(t: Tensor, name: str, device=None) -> None:
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    if t.dim() != 2:
        raise ValueError(f"{name} must be 2-D [rows, width], got {tuple(t.shape)}")
    if t.size(1) > 1 and t.stride(1) != 1:
        raise ValueError(f"{name} must be row-major (unit stride along the width)")
    if device is not None and t.device != device:
        raise ValueError(f"{name} is on {t.device}, expected {device}")


def _ld(t: Tensor) -> int:
    return t.stride(0) if t.size(0) > 1 else max(t.size(1), 1)


def row_pitch(width: int) -> int:
    """Floats between the starts of two rows of a matrix the aggregation gathers from.  The
    16-byte-aligned kernel pins every 128-byte feature slice to one XCD, so an XCD only touches
    addresses congruent to ``slice * 128`` modulo the pitch; with a pitch that is an even number
    of 128-byte lines (256 floats = 8 lines, worst: a power of two) those addresses share their
    low bits and crowd into a fraction of the L2's channels and sets.  One line of padding makes
    the line count odd: measured on MI355X, products batch at width 256, forward 95.6 -> 81.2 us,
    transposed window 52.8 -> 44.2 us.  Widths that are not a multiple of 4 floats are rounded
    up to a multiple of 16: rows then start 16-byte aligned, which is what the feature-sliced
    kernel needs (width 41, same batch: 108 -> 75 us against the whole-row kernel)."""
    if width >= 8 and width % 4 != 0:
        width = (width + 15) // 16 * 16
    if width < 64 or width % 32 != 0 or (width // 32) % 2 == 1:
        return width
    return width + 32


def rows_empty(rows: int, width: int, device, dtype=torch.float32, with_base: bool = False):
    """Uninitialised ``[rows, width]`` matrix with the row pitch of :func:`row_pitch` (a view
    of a wider allocation when the pitch differs from the width).  ``with_base``: also the
    contiguous ``[rows, pitch]`` matrix it is a view of."""
    pitch = row_pitch(width) if torch.device(device).type == "cuda" else width
    if not with_base and pitch != width and rows > 0:
        # one allocation call instead of allocate + slice + re-wrap (this function runs ~20 times
        # per eagerly issued step); the storage ends with the last row's last element -- no
        # kernel touches the padding behind a row
        return torch.empty_strided((rows, width), (pitch, 1), dtype=dtype, device=device)
    buf = torch.empty((rows, pitch), dtype=dtype, device=device)
    # (.data: same memory, but not a view in autograd's books -- results of custom autograd
    # functions that are views may not be updated in place, e.g. by a following relu_)
    out = buf if pitch == width else buf[:, :width].data
    return (out, buf) if with_base else out


def _need_cuda(t: Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"{name} must live on the GPU: the aggregation and History-row kernels are HIP "
            "only and there is no CPU implementation")


# --------------------------------------------------------------------------- aggregation
class SpmmPlan:
    """Work plan of one row-pointer array (``incagg_spmm_plan``): the rows by descending
    length and the (row, segment) items of the long ones.  A pure function of the row pointers,
    built once per adjacency at batch construction -- on the host for host-built batches
    (then uploaded), on the GPU for device-built ones -- and passed to every aggregation call
    over those row pointers.  The arrays are never replaced once built: recorded HIP graphs
    hold their addresses.

    A plan on the GPU with rows cut into several segments also owns their ``combine`` buffer
    (where the segments of such a row meet; all zero between launches).  It makes the plan
    stateful: launches that use one plan must not overlap in time (one stream, or ordered
    streams); ``combine = None`` falls back to zeroing those rows with a launch in front."""

    def __init__(self, order: Tensor, items: Tensor, multi_rows: Tensor, n_hub: int,
                 n_items: int, n_multi: int, hub_threshold: int, segment_edges: int):
        self.order, self.items, self.multi_rows = order, items, multi_rows
        self.n_rows = int(order.numel())
        self.n_hub, self.n_items, self.n_multi = int(n_hub), int(n_items), int(n_multi)
        self.hub_threshold, self.segment_edges = int(hub_threshold), int(segment_edges)
        self.edges = None      # packed (col, val) pairs of the calls it is used with
        self.combine = None
        if order.is_cuda and self.n_multi > 0:
            self.combine = torch.zeros(self.n_multi * _lib.COMBINE_SLOT_FLOATS,
                                       dtype=torch.float32, device=order.device)
        self._struct = None

    @property
    def device(self):
        return self.order.device

    def to(self, device, non_blocking: bool = False) -> "SpmmPlan":
        device = torch.device(device)
        if device == self.order.device:
            return self
        return SpmmPlan(self.order.to(device, non_blocking=non_blocking),
                        self.items.to(device, non_blocking=non_blocking),
                        self.multi_rows.to(device, non_blocking=non_blocking), self.n_hub,
                        self.n_items, self.n_multi, self.hub_threshold, self.segment_edges)

    def struct(self) -> "_lib.SpmmPlanStruct":
        if self._struct is None:
            self._struct = _lib.SpmmPlanStruct(
                _ptr(self.order), _ptr(self.items), _ptr(self.multi_rows), self.n_rows,
                self.n_hub, self.n_items, self.n_multi, self.hub_threshold, self.segment_edges,
                _ptr(self.combine), 0 if self.combine is None else self.combine.numel(),
                _ptr(self.edges))
        return self._struct

    def with_edges(self, edges: Optional[Tensor]) -> "SpmmPlan":
        """The same plan (same arrays, same combine buffer) for calls whose edge list is also
        available packed (:func:`pack_edges` of the call's ``col`` / ``val``)."""
        twin = SpmmPlan.__new__(SpmmPlan)
        twin.__dict__.update(self.__dict__)
        twin.edges, twin._struct = edges, None
        if edges is not None and (edges.dtype != torch.int32 or edges.dim() != 2
                                  or edges.size(1) != 2 or not edges.is_contiguous()
                                  or edges.device != self.order.device):
            raise ValueError("edges must be a contiguous [nnz, 2] int32 tensor on the plan's device")
        return twin

    def without_combine(self) -> "SpmmPlan":
        """The same plan (same arrays) without the combine buffer."""
        twin = SpmmPlan(self.order, self.items, self.multi_rows, self.n_hub, self.n_items,
                        self.n_multi, self.hub_threshold, self.segment_edges)
        twin.combine, twin.edges = None, self.edges
        return twin

    @staticmethod
    def build(rowptr32: Tensor, nnz: Optional[int] = None, hub_threshold: int = 0,
              segment_edges: int = 0) -> "SpmmPlan":
        """Plan of ``rowptr32`` (int32; host or device, also a window of a longer array).
        ``nnz``: an upper bound on the edges the rows address (sizes the item list without
        reading device memory); default: read from the row pointers."""
        if rowptr32.dtype != torch.int32 or rowptr32.dim() != 1 or rowptr32.numel() < 1:
            raise TypeError("rowptr must be a 1-D int32 tensor with n_rows + 1 entries")
        rowptr32 = rowptr32.contiguous()
        n_rows = rowptr32.numel() - 1
        if nnz is None:
            nnz = int(rowptr32[-1] - rowptr32[0]) if n_rows else 0
        L = lib()
        cap = L.incagg_spmm_plan_items_bound(n_rows, int(nnz), hub_threshold, segment_edges)
        check(cap)
        dev = rowptr32.device
        order = torch.empty(n_rows, dtype=torch.int32, device=dev)
        items = torch.empty((max(cap, 1), 2), dtype=torch.int32, device=dev)
        multi = torch.empty(max(min(n_rows, cap), 1), dtype=torch.int32, device=dev)
        counts = (ctypes.c_int64 * 5)()
        if dev.type == "cuda":
            with _on_device(dev):
                need = L.incagg_spmm_plan_device_scratch_bytes(n_rows)
                check(need)
                _, _, scratch = _build_workspace(dev, 0, need)
                check(L.incagg_spmm_plan_device(
                    rowptr32.data_ptr(), n_rows, hub_threshold, segment_edges,
                    scratch.data_ptr(), scratch.numel(), _ptr(order), items.data_ptr(), cap,
                    multi.data_ptr(), multi.numel(), ctypes.addressof(counts), _stream_of(dev)))
        else:
            check(L.incagg_spmm_plan_host(
                rowptr32.data_ptr(), n_rows, hub_threshold, segment_edges, _ptr(order),
                items.data_ptr(), cap, multi.data_ptr(), multi.numel(),
                ctypes.addressof(counts)))
        n_hub, n_items, n_multi, thr, seg = (int(c) for c in counts)
        # keep only what is used (one copy at batch construction, never again)
        return SpmmPlan(order, items[:max(n_items, 1)].clone(), multi[:max(n_multi, 1)].clone(),
                        n_hub, n_items, n_multi, thr, seg)


def pack_edges(col32: Tensor, val: Optional[Tensor], out: Optional[Tensor] = None) -> Tensor:
    """``[nnz, 2]`` int32: ``(col[e], bit pattern of val[e])`` (of 1.0 without values)."""
    if out is None:
        out = torch.empty((col32.numel(), 2), dtype=torch.int32, device=col32.device)
    out[:, 0] = col32
    if val is None:
        out[:, 1] = 0x3F800000
    else:
        out[:, 1] = val.view(torch.int32)
    return out


def _plan_ref(plan: Optional[SpmmPlan], n_rows: int, dev):
    if plan is None:
        return None
    if plan.n_rows != n_rows:
        raise ValueError(f"the plan was built for {plan.n_rows} rows, the call has {n_rows}")
    if plan.device != dev:
        raise ValueError(f"the plan is on {plan.device}, the call runs on {dev}")
    return ctypes.addressof(plan.struct())


def _check_edges(plan: Optional[SpmmPlan], col32: Tensor) -> None:
    if plan is not None and plan.edges is not None and plan.edges.size(0) != col32.numel():
        raise ValueError("the plan's packed edge list does not have one entry per edge")


def spmm_csr(rowptr32: Tensor, col32: Tensor, val: Optional[Tensor], x: Tensor, n_cols: int,
             reduce: str = "sum", scale: float = 1.0, add: Optional[Tensor] = None,
             add_scale: float = 0.0, out: Optional[Tensor] = None,
             x_halo: Optional[Tensor] = None, plan: Optional[SpmmPlan] = None,
             relu_x: bool = False, gate: Optional[Tensor] = None,
             add2: Optional[Tensor] = None, add2_scale: float = 0.0) -> Tensor:
    """y = scale * reduce_rows(A, [x ; x_halo]) + add_scale * add [+ add2_scale * add2]
    (incagg_spmm_csr_f32).

    With ``x_halo`` the source rows ``[x.size(0), n_cols)`` are read from ``x_halo`` -- the
    concatenation is never materialised.  ``plan``: :class:`SpmmPlan` of ``rowptr32``.
    ``relu_x``: rows of ``x`` (not of ``x_halo``) pass through ``max(0, .)`` as they are read;
    ``gate`` ([n_rows, width]): the result is zeroed wherever ``gate <= 0``."""
    _need_cuda(x, "x")
    dev = x.device
    _rows_f32(x, "x")
    n_rows = rowptr32.numel() - 1
    if rowptr32.dtype != torch.int32 or col32.dtype != torch.int32:
        raise TypeError("device CSR must be int32")
    if rowptr32.device != dev or col32.device != dev or (val is not None and val.device != dev):
        raise ValueError("CSR and x must be on the same device")
    if not (rowptr32.is_contiguous() and col32.is_contiguous()):
        raise ValueError("device CSR arrays must be contiguous")
    if val is not None and (val.dtype != torch.float32 or val.numel() != col32.numel()
                            or not val.is_contiguous()):
        raise ValueError("val must be contiguous float32 with one entry per edge")
    width = x.size(1)
    n_split = n_cols
    if x_halo is not None:
        _rows_f32(x_halo, "x_halo", dev)
        if x_halo.size(1) != width:
            raise ValueError("x and x_halo differ in width")
        n_split = x.size(0)
        if n_split > n_cols or n_split + x_halo.size(0) < n_cols:
            raise ValueError(f"x ({x.size(0)} rows) + x_halo ({x_halo.size(0)} rows) do not cover "
                             f"the {n_cols} columns of the adjacency")
    elif x.size(0) < n_cols:
        raise ValueError(f"x has {x.size(0)} rows but the adjacency has {n_cols} columns")
    if add is not None:
        _rows_f32(add, "add", dev)
        if add.size(0) < n_rows or add.size(1) != width:
            raise ValueError("add must be [>= n_rows, width]")
    if out is None:
        out = rows_empty(n_rows, width, dev)
    else:
        _rows_f32(out, "out", dev)
        if out.size(0) != n_rows or out.size(1) != width:
            raise ValueError("out has the wrong shape")
    _check_edges(plan, col32)
    fuse = None
    if relu_x or gate is not None or add2 is not None:
        for t, name in ((gate, "gate"), (add2, "add2")):
            if t is not None:
                _rows_f32(t, name, dev)
                if t.size(0) < n_rows or t.size(1) != width:
                    raise ValueError(f"{name} must be [>= n_rows, width]")
        fuse_struct = _lib.SpmmFuseStruct(1 if relu_x else 0, _ptr(gate),
                                          0 if gate is None else _ld(gate), _ptr(add2),
                                          0 if add2 is None else _ld(add2), float(add2_scale))
        fuse = ctypes.addressof(fuse_struct)
    with _on_device(dev):
        check(lib().incagg_spmm_csr_f32(
            _ptr(rowptr32), _ptr(col32), _ptr(val), _ptr(x), _ld(x), _ptr(x_halo),
            0 if x_halo is None else _ld(x_halo), n_split, n_rows, n_cols, width,
            REDUCE[reduce], float(scale), _ptr(add), 0 if add is None else _ld(add),
            float(add_scale), int(col32.numel()), _plan_ref(plan, n_rows, dev), fuse, _ptr(out),
            _ld(out), _stream_of(dev)))
    return out


def spmm_csr_incagg(rowptr32: Tensor, col32: Tensor, val: Optional[Tensor], x: Tensor,
                    m_in: Tensor, m_ag: Tensor, reduce: str = "sum", scale: float = 1.0,
                    add: Optional[Tensor] = None, add_scale: float = 0.0,
                    out: Optional[Tensor] = None, plan: Optional[SpmmPlan] = None) -> Tensor:
    """h = scale * (reduce_rows(A, x - m_in) + m_ag) + add_scale * add."""
    _need_cuda(x, "x")
    dev = x.device
    n_rows = rowptr32.numel() - 1
    width = x.size(1)
    for t, name in ((x, "x"), (m_in, "m_in"), (m_ag, "m_ag")):
        _rows_f32(t, name, dev)
        if t.size(0) < n_rows or t.size(1) < width:
            raise ValueError(f"{name} must be at least [{n_rows}, {width}], got {tuple(t.shape)}")
    if rowptr32.dtype != torch.int32 or col32.dtype != torch.int32:
        raise TypeError("device CSR must be int32")
    if not (rowptr32.is_contiguous() and col32.is_contiguous()):
        raise ValueError("device CSR arrays must be contiguous")
    if add is not None:
        _rows_f32(add, "add", dev)
        if add.size(0) < n_rows or add.size(1) != width:
            raise ValueError("add must be [>= n_rows, width]")
    if out is None:
        out = rows_empty(n_rows, width, dev)
    _check_edges(plan, col32)
    with _on_device(dev):
        check(lib().incagg_spmm_csr_incagg_f32(
            _ptr(rowptr32), _ptr(col32), _ptr(val), _ptr(x), _ld(x), _ptr(m_in), _ld(m_in),
            _ptr(m_ag), _ld(m_ag), n_rows, width, REDUCE[reduce], float(scale), _ptr(add),
            0 if add is None else _ld(add), float(add_scale), int(col32.numel()),
            _plan_ref(plan, n_rows, dev), _ptr(out), _ld(out), _stream_of(dev)))
    return out


def rows_sub(a: Tensor, b: Tensor, relu_a: bool = False) -> Tensor:
    """``a - b[:rows, :width]`` (``max(0, a) - b`` with ``relu_a``), written with the row pitch
    the aggregation wants for a matrix it gathers from (incagg_rows_sub_relu_f32)."""
    _need_cuda(a, "a")
    _rows_f32(a, "a")
    _rows_f32(b, "b", a.device)
    n, w = a.size(0), a.size(1)
    if b.size(0) < n or b.size(1) < w:
        raise ValueError("b is smaller than a")
    out = rows_empty(n, w, a.device)
    with _on_device(a.device):
        check(lib().incagg_rows_sub_relu_f32(_ptr(a), _ld(a), _ptr(b), _ld(b), n, w,
                                             1 if relu_a else 0, _ptr(out), _ld(out),
                                             _stream_of(a.device)))
    return out


def tn_finish(parts: Tensor, a_tail: Optional[Tensor], b_tail: Optional[Tensor],
              out: Optional[Tensor] = None) -> Tensor:
    """``parts.sum(0) + a_tail^T @ b_tail`` (incagg_tn_finish_f32): the slab sum of a batched
    weight-gradient product and its few remainder rows in one launch; into ``out`` (contiguous
    ``[m, n]``) when given."""
    _need_cuda(parts, "parts")
    if parts.dim() != 3 or parts.dtype != torch.float32 or not parts.is_contiguous():
        raise ValueError("parts must be a contiguous [slabs, m, n] float32 tensor")
    s_, m, n = parts.shape
    r = 0
    if a_tail is not None and a_tail.size(0) > 0:
        _rows_f32(a_tail, "a_tail", parts.device)
        _rows_f32(b_tail, "b_tail", parts.device)
        r = a_tail.size(0)
        if b_tail.size(0) != r or a_tail.size(1) != m or b_tail.size(1) != n:
            raise ValueError("remainder rows must be [r, m] and [r, n]")
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=parts.device)
    elif (tuple(out.shape) != (m, n) or out.dtype != torch.float32 or not out.is_contiguous()
          or out.device != parts.device):
        raise ValueError("out must be a contiguous [m, n] float32 tensor on the device of parts")
    with _on_device(parts.device):
        check(lib().incagg_tn_finish_f32(
            _ptr(parts), s_, _ptr(a_tail) if r else None, _ld(a_tail) if r else 0,
            _ptr(b_tail) if r else None, _ld(b_tail) if r else 0, r, m, n, _ptr(out), n,
            _stream_of(parts.device)))
    return out


def tn_finish_batched(parts, tails, out: Tensor) -> Tensor:
    """``out[l] = parts[l].sum(0) + a_l^T @ b_l`` for all l in one launch
    (incagg_tn_finish_batched_f32).  ``parts``: L contiguous ``[slabs, m, n]`` tensors;
    ``tails``: L pairs ``(a_l [r, m], b_l [r, n])`` of remainder rows (or L times ``None``);
    ``out``: contiguous ``[L, m, n]``."""
    n_batch = len(parts)
    if n_batch == 0:
        return out
    _need_cuda(parts[0], "parts")
    dev = parts[0].device
    s_, m, n = parts[0].shape
    for p in parts:
        if p.shape != (s_, m, n) or p.dtype != torch.float32 or not p.is_contiguous() \
                or p.device != dev:
            raise ValueError("parts must be contiguous [slabs, m, n] float32 tensors of one shape")
    if (tuple(out.shape) != (n_batch, m, n) or out.dtype != torch.float32
            or not out.is_contiguous() or out.device != dev):
        raise ValueError("out must be a contiguous [L, m, n] float32 tensor on the device of parts")
    if len(tails) != n_batch:
        raise ValueError("one pair of remainder rows per product")
    r = 0 if tails[0] is None else tails[0][0].size(0)
    p_ptr = (ctypes.c_void_p * n_batch)(*[_ptr(p) for p in parts])
    a_ptr, b_ptr = (ctypes.c_void_p * n_batch)(), (ctypes.c_void_p * n_batch)()
    lda, ldb = (ctypes.c_int64 * n_batch)(), (ctypes.c_int64 * n_batch)()
    for l, pair in enumerate(tails):
        if r == 0:
            if pair is not None and pair[0].size(0) != 0:
                raise ValueError("all products must have the same number of remainder rows")
            continue
        if pair is None:
            raise ValueError("all products must have the same number of remainder rows")
        a, b = pair
        _rows_f32(a, "a_tail", dev)
        _rows_f32(b, "b_tail", dev)
        if a.shape != (r, m) or b.shape != (r, n):
            raise ValueError("remainder rows must be [r, m] and [r, n], r the same everywhere")
        a_ptr[l], b_ptr[l], lda[l], ldb[l] = _ptr(a), _ptr(b), _ld(a), _ld(b)
    with _on_device(dev):
        check(lib().incagg_tn_finish_batched_f32(
            ctypes.addressof(p_ptr), n_batch, s_, ctypes.addressof(a_ptr), ctypes.addressof(lda),
            ctypes.addressof(b_ptr), ctypes.addressof(ldb), r, m, n, _ptr(out),
            _stream_of(dev)))
    return out


def rows_gemm_supported(k: int, n: int) -> bool:
    """Whether :func:`rows_gemm` takes a ``[rows, k] x [k, n]`` product."""
    return bool(lib().incagg_rows_gemm_supported(int(k), int(n)))


def rows_gemm(x: Tensor, w: Tensor, trans_w: bool = False, out: Optional[Tensor] = None,
              accumulate: bool = False, table: Optional[Tensor] = None, table_row0: int = 0,
              relu_push: bool = False, bias: Optional[Tensor] = None,
              table_only: bool = False) -> Tensor:
    """``out = (out if accumulate else 0) + bias + x @ w`` (``trans_w``: ``x @ w.T``) for a tall ``x``
    ``[rows, k]`` and a small weight, on the matrix cores in exact fp32
    (incagg_rows_gemm_f32); ``table``: rows ``[table_row0, table_row0 + rows)`` of it receive
    the result too (``relu_push``: ``max(0, result)``) -- the History push of a layer's output,
    done by the producer; ``table_only``: ``out`` is then not written (it still supplies the
    accumulated-into values).  ``out`` defaults to a fresh matrix at the gather pitch."""
    _need_cuda(x, "x")
    _rows_f32(x, "x")
    if w.dim() != 2 or w.dtype != torch.float32 or w.device != x.device or w.stride(1) != 1:
        raise ValueError("w must be an fp32 matrix with unit column stride on the device of x")
    k = x.size(1)
    n = w.size(0) if trans_w else w.size(1)
    if (w.size(1) if trans_w else w.size(0)) != k:
        raise ValueError(f"shapes do not match: x is {tuple(x.shape)}, w {tuple(w.shape)}"
                         f"{' (transposed)' if trans_w else ''}")
    if out is None:
        if accumulate:
            raise ValueError("accumulate needs an out matrix")
        out = rows_empty(x.size(0), n, x.device)
    else:
        _rows_f32(out, "out", x.device)
        if out.size(0) != x.size(0) or out.size(1) != n:
            raise ValueError("out has the wrong shape")
    if bias is not None and (bias.dtype != torch.float32 or bias.device != x.device
                             or bias.dim() != 1 or bias.numel() != n or not bias.is_contiguous()):
        raise ValueError("bias must be a contiguous fp32 vector with one entry per output column")
    t_ptr = t_ld = t_rows = 0
    if table is not None:
        _rows_f32(table, "table", x.device)
        if table.size(1) < n:
            raise ValueError("table rows are narrower than the result")
        t_ptr, t_ld, t_rows = table.data_ptr(), _ld(table), table.size(0)
    with _on_device(x.device):
        check(lib().incagg_rows_gemm_f32(
            _ptr(x), _ld(x), _ptr(w), w.stride(0), 1 if trans_w else 0, _ptr(bias), x.size(0), k, n,
            1 if accumulate else 0, _ptr(out), _ld(out), t_ptr or None, t_ld, t_rows,
            int(table_row0), 1 if relu_push else 0, 1 if table_only and table is not None else 0,
            _stream_of(x.device)))
    return out


def relu_rows_(x: Tensor) -> Tensor:
    """``x.relu_()`` over a [rows, width] matrix at streaming rate (incagg_relu_rows_f32).
    Outside autograd."""
    _need_cuda(x, "x")
    _rows_f32(x, "x")
    with _on_device(x.device):
        check(lib().incagg_relu_rows_f32(_ptr(x), _ld(x), x.size(0), x.size(1),
                                         _stream_of(x.device)))
    return x


def rows_sum(srcs, scale: float = 1.0) -> Tensor:
    """``scale * (srcs[0] + srcs[1] + ...)`` over up to 16 [rows, width] matrices of any row
    pitch in one launch (incagg_rows_sum_f32); the result has the pitch the aggregation wants
    for its operands (ops.row_pitch)."""
    srcs = list(srcs)
    if not 1 <= len(srcs) <= 16:
        raise ValueError("1 to 16 sources")
    first = srcs[0]
    _need_cuda(first, "srcs[0]")
    for i, t in enumerate(srcs):
        _rows_f32(t, f"srcs[{i}]", first.device)
        if t.shape != first.shape:
            raise ValueError("sources differ in shape")
    rows, width = first.shape
    out = rows_empty(rows, width, first.device)
    ptrs = (ctypes.c_void_p * len(srcs))(*[t.data_ptr() for t in srcs])
    lds = (ctypes.c_int64 * len(srcs))(*[_ld(t) for t in srcs])
    with _on_device(first.device):
        check(lib().incagg_rows_sum_f32(ctypes.addressof(ptrs), ctypes.addressof(lds), len(srcs),
                                        float(scale), rows, width, _ptr(out), _ld(out),
                                        _stream_of(first.device)))
    return out


def gate_add_rows_(dst: Tensor, src: Tensor, gate: Tensor) -> Tensor:
    """``dst += where(gate > 0, src, 0)`` in one launch (incagg_gate_add_rows_f32)."""
    _need_cuda(dst, "dst")
    for t, name in ((dst, "dst"), (src, "src"), (gate, "gate")):
        _rows_f32(t, name, dst.device)
    if src.shape != dst.shape or gate.shape != dst.shape:
        raise ValueError("dst, src and gate must have one shape")
    with _on_device(dst.device):
        check(lib().incagg_gate_add_rows_f32(_ptr(dst), _ld(dst), _ptr(src), _ld(src), _ptr(gate),
                                             _ld(gate), dst.size(0), dst.size(1),
                                             _stream_of(dst.device)))
    return dst


def gate_rows(src: Tensor, gate: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """``where(gate > 0, src, 0)`` in one launch (incagg_gate_rows_f32): the backward of a relu
    whose output (or input) is ``gate``.  ``out``: where to write, ``src`` itself allowed;
    default a fresh matrix at the gather pitch."""
    _need_cuda(src, "src")
    if out is None:
        out = rows_empty(src.size(0), src.size(1), src.device)
    for t, name in ((out, "out"), (src, "src"), (gate, "gate")):
        _rows_f32(t, name, src.device)
    if src.shape != out.shape or gate.shape != out.shape:
        raise ValueError("out, src and gate must have one shape")
    with _on_device(src.device):
        check(lib().incagg_gate_rows_f32(_ptr(out), _ld(out), _ptr(src), _ld(src), _ptr(gate),
                                         _ld(gate), out.size(0), out.size(1),
                                         _stream_of(src.device)))
    return out


def _fold_check(n_layers: int, c: int, betas) -> "ctypes.Array":
    if len(betas) != n_layers:
        raise ValueError("one beta per layer")
    return (ctypes.c_float * n_layers)(*[float(b) for b in betas])


def gcn2_fold_weights(ws: Tensor, alpha: float, betas) -> Tuple[Tensor, Tensor]:
    """GCNII's folded weights of all layers in one launch (incagg_gcn2_fold_weights_f32):
    ``ws`` = ``[W1_0 .. W1_{L-1}, W2_0 .. W2_{L-1}]`` ([2L, C, C]) ->
    ``w1f[l] = (1-alpha)(beta_l W1_l + (1-beta_l) I)`` ([L, C, C]) and ``w2cat`` ([C, L*C]) whose
    column block l is ``alpha (beta_l W2_l + (1-beta_l) I)``."""
    _need_cuda(ws, "ws")
    if ws.dim() != 3 or ws.size(0) % 2 or ws.size(1) != ws.size(2) or ws.dtype != torch.float32 \
            or not ws.is_contiguous():
        raise ValueError("ws must be a contiguous [2L, C, C] float32 tensor")
    n_layers, c = ws.size(0) // 2, ws.size(1)
    beta = _fold_check(n_layers, c, betas)
    w1f = torch.empty((n_layers, c, c), dtype=torch.float32, device=ws.device)
    w2cat = torch.empty((c, n_layers * c), dtype=torch.float32, device=ws.device)
    with _on_device(ws.device):
        check(lib().incagg_gcn2_fold_weights_f32(_ptr(ws), n_layers, c, float(alpha),
                                                 ctypes.addressof(beta), _ptr(w1f), _ptr(w2cat),
                                                 _stream_of(ws.device)))
    return w1f, w2cat


def gcn2_fold_weights_backward(dw1f: Tensor, dw2cat: Tensor, alpha: float, betas) -> Tensor:
    """Gradient of ``ws`` from the gradients of :func:`gcn2_fold_weights`' results."""
    _need_cuda(dw1f, "dw1f")
    n_layers, c = dw1f.size(0), dw1f.size(1)
    if dw1f.shape != (n_layers, c, c) or dw2cat.shape != (c, n_layers * c) \
            or dw1f.dtype != torch.float32 or dw2cat.dtype != torch.float32 \
            or not dw1f.is_contiguous() or not dw2cat.is_contiguous() \
            or dw2cat.device != dw1f.device:
        raise ValueError("gradients must be contiguous float32 [L, C, C] and [C, L*C]")
    beta = _fold_check(n_layers, c, betas)
    dws = torch.empty((2 * n_layers, c, c), dtype=torch.float32, device=dw1f.device)
    with _on_device(dw1f.device):
        check(lib().incagg_gcn2_fold_weights_backward_f32(
            _ptr(dw1f), _ptr(dw2cat), n_layers, c, float(alpha), ctypes.addressof(beta),
            _ptr(dws), _stream_of(dw1f.device)))
    return dws


def column_sums(x: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """``x.sum(0)`` of a tall ``[rows, width]`` matrix (incagg_column_sums_f32): bias gradients.
    Two launches, fixed summation order."""
    _need_cuda(x, "x")
    _rows_f32(x, "x")
    rows, width = x.shape
    if out is None:
        out = torch.empty(width, dtype=torch.float32, device=x.device)
    elif out.shape != (width,) or out.dtype != torch.float32 or not out.is_contiguous() \
            or out.device != x.device:
        raise ValueError("out must be a contiguous [width] float32 tensor on the device of x")
    L = lib()
    need = L.incagg_column_sums_ws_floats(rows, width)
    check(need)
    ws = torch.empty(max(int(need), 1), dtype=torch.float32, device=x.device)
    with _on_device(x.device):
        check(L.incagg_column_sums_f32(_ptr(x), _ld(x), rows, width, _ptr(ws), ws.numel(),
                                       _ptr(out), _stream_of(x.device)))
    return out


def batch_norm_supported(x: Tensor) -> bool:
    """Whether :func:`batch_norm_forward` takes ``x``: a float32 device matrix of at least two
    rows whose width is a multiple of 4 and whose rows start 16-byte aligned."""
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.size(1) > 0
            and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0
            and bool(lib().incagg_batch_norm_supported(x.size(0), x.size(1))))


def _bn_workspace(x: Tensor) -> Tensor:
    need = lib().incagg_batch_norm_ws_floats(x.size(0), x.size(1))
    check(need)
    return torch.empty(max(int(need), 4), dtype=torch.float32, device=x.device)


def batch_norm_forward(x: Tensor, weight: Optional[Tensor], bias: Optional[Tensor],
                       running_mean: Optional[Tensor], running_var: Optional[Tensor],
                       momentum: float, eps: float) -> Tuple[Tensor, Tensor, Tensor]:
    """Training-mode batch normalisation over the rows of ``x`` (incagg_batch_norm_forward_f32):
    ``(y, mean, rstd)``; the running statistics are updated in place.  ``y`` has the row pitch
    the aggregation kernels want."""
    _need_cuda(x, "x")
    if not batch_norm_supported(x):
        raise ValueError("batch_norm_forward needs a float32 [rows >= 2, width % 4 == 0] matrix "
                         "with 16-byte aligned rows")
    rows, width = x.shape
    for t, name in ((weight, "weight"), (bias, "bias"), (running_mean, "running_mean"),
                    (running_var, "running_var")):
        if t is not None and (t.shape != (width,) or t.dtype != torch.float32
                              or t.device != x.device or not t.is_contiguous()):
            raise ValueError(f"{name} must be a contiguous [width] float32 tensor on the device of x")
    y = rows_empty(rows, width, x.device)
    stats = torch.empty((2, -(-width // 4) * 4), dtype=torch.float32, device=x.device)
    ws = _bn_workspace(x)
    with _on_device(x.device):
        check(lib().incagg_batch_norm_forward_f32(
            _ptr(x), _ld(x), rows, width, _ptr(weight), _ptr(bias), float(eps), float(momentum),
            _ptr(running_mean), _ptr(running_var), _ptr(ws), ws.numel(), _ptr(stats[0]),
            _ptr(stats[1]), _ptr(y), _ld(y), _stream_of(x.device)))
    return y, stats[0], stats[1]


def batch_norm_backward(dy: Tensor, x: Tensor, weight: Optional[Tensor], mean: Tensor, rstd: Tensor,
                        dx: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor]:
    """``(dx, dweight, dbias)`` of :func:`batch_norm_forward` (incagg_batch_norm_backward_f32).
    ``dx``: where to write (may be ``dy`` itself); default a fresh matrix at the gather pitch."""
    _need_cuda(dy, "dy")
    rows, width = x.shape
    if dy.shape != x.shape or not batch_norm_supported(dy) or not batch_norm_supported(x):
        raise ValueError("dy and x must be float32 [rows, width] matrices with 16-byte aligned rows")
    if dx is None:
        dx = rows_empty(rows, width, x.device)
    elif dx.shape != x.shape or not batch_norm_supported(dx):
        raise ValueError("dx must be a float32 [rows, width] matrix with 16-byte aligned rows")
    grads = torch.empty((2, -(-width // 4) * 4), dtype=torch.float32, device=x.device)
    ws = _bn_workspace(x)
    with _on_device(x.device):
        check(lib().incagg_batch_norm_backward_f32(
            _ptr(dy), _ld(dy), _ptr(x), _ld(x), rows, width, _ptr(weight), _ptr(mean), _ptr(rstd),
            _ptr(ws), ws.numel(), _ptr(grads[0]), _ptr(grads[1]), _ptr(dx), _ld(dx),
            _stream_of(x.device)))
    return dx, grads[0], grads[1]


class _SoftmaxXent(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels):
        rows, classes = logits.shape
        L = lib()
        ws = torch.empty(int(L.incagg_softmax_xent_ws_floats(rows)), dtype=torch.float32,
                         device=logits.device)
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        with _on_device(logits.device):
            check(L.incagg_softmax_xent_forward_f32(
                _ptr(logits), _ld(logits), _ptr(labels), rows, classes, _ptr(ws), ws.numel(),
                _ptr(loss), _stream_of(logits.device)))
        ctx.save_for_backward(logits, labels, ws)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        logits, labels, ws = ctx.saved_tensors
        rows, classes = logits.shape
        dloss = dloss.contiguous()
        # laid out like the logits: where those are rows of a wider allocation (the padded,
        # 16-byte-aligned row pitch of an aggregation's output), so is their gradient -- it is
        # the input of that aggregation's backward launch, which takes the feature-sliced kernel
        # only for aligned rows
        pitch = logits.stride(0) if rows > 1 and logits.stride(0) >= classes else classes
        dlogits = torch.empty((rows, pitch), dtype=torch.float32, device=logits.device)
        if pitch != classes:
            dlogits = dlogits[:, :classes]
        with _on_device(logits.device):
            check(lib().incagg_softmax_xent_backward_f32(
                _ptr(logits), _ld(logits), _ptr(labels), rows, classes, _ptr(ws), _ptr(dloss),
                _ptr(dlogits), pitch, _stream_of(logits.device)))
        return dlogits, None


def softmax_cross_entropy(logits: Tensor, labels: Tensor) -> Tensor:
    """Mean softmax cross entropy of ``logits`` [rows, classes] against int64 ``labels`` [rows]
    -- ``torch.nn.CrossEntropyLoss()`` with its defaults (main.py:60,78) -- as two launches
    forward and one backward (incagg_softmax_xent_*_f32).  Labels outside ``[0, classes)`` are
    reported by ``check_device_errors`` -- torch's ``ignore_index`` (-100) included: rows to be
    left out of the loss are selected away before the call, as the training loop does with its
    mask (main.py:80)."""
    _need_cuda(logits, "logits")
    _rows_f32(logits, "logits")
    if labels.dtype != torch.int64 or labels.dim() != 1 or labels.numel() != logits.size(0) \
            or labels.device != logits.device or not labels.is_contiguous():
        raise ValueError("labels must be a contiguous int64 vector with one entry per row, on "
                         "the device of the logits")
    if logits.size(0) < 1:
        raise ValueError("no rows")
    return _SoftmaxXent.apply(logits, labels)


# --------------------------------------------------------------------------- History rows
def _device_visible(t: Tensor) -> bool:
    return t.is_cuda or t.is_pinned()


def gather_rows(emb: Tensor, n_id: Tensor, out: Optional[Tensor] = None,
                device=None) -> Tensor:
    """out[k] = emb[n_id[k]].  ``emb`` is HBM or pinned host memory; ``n_id`` on the GPU."""
    _need_cuda(n_id, "n_id")
    dev = n_id.device if device is None else torch.device(device)
    _rows_f32(emb, "emb")
    if not _device_visible(emb):
        raise RuntimeError("emb must be on the GPU or in pinned host memory")
    if emb.is_cuda and emb.device != dev:
        raise ValueError("emb and n_id are on different GPUs")
    if n_id.dtype != torch.int64 or n_id.dim() != 1 or not n_id.is_contiguous():
        raise TypeError("n_id must be a contiguous 1-D int64 tensor")
    k, w = n_id.numel(), emb.size(1)
    if out is None:
        out = rows_empty(k, w, dev)
    else:
        _rows_f32(out, "out", dev)
        if out.size(0) < k or out.size(1) < w:
            raise ValueError("out too small")
    with _on_device(dev):
        check(lib().incagg_gather_rows_f32(_ptr(emb), _ld(emb), emb.size(0), _ptr(n_id), k, w,
                                           _ptr(out), _ld(out), _stream_of(dev)))
    return out


def scatter_rows(emb: Tensor, n_id: Tensor, x: Tensor) -> None:
    """emb[n_id[k], :x.size(1)] = x[k]."""
    _need_cuda(n_id, "n_id")
    dev = n_id.device
    _rows_f32(emb, "emb")
    _rows_f32(x, "x", dev)
    if not _device_visible(emb):
        raise RuntimeError("emb must be on the GPU or in pinned host memory")
    if n_id.dtype != torch.int64 or n_id.dim() != 1 or not n_id.is_contiguous():
        raise TypeError("n_id must be a contiguous 1-D int64 tensor")
    if x.size(0) != n_id.numel() or x.size(1) > emb.size(1):
        raise ValueError("x must be [len(n_id), <= emb width]")
    with _on_device(dev):
        check(lib().incagg_scatter_rows_f32(_ptr(emb), _ld(emb), emb.size(0), _ptr(n_id),
                                            n_id.numel(), x.size(1), _ptr(x), _ld(x),
                                            _stream_of(dev)))


def _host_i64(t: Tensor, name: str) -> Tensor:
    if t.is_cuda:
        raise RuntimeError(f"{name} tensor must be a CPU tensor")
    if t.dim() != 1:
        raise RuntimeError(f"{name} tensor must be one-dimensional")
    return t.to(torch.int64).contiguous()


def merge_row_chunks(offset: Tensor, count: Tensor) -> Tuple[Tensor, Tensor]:
    """The same rows in the same order as the chunks ``(offset[i], count[i])`` describe, in as
    few chunks as possible: a chunk that starts where its predecessor ends is joined to it,
    empty chunks are dropped.  The partitions of a batch are often neighbours in the table
    (consecutive partition ids, e.g. every evaluation batch and every batch of an unshuffled
    pass): their rows then form ONE range, which is copied by one launch instead of one per 32
    chunks -- or not copied at all where a view of the table will do."""
    offset, count = _host_i64(offset, "Offset"), _host_i64(count, "Count")
    if offset.numel() != count.numel():
        raise RuntimeError("Size mismatch")
    if offset.numel() <= 1 and bool((count > 0).all()):
        return offset, count
    o, c = offset.numpy(), count.numpy()
    if (c < 0).any():   # refused here: joined to a neighbour it would pass for a shorter chunk
        raise RuntimeError("Invalid index: negative row count")
    keep = c != 0
    o, c = o[keep], c[keep]
    if o.size > 1:
        starts = np.concatenate(([0], np.flatnonzero(o[1:] != o[:-1] + c[:-1]) + 1))
        o, c = o[starts], np.add.reduceat(c, starts)
    return torch.from_numpy(np.ascontiguousarray(o)), torch.from_numpy(np.ascontiguousarray(c))


def copy_row_chunks(dst: Tensor, src: Tensor, offset: Tensor, count: Tensor,
                    packed_is_src: bool, device=None, relu: bool = False) -> None:
    """Chunked row copy with zero padding (incagg_copy_row_chunks_f32); ``relu``: the copied
    values pass through ``max(0, .)``.  Chunks that adjoin are copied as one."""
    _rows_f32(dst, "dst")
    _rows_f32(src, "src")
    if not (_device_visible(dst) and _device_visible(src)):
        raise RuntimeError("both sides must be on the GPU or in pinned host memory")
    dev = device or (dst.device if dst.is_cuda else src.device)
    dev = torch.device(dev)
    if dev.type != "cuda":
        raise RuntimeError("at least one side must be on the GPU")
    offset, count = merge_row_chunks(offset, count)
    if offset.numel() == 0:
        return
    with _on_device(dev):
        check(lib().incagg_copy_row_chunks_f32(
            _ptr(dst), _ld(dst), dst.size(0), dst.size(1), _ptr(src), _ld(src), src.size(0),
            src.size(1), _ptr(offset), _ptr(count), offset.numel(), 1 if packed_is_src else 0,
            1 if relu else 0, _stream_of(dev)))


# --------------------------------------------------------------------------- batch construction
def _relabel(fn, rowptr: Tensor, col: Tensor, value: Optional[Tensor], idx: Tensor,
             bipartite: bool) -> Tuple[Tensor, Tensor, Optional[Tensor], Tensor]:
    if rowptr.is_cuda:
        raise RuntimeError("No GPU version supported: relabelling runs on the host")
    if col.is_cuda:
        raise RuntimeError("Col tensor must be a CPU tensor")
    if idx.is_cuda:
        raise RuntimeError("Index tensor must be a CPU tensor")
    if value is not None:
        if value.is_cuda:
            raise RuntimeError("Value tensor must be a CPU tensor")
        if value.dim() != 1:
            raise RuntimeError("Value tensor must be one-dimensional")
        value = value.contiguous()
    for t, name in ((rowptr, "rowptr"), (col, "col"), (idx, "idx")):
        if t.dtype != torch.int64:
            raise RuntimeError(f"{name} must be int64")
    rowptr, col, idx = rowptr.contiguous(), col.contiguous(), idx.contiguous()
    n_nodes, n_idx = rowptr.numel() - 1, idx.numel()
    ub = lib().incagg_relabel_bound(_ptr(rowptr), n_nodes, _ptr(idx), n_idx)
    check(ub)
    out_rowptr = torch.empty(2 * n_idx + 1 + ub, dtype=torch.int64)
    out_col = torch.empty(ub, dtype=torch.int64)
    out_n_id = torch.empty(n_idx + ub, dtype=torch.int64)
    out_value = None if value is None else torch.empty(ub, dtype=value.dtype)
    counts = torch.zeros(3, dtype=torch.int64)
    check(fn(_ptr(rowptr), n_nodes, _ptr(col), _ptr(value),
             0 if value is None else value.element_size(), _ptr(idx), n_idx, 1 if bipartite else 0,
             out_rowptr.data_ptr(), _ptr(out_col), _ptr(out_value), _ptr(out_n_id),
             counts.data_ptr()))
    nnz, n_total, rp_len = counts.tolist()
    return (out_rowptr[:rp_len].clone(), out_col[:nnz].clone(),
            None if out_value is None else out_value[:nnz].clone(), out_n_id[:n_total].clone())


def relabel_one_hop(rowptr, col, value, idx, bipartite):
    """csrc/relabel.cpp:10-24."""
    return _relabel(lib().incagg_relabel_one_hop, rowptr, col, value, idx, bipartite)


def relabel_one_hop_within_batch(rowptr, col, value, idx, bipartite):
    """csrc/relabel.cpp:27-38."""
    return _relabel(lib().incagg_relabel_one_hop_within_batch, rowptr, col, value, idx, bipartite)


def csr_transpose(rowptr: Tensor, col: Tensor, val: Optional[Tensor], n_rows: int, n_cols: int,
                  reduce: str = "sum") -> Tuple[Tensor, Tensor, Tensor]:
    """Host CSR (int64) -> transposed device-format CSR (int32, fp32 values), on the CPU."""
    rowptr = rowptr.to("cpu", torch.int64).contiguous()
    col = col.to("cpu", torch.int64).contiguous()
    val = None if val is None else val.to("cpu", torch.float32).contiguous()
    nnz = col.numel()
    t_rowptr = torch.empty(n_cols + 1, dtype=torch.int32)
    t_col = torch.empty(nnz, dtype=torch.int32)
    t_val = torch.empty(nnz, dtype=torch.float32)
    check(lib().incagg_csr_transpose(_ptr(rowptr), _ptr(col), _ptr(val), n_rows, n_cols,
                                     REDUCE[reduce], t_rowptr.data_ptr(), _ptr(t_col), _ptr(t_val)))
    return t_rowptr, t_col, t_val


def partition_bfs(rowptr: Tensor, col: Tensor, num_parts: int, seeds: Tensor) -> Tensor:
    """Part id per node by breadth-first block growing (host; see include/incagg_gnn.h)."""
    rowptr, col = _host_i64(rowptr, "rowptr"), _host_i64(col, "col")
    seeds = _host_i64(seeds, "seeds")
    n = rowptr.numel() - 1
    if seeds.numel() != n:
        raise ValueError("seeds must list every node once")
    cluster = torch.empty(n, dtype=torch.int64)
    check(lib().incagg_partition_bfs(_ptr(rowptr) or rowptr.data_ptr(), _ptr(col), n,
                                     int(num_parts), _ptr(seeds), _ptr(cluster)))
    return cluster


def partition_refine(rowptr: Tensor, col: Tensor, num_parts: int, cluster: Tensor,
                     max_part_size: int, sweeps: int = 4) -> Tuple[Tensor, int]:
    """Boundary refinement of a partition (host); returns the new part ids and the moves made."""
    rowptr, col = _host_i64(rowptr, "rowptr"), _host_i64(col, "col")
    cluster = _host_i64(cluster, "cluster").clone()
    n = rowptr.numel() - 1
    if cluster.numel() != n:
        raise ValueError("cluster must have one entry per node")
    moves = ctypes.c_int64(0)
    check(lib().incagg_partition_refine(_ptr(rowptr) or rowptr.data_ptr(), _ptr(col), n,
                                        int(num_parts), int(max_part_size), int(sweeps),
                                        _ptr(cluster), ctypes.addressof(moves)))
    return cluster, int(moves.value)


# --------------------------------------------------------------------------- batch construction on the GPU
_build_ws = {}   # device -> {"maps": (a, b), "scratch": uint8 tensor}


def _build_workspace(dev, n_nodes: int, scratch_bytes: int):
    ws = _build_ws.setdefault(dev, {"maps": None, "scratch": None})
    maps = ws["maps"]
    if maps is None or maps[0].numel() < n_nodes:
        # dense id tables, all -1 between calls (every call restores the entries it touched)
        ws["maps"] = maps = tuple(torch.full((n_nodes,), -1, dtype=torch.int32, device=dev)
                                  for _ in range(2))
    if ws["scratch"] is None or ws["scratch"].numel() < scratch_bytes:
        ws["scratch"] = torch.empty(int(scratch_bytes * 1.25) + 256, dtype=torch.uint8, device=dev)
    return maps[0], maps[1], ws["scratch"]


def relabel_one_hop_device(rowptr: Tensor, col: Tensor, value: Optional[Tensor], idx: Tensor,
                           within_batch: bool = False, edge_capacity: Optional[int] = None
                           ) -> Tuple[Tensor, Tensor, Optional[Tensor], Tensor, int]:
    """The relabelling of csrc/cpu/relabel_cpu.cpp:3-108 (``within_batch``: :111-214) on the
    GPU, from a device-resident global CSR (int64).  Returns the bipartite device-format batch
    ``(rowptr int32, col int32, value fp32 | None, n_id int64, distinct ids in idx)``, element
    for element what the host operators return.  Waits for the stream (sizes come back)."""
    _need_cuda(rowptr, "rowptr")
    dev = rowptr.device
    for t, name in ((rowptr, "rowptr"), (col, "col"), (idx, "idx")):
        if t.dtype != torch.int64 or t.device != dev or not t.is_contiguous():
            raise RuntimeError(f"{name} must be a contiguous int64 tensor on {dev}")
    if value is not None and (value.dtype != torch.float32 or value.device != dev
                              or value.dim() != 1 or not value.is_contiguous()):
        raise RuntimeError("Value tensor must be a contiguous one-dimensional float32 device tensor")
    n_nodes, n_idx = rowptr.numel() - 1, idx.numel()
    L = lib()
    with _on_device(dev):
        stream = _stream_of(dev)
        if edge_capacity is None:
            need = 4096 + 16 * (n_idx + 2)
            _, _, scratch = _build_workspace(dev, n_nodes, need)
            bound = ctypes.c_int64(0)
            check(L.incagg_relabel_bound_device(_ptr(rowptr), n_nodes, _ptr(idx), n_idx,
                                                scratch.data_ptr(), scratch.numel(),
                                                ctypes.addressof(bound), stream))
            edge_capacity = int(bound.value)
        need = L.incagg_relabel_device_scratch_bytes(n_idx, edge_capacity)
        check(need)
        map_a, map_b, scratch = _build_workspace(dev, n_nodes, need)
        node_capacity = n_idx if within_batch else n_idx + edge_capacity
        out_rowptr = torch.empty(n_idx + 1, dtype=torch.int32, device=dev)
        out_col = torch.empty(edge_capacity, dtype=torch.int32, device=dev)
        out_val = None if value is None else torch.empty(edge_capacity, dtype=torch.float32, device=dev)
        out_n_id = torch.empty(node_capacity, dtype=torch.int64, device=dev)
        counts = torch.zeros(3, dtype=torch.int64)
        check(L.incagg_relabel_one_hop_device(
            _ptr(rowptr), n_nodes, _ptr(col), _ptr(value), _ptr(idx), n_idx,
            1 if within_batch else 0, edge_capacity, node_capacity, map_a.data_ptr(),
            map_b.data_ptr(), scratch.data_ptr(), scratch.numel(), out_rowptr.data_ptr(),
            _ptr(out_col), _ptr(out_val), _ptr(out_n_id), counts.data_ptr(), stream))
    nnz, n_total, distinct = counts.tolist()
    return (out_rowptr, out_col[:nnz], None if out_val is None else out_val[:nnz],
            out_n_id[:n_total], distinct)


def csr_transpose_device(rowptr32: Tensor, col32: Tensor, val: Optional[Tensor], n_rows: int,
                         n_cols: int, reduce: str = "sum") -> Tuple[Tensor, Tensor, Tensor]:
    """:func:`csr_transpose` on the GPU: device-format CSR in, transposed device-format CSR out
    (same element order as the host routine).  Does not synchronise."""
    _need_cuda(rowptr32, "rowptr")
    dev = rowptr32.device
    if rowptr32.dtype != torch.int32 or col32.dtype != torch.int32:
        raise TypeError("device CSR must be int32")
    if rowptr32.numel() != n_rows + 1:
        raise ValueError("rowptr does not have n_rows + 1 entries")
    rowptr32, col32 = rowptr32.contiguous(), col32.contiguous()
    val = None if val is None else val.to(torch.float32).contiguous()
    nnz = col32.numel()
    L = lib()
    need = L.incagg_csr_transpose_device_scratch_bytes(n_cols, nnz)
    check(need)
    t_rowptr = torch.empty(n_cols + 1, dtype=torch.int32, device=dev)
    t_col = torch.empty(nnz, dtype=torch.int32, device=dev)
    t_val = torch.empty(nnz, dtype=torch.float32, device=dev)
    with _on_device(dev):
        _, _, scratch = _build_workspace(dev, 0, need)
        check(L.incagg_csr_transpose_device(
            _ptr(rowptr32) or rowptr32.data_ptr(), _ptr(col32), _ptr(val), n_rows, n_cols, nnz,
            REDUCE[reduce], scratch.data_ptr(), scratch.numel(), t_rowptr.data_ptr(), _ptr(t_col),
            _ptr(t_val), _stream_of(dev)))
    return t_rowptr, t_col, t_val


# --------------------------------------------------------------------------- staging
# Tensors of queued reads are held here until the matching synchronize(), which is
# what capturing them by value in the worker lambda does in the reference
# (csrc/cuda/async_cuda.cu:63).
_inflight_reads = collections.deque()


def _non_default_stream(device) -> int:
    cur = torch.cuda.current_stream(device)
    if cur == torch.cuda.default_stream(device):
        raise RuntimeError("Asynchronous read/write requires a non-default stream")
    return cur.cuda_stream


def read_async(src: Tensor, offset: Optional[Tensor], count: Optional[Tensor], index: Tensor,
               dst: Tensor, buffer: Tensor) -> None:
    """csrc/async.cpp:21-32 / csrc/cuda/async_cuda.cu:14-115."""
    if src.is_cuda:
        raise RuntimeError("Source tensor must be a CPU tensor")
    if index.is_cuda:
        raise RuntimeError("Index tensor must be a CPU tensor")
    if not dst.is_cuda:
        raise RuntimeError("Target tensor must be a GPU tensor")
    if buffer.is_cuda:
        raise RuntimeError("Buffer tensor must be a CPU tensor")
    if not buffer.is_pinned():
        raise RuntimeError("Buffer tensor must be pinned")
    for t, name in ((src, "Source"), (dst, "Target"), (buffer, "Buffer")):
        if not t.is_contiguous():
            raise RuntimeError(f"{name} tensor must be contiguous")
    if index.dim() != 1:
        raise RuntimeError("Index tensor must be one-dimensional")
    n_chunks = 0
    if offset is not None:
        if not src.is_pinned():
            raise RuntimeError("Source tensor must be pinned")
        if count is None:
            raise RuntimeError("Count tensor is undefined")
        offset, count = _host_i64(offset, "Offset"), _host_i64(count, "Count")
        if offset.numel() != count.numel():
            raise RuntimeError("Size mismatch")
        n_chunks = offset.numel()
    if index.numel() and index.dtype != torch.int64:
        raise RuntimeError("Index tensor must be int64")
    index = index.contiguous()
    if src.dtype != dst.dtype or src.dtype != buffer.dtype:
        raise RuntimeError("Source, target and buffer must share one dtype")
    row_elems = src.numel() // max(src.size(0), 1)
    if dst.numel() // max(dst.size(0), 1) != row_elems or \
            buffer.numel() // max(buffer.size(0), 1) != row_elems:
        raise RuntimeError("Source, target and buffer rows must have the same size")
    stream = _non_default_stream(dst.device)
    with _on_device(dst.device):
        check(lib().incagg_stage_read_async(
            _ptr(src), src.size(0), row_elems * src.element_size(), _ptr(offset), _ptr(count),
            n_chunks, _ptr(index), index.numel(), _ptr(dst), dst.size(0), _ptr(buffer),
            buffer.size(0), stream))
    _inflight_reads.append((src, offset, count, index, dst, buffer))


def synchronize() -> None:
    """csrc/async.cpp:13-19: join exactly one queued read, oldest first."""
    try:
        check(lib().incagg_stage_synchronize())
    finally:
        if _inflight_reads:
            _inflight_reads.popleft()


def write_async(src: Tensor, offset: Tensor, count: Tensor, dst: Tensor) -> None:
    """csrc/async.cpp:34-41 / csrc/cuda/async_cuda.cu:117-165."""
    if not src.is_cuda:
        raise RuntimeError("Source tensor must be a GPU tensor")
    if dst.is_cuda:
        raise RuntimeError("Target tensor must be a CPU tensor")
    if not dst.is_pinned():
        raise RuntimeError("Target tensor must be pinned")
    if not src.is_contiguous():
        raise RuntimeError("Source tensor must be contiguous")
    if not dst.is_contiguous():
        raise RuntimeError("Target tensor must be contiguous")
    offset, count = _host_i64(offset, "Offset"), _host_i64(count, "Count")
    if offset.numel() != count.numel():
        raise RuntimeError("Size mismatch")
    if src.dtype != dst.dtype:
        raise RuntimeError("Source and target must share one dtype")
    row_elems = src.numel() // max(src.size(0), 1)
    if dst.numel() // max(dst.size(0), 1) != row_elems:
        raise RuntimeError("Source and target rows must have the same size")
    stream = _non_default_stream(src.device)
    with _on_device(src.device):
        check(lib().incagg_stage_write_async(
            _ptr(src), src.size(0), row_elems * src.element_size(), _ptr(offset), _ptr(count),
            offset.numel(), _ptr(dst), dst.size(0), stream))


# --------------------------------------------------------------------------- torch.ops registration
_registered = None


def torch_ops_defined() -> bool:
    """True when the five operators already exist in this process (operator objects loaded by
    the reference's own ``__init__``, or an earlier registration)."""
    ns = torch.ops.torch_geometric_autoscale
    try:
        return all(hasattr(ns, n) for n in ("synchronize", "read_async", "write_async",
                                            "relabel_one_hop", "relabel_one_hop_within_batch"))
    except RuntimeError:
        return False


def register_torch_ops() -> bool:
    """Defines torch.ops.torch_geometric_autoscale.{synchronize, read_async, write_async,
    relabel_one_hop, relabel_one_hop_within_batch} with the reference's schemas."""
    global _registered
    if _registered is not None or torch_ops_defined():
        return True
    try:
        reg = torch.library.Library("torch_geometric_autoscale", "DEF")
    except RuntimeError:
        return False  # namespace already defined by another library in this process
    relabel_schema = ("(Tensor rowptr, Tensor col, Tensor? value, Tensor idx, bool bipartite)"
                      " -> (Tensor, Tensor, Tensor?, Tensor)")
    reg.define("synchronize() -> ()")
    reg.define("read_async(Tensor src, Tensor? offset, Tensor? count, Tensor index, Tensor dst, "
               "Tensor buffer) -> ()")
    reg.define("write_async(Tensor src, Tensor offset, Tensor count, Tensor dst) -> ()")
    reg.define("relabel_one_hop" + relabel_schema)
    reg.define("relabel_one_hop_within_batch" + relabel_schema)
    key = "CompositeExplicitAutograd"
    reg.impl("synchronize", synchronize, key)
    reg.impl("read_async", read_async, key)
    reg.impl("write_async", write_async, key)
    reg.impl("relabel_one_hop", relabel_one_hop, key)
    reg.impl("relabel_one_hop_within_batch", relabel_one_hop_within_batch, key)
    _registered = reg
    return True


__all__ = [
    "spmm_csr", "spmm_csr_incagg", "rows_sub", "gather_rows", "scatter_rows", "copy_row_chunks",
    "relabel_one_hop", "relabel_one_hop_within_batch", "csr_transpose", "read_async",
    "write_async", "synchronize", "register_torch_ops",
]
