"""Differentiable neighbour aggregation on top of the HIP kernels.

``spmm``            y = scale * reduce(A, x) + add_scale * add
``incagg_aggregate``  h = scale * (reduce(A_ib, x - M_in) + M_ag) + add_scale * add
                    (the incremental aggregate of models/gcn2.py:255 and its siblings)

The backward pass of both is dX = scale * A^T dY: the same row-parallel kernel run on
the transposed CSR cached on the adjacency, so no atomics are involved
(autograd through every aggregation call site, main.py:81).
"""
from typing import Optional

import torch
from torch import Tensor

from . import ops
from .sparse import SparseTensor


def _as_rows(x: Tensor) -> Tensor:
    if x.dim() != 2:
        raise ValueError(f"expected a [rows, width] matrix, got {tuple(x.shape)}")
    if x.dtype != torch.float32:
        raise TypeError(f"aggregation computes in float32, got {x.dtype}")
    if x.size(1) > 1 and x.stride(1) != 1:
        x = x.contiguous()
    return x


def _aligned_rows(x: Tensor) -> Tensor:
    """``x`` with rows that start 16-byte aligned (a copy at the padded pitch of ops.row_pitch
    when they do not, e.g. a contiguous [rows, 41] matrix): the feature-sliced kernel needs
    aligned rows, and at such widths it is 1.5-2x the whole-row kernel on large batches (reddit-
    sized batch at width 41, MI355X: 390 us against 216-250 us per launch) -- far more than the
    copy costs."""
    if (not x.is_cuda or x.size(1) < 8 or x.size(0) < 4096
            or (x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0)):
        return x
    out = ops.rows_empty(x.size(0), x.size(1), x.device, x.dtype)
    if out.stride(0) % 4 != 0:
        return x
    return out.copy_(x)


class _SpMM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, x_halo, add, adj, reduce, scale, add_scale, halo_ids, relu_x=False,
                gate_grad=False, grad_out=None, out=None):
        rowptr, col, val = adj.device_csr()
        n_cols = adj.size(1)
        if halo_ids is not None:
            # out-of-batch column c reads row halo_ids[c - B] of the table x_halo in place
            col = adj.col_in_place(halo_ids, x.size(0))
            n_cols = x.size(0) + x_halo.size(0)
            plan = adj.plan_edges_in_place(halo_ids, x.size(0))
        else:
            plan = adj.plan_edges()
        y = ops.spmm_csr(rowptr, col, val, x, n_cols, reduce, scale, add, add_scale, out=out,
                         x_halo=x_halo, plan=plan, relu_x=relu_x)
        # relu_x: x holds pre-activations and this call consumes relu(x); gate_grad: x IS the
        # output of a relu whose backward the caller leaves to this call.  Either way the
        # gradient that leaves here is already zeroed where x <= 0.
        ctx.gated = relu_x or gate_grad
        if ctx.gated:
            ctx.save_for_backward(x)
        ctx.adj, ctx.reduce, ctx.scale, ctx.add_scale = adj, reduce, scale, add_scale
        ctx.x_rows = x.size(0)
        ctx.halo_rows = None if x_halo is None else x_halo.size(0)
        ctx.in_place = halo_ids is not None
        ctx.has_add = add is not None
        ctx.grad_out = grad_out
        return y

    @staticmethod
    def backward(ctx, dy):
        dx = dhalo = dadd = None
        dy = _as_rows(dy)
        n_cols = ctx.adj.size(1)
        gate = ctx.saved_tensors[0] if ctx.gated else None
        if ctx.halo_rows is None:
            if ctx.needs_input_grad[0]:
                dx = _transposed_aggregate(ctx.adj, dy, ctx.reduce, ctx.scale, 0, n_cols,
                                           ctx.x_rows, gate, ctx.grad_out)
        else:
            # only the rows somebody needs: the pulled History rows are constants
            if ctx.needs_input_grad[0]:
                dx = _transposed_aggregate(ctx.adj, dy, ctx.reduce, ctx.scale, 0, ctx.x_rows,
                                           ctx.x_rows, gate, ctx.grad_out)
            if ctx.needs_input_grad[1]:
                if ctx.in_place:
                    raise RuntimeError("rows consumed in place from a History table are "
                                       "constants: the table must not require a gradient")
                dhalo = _transposed_aggregate(ctx.adj, dy, ctx.reduce, ctx.scale, ctx.x_rows,
                                              n_cols, ctx.halo_rows)
        if ctx.has_add and ctx.needs_input_grad[2]:
            dadd = dy if ctx.add_scale == 1.0 else dy * ctx.add_scale
        return dx, dhalo, dadd, None, None, None, None, None, None, None, None, None


class _IncAgg(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, add, m_in, m_ag, adj, reduce, scale, add_scale):
        rowptr, col, val = adj.device_csr()
        h = ops.spmm_csr_incagg(rowptr, col, val, x, m_in, m_ag, reduce, scale, add, add_scale,
                                plan=adj.plan_edges())
        ctx.adj, ctx.reduce, ctx.scale, ctx.add_scale = adj, reduce, scale, add_scale
        ctx.x_rows = x.size(0)
        ctx.has_add = add is not None
        return h

    @staticmethod
    def backward(ctx, dy):
        dx = dadd = None
        dy = _as_rows(dy)
        if ctx.needs_input_grad[0]:
            dx = _transposed_aggregate(ctx.adj, dy, ctx.reduce, ctx.scale, 0, ctx.adj.size(1),
                                       ctx.x_rows)
        if ctx.has_add and ctx.needs_input_grad[1]:
            dadd = dy if ctx.add_scale == 1.0 else dy * ctx.add_scale
        return dx, dadd, None, None, None, None, None, None


def _transposed_aggregate(adj: SparseTensor, dy: Tensor, reduce: str, scale: float, lo: int,
                          hi: int, out_rows: int, gate: Optional[Tensor] = None,
                          out: Optional[Tensor] = None) -> Tensor:
    """Rows [lo, hi) of  scale * A^T dY  (A^T has one row per source row of A), padded
    with zero rows up to ``out_rows``; ``gate`` (rows [lo, ...) of it): zeroed where <= 0.
    ``out``: where the caller wants the result (used when its shape is the result's)."""
    t_rowptr, t_col, t_val = adj.device_csr_t(reduce)
    hi = min(hi, adj.size(1))
    dy = _aligned_rows(dy)
    if gate is not None:
        gate = gate[lo:] if lo else gate
    if out is not None and (out_rows != hi - lo or tuple(out.shape) != (out_rows, dy.size(1))
                            or out.dtype != dy.dtype or out.device != dy.device):
        out = None
    # a window of row pointers addresses the same col / val arrays: no copy
    dx = ops.spmm_csr(t_rowptr[lo:hi + 1], t_col, t_val, dy, adj.size(0), "sum", scale,
                      plan=adj.plan_t(reduce, lo, hi), gate=gate, out=out)
    if out_rows > hi - lo:  # x had rows the adjacency never references
        pad = torch.zeros((out_rows - (hi - lo), dy.size(1)), dtype=dy.dtype, device=dy.device)
        dx = torch.cat([dx, pad], dim=0)
    return dx


def spmm(adj: SparseTensor, x: Tensor, reduce: str = "sum", scale: float = 1.0,
         add: Optional[Tensor] = None, add_scale: float = 0.0,
         x_halo: Optional[Tensor] = None, halo_ids: Optional[Tensor] = None,
         relu_x: bool = False, gate_grad: bool = False,
         grad_out: Optional[Tensor] = None, out: Optional[Tensor] = None) -> Tensor:
    """``adj @ x`` / ``spmm(adj, x, reduce)`` with an optional fused ``+ add_scale * add``.

    ``x_halo``: the source rows are ``[x ; x_halo]`` without concatenating them -- ``x``
    the fresh in-batch rows, ``x_halo`` the pulled out-of-batch History rows.  The
    backward pass then aggregates only over the columns of ``x`` (and of ``x_halo`` if
    it requires a gradient, which pulled rows never do).

    ``halo_ids`` (device int64, one per out-of-batch column): ``x_halo`` is then a whole
    History table and column ``B + k`` reads its row ``halo_ids[k]`` in place -- the pull
    (history.py:33-39) folded into the aggregation, no gathered copy is ever written.

    ``relu_x``: ``x`` holds pre-activations; the call aggregates ``relu(x)`` (``x_halo`` rows
    are taken as they are) and its backward returns the gradient w.r.t. the pre-activations.
    ``gate_grad``: ``x`` is the output of a relu whose backward is folded into this call's
    backward (the gradient returned is already zeroed where ``x <= 0``).
    ``grad_out``: a [rows of x, width] buffer (rows may be strided) the backward pass writes the
    gradient w.r.t. ``x`` into, instead of a matrix of its own.
    ``out`` (outside autograd only): the [rows of adj, width] matrix the result is written into,
    e.g. the batch's rows of a History table of aggregates."""
    if out is not None and torch.is_grad_enabled() and (
            x.requires_grad or (add is not None and add.requires_grad)):
        raise RuntimeError("out= is for calls outside autograd")
    if reduce not in ("sum", "add", "mean"):
        raise ValueError(f"unsupported reduce '{reduce}'")
    x = _aligned_rows(_as_rows(x))
    if add is not None:
        add = _as_rows(add)
    if x_halo is not None:
        x_halo = _as_rows(x_halo)
    if halo_ids is not None and x_halo is None:
        raise ValueError("halo_ids needs the table to read from (x_halo)")
    return _SpMM.apply(x, x_halo, add, adj, "mean" if reduce == "mean" else "sum", float(scale),
                       float(add_scale), halo_ids, bool(relu_x), bool(gate_grad), grad_out, out)


def incagg_aggregate(adj: SparseTensor, x: Tensor, m_in: Tensor, m_ag: Tensor,
                     reduce: str = "sum", scale: float = 1.0, add: Optional[Tensor] = None,
                     add_scale: float = 0.0, fused: bool = True, relu_x: bool = False,
                     gate_grad: bool = False, grad_out: Optional[Tensor] = None) -> Tensor:
    """``adj @ (x - m_in) + m_ag``; ``m_in`` / ``m_ag`` are constants (the reference detaches
    both, models/gcn2.py:246-247).  ``fused``: one launch gathering both operands; otherwise a
    streaming subtract and the plain aggregation with ``m_ag`` riding in its epilogue (gathers
    touch one operand instead of two).

    ``relu_x`` / ``gate_grad`` / ``grad_out`` as in :func:`spmm` (two-pass form only): ``x``
    holds pre-activations and ``relu(x) - m_in`` is aggregated / ``x`` is the output of a relu
    whose backward this call's backward takes over / where the backward pass leaves the gradient
    w.r.t. ``x``."""
    if adj.size(0) != adj.size(1):
        raise ValueError("the incremental aggregate needs the square in-batch adjacency")
    x = _as_rows(x)
    b = adj.size(0)
    if x.size(0) < b:
        raise ValueError("x has fewer rows than the batch")
    m_in, m_ag = m_in.detach(), m_ag.detach()
    if add is not None:
        add = _as_rows(add)
    red = "mean" if reduce == "mean" else "sum"
    if fused:
        if relu_x or gate_grad or grad_out is not None:
            raise ValueError("relu_x / gate_grad / grad_out need the two-pass form (fused=False)")
        return _IncAgg.apply(x, add, m_in, m_ag, adj, red, float(scale), float(add_scale))
    return _IncAggRows.apply(x, add, m_in, m_ag, adj, red, float(scale), float(add_scale),
                             bool(relu_x), bool(gate_grad), grad_out)


class _IncAggRows(torch.autograd.Function):
    """``scale * (reduce(A_ib, f(x) - M_in) + M_ag) + add_scale * add`` as a streaming subtract
    (ops.rows_sub) followed by ONE plain aggregation launch with ``M_ag`` and ``add`` as its two
    epilogue terms; ``f`` = relu with ``relu_x``.  Backward: the transposed aggregation, its
    result zeroed where ``x <= 0`` when the relu is this call's (``relu_x``) or the caller's to
    be folded in (``gate_grad``)."""

    @staticmethod
    def forward(ctx, x, add, m_in, m_ag, adj, reduce, scale, add_scale, relu_x, gate_grad,
                grad_out):
        b, w = adj.size(0), x.size(1)
        d = ops.rows_sub(x if x.size(0) == b else x[:b], m_in, relu_a=relu_x)
        rowptr, col, val = adj.device_csr()
        h = ops.spmm_csr(rowptr, col, val, d, adj.size(1), reduce, scale, m_ag[:b, :w], scale,
                         plan=adj.plan_edges(), add2=add, add2_scale=add_scale)
        ctx.gated = relu_x or gate_grad
        if ctx.gated:
            ctx.save_for_backward(x)
        ctx.adj, ctx.reduce, ctx.scale, ctx.add_scale = adj, reduce, scale, add_scale
        ctx.x_rows, ctx.grad_out, ctx.has_add = x.size(0), grad_out, add is not None
        return h

    @staticmethod
    def backward(ctx, dy):
        dx = dadd = None
        dy = _as_rows(dy)
        if ctx.needs_input_grad[0]:
            gate = ctx.saved_tensors[0] if ctx.gated else None
            dx = _transposed_aggregate(ctx.adj, dy, ctx.reduce, ctx.scale, 0, ctx.adj.size(1),
                                       ctx.x_rows, gate, ctx.grad_out)
        if ctx.has_add and ctx.needs_input_grad[1]:
            dadd = dy if ctx.add_scale == 1.0 else dy * ctx.add_scale
        return (dx, dadd) + (None,) * 9
