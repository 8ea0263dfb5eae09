"""The three convolution modules the reference takes from PyG, rebuilt on the HIP
aggregation kernels.  Parameter names follow PyG (``lin.weight``, ``bias``,
``weight1``, ``weight2``, ``lin_l``, ``lin_r``) so that state dicts interchange.

Every module is split into ``aggregate`` (the sparse part, a HIP kernel) and
``update`` (the dense part, hipBLASLt through torch) because the history-based
forward passes need to substitute their own aggregate:
``h = A_ib (x - M_in) + M_ag`` (models/gcn.py:241-244, models/gcn2.py:255-257,
models/graphsage.py:634-642).
"""
import math
from typing import Optional

import torch
from torch import Tensor
from torch.nn import Linear, Parameter

from . import ops
from .aggregate import spmm
from .sparse import SparseTensor


def _mm_rows(a: Tensor, b: Tensor) -> Tensor:
    """``a @ b`` written into a matrix with the row pitch the aggregation kernels want for
    their sources (ops.row_pitch): layer outputs and input gradients are what the next
    aggregation gathers from."""
    if not a.is_cuda or a.dim() != 2:
        return a @ b
    return torch.mm(a, b, out=ops.rows_empty(a.size(0), b.size(1), a.device, a.dtype))


def _slab_count(rows: int) -> int:
    """Number of row slabs for :func:`_tn_product`.  hipBLASLt's choice of kernel for the
    batched product swings with the count (16k rows x 256 x 256 on MI355X, product alone: 26 us
    with 8 slabs, 42-44 us with 6 or 9, 140-250 us with 16 to 32; 128k rows x 256 x 100: 72-77
    us with 16, 32 or 64, 116 us with 8), so only the best measured counts are used; rows that
    do not fill a slab go to the finishing kernel."""
    return 8 if rows < 65536 else 32


class _DeferredProducts:
    """The ``a_l^T @ b_l`` products of L layers (all of one shape) whose last step -- slab sum +
    remainder rows -- is run for all layers by ONE launch once the last of them has been
    started, into ``stack`` ([L, m, n]); a launch per layer costs 5-6 us each for a few
    microseconds' worth of work."""

    def __init__(self, stack: Tensor):
        self.stack = stack
        n_layers = stack.size(0)
        self.parts, self.tails = [None] * n_layers, [None] * n_layers
        self.pending = []

    def slot(self, index: int) -> "_DeferredSlot":
        return _DeferredSlot(self, index)

    def add(self, index: int, a: Tensor, b: Tensor) -> Tensor:
        """Starts product ``index`` (the batched GEMM over row slabs); its value is in
        ``stack[index]`` after :meth:`finish`."""
        rows = a.size(0)
        slabs = _slab_count(rows)
        k = (rows // slabs) * slabs
        self.parts[index] = torch.bmm(a[:k].unflatten(0, (slabs, k // slabs)).transpose(1, 2),
                                      b[:k].unflatten(0, (slabs, k // slabs)))
        self.tails[index] = (a[k:], b[k:]) if k < rows else None
        self.pending.append(index)
        return self.stack[index]

    def finish(self) -> None:
        if not self.pending:
            return
        done = sorted(self.pending)
        shapes = {(tuple(self.parts[i].shape), None if self.tails[i] is None
                   else self.tails[i][0].size(0)) for i in done}
        if done == list(range(self.stack.size(0))) and len(shapes) == 1 and len(done) <= 16:
            ops.tn_finish_batched(self.parts, self.tails, self.stack)
        else:
            for i in done:
                t = self.tails[i]
                ops.tn_finish(self.parts[i], None if t is None else t[0],
                              None if t is None else t[1], self.stack[i])
        self.pending = []
        self.parts = [None] * len(self.parts)
        self.tails = [None] * len(self.tails)


class _DeferredSlot:
    def __init__(self, owner: _DeferredProducts, index: int):
        self.owner, self.index = owner, index


def _tn_product(a: Tensor, b: Tensor, out=None) -> Tensor:
    """``a^T @ b`` for tall operands ([rows, m], [rows, n], rows >> m, n): the shape of every
    weight gradient here.  A [m, n] output gives a GEMM only a handful of tiles while the
    reduction runs over 10^4..10^5 rows, so the rows are cut into slabs that run as one
    batched product (8x / 32x the tiles) and are summed afterwards.  Measured on MI355X:
    16k rows x 256 x 256: 49 us with 8 slabs vs 107 us as one GEMM; 128k rows x 256 x 100:
    77 us with 32 slabs, 123 us with 8, 376 us as one GEMM.  Rows may be strided (slices of a
    wider matrix): nothing is copied."""
    rows = a.size(0)
    direct = rows < 4096 or a.dim() != 2 or a.stride(1) != 1 or b.stride(1) != 1
    if isinstance(out, _DeferredSlot):
        # ``out``: a slot of _DeferredProducts -- the product is started here and finished,
        # together with the other layers', by its owner
        if a.is_cuda and a.dtype == torch.float32 and not direct:
            return out.owner.add(out.index, a, b)
        out = out.owner.stack[out.index]
        return out.copy_(_tn_product(a, b))
    if out is not None and not (a.is_cuda and a.dtype == torch.float32 and rows >= 4096):
        out = None          # (only the finishing kernel writes where it is told to)
    if direct:
        return a.t() @ b
    slabs = _slab_count(rows)
    k = (rows // slabs) * slabs
    parts = torch.bmm(a[:k].unflatten(0, (slabs, k // slabs)).transpose(1, 2),
                      b[:k].unflatten(0, (slabs, k // slabs)))
    if not parts.is_cuda or parts.dtype != torch.float32:
        out = parts.sum(0)
        if k < rows:
            out.addmm_(a[k:].t(), b[k:])
        return out
    # slab sum and remainder rows in one launch (a GEMM over < 10 rows costs a full launch)
    return ops.tn_finish(parts, a[k:] if k < rows else None, b[k:] if k < rows else None, out)


def _column_sums(x: Tensor) -> Tensor:
    """``x.sum(0)`` of a tall matrix (bias gradients).  On the GPU: ops.column_sums (two
    launches at streaming rate; 128k x 256: 54 us in four torch launches -> 27 us, 16k x 47:
    20 -> 7 us on MI355X).  Elsewhere in two stages: 64 row slabs reduced side by side, then
    the 64 partial rows."""
    rows = x.size(0)
    if rows < 4096 or x.dim() != 2:
        return x.sum(0)
    if x.is_cuda and x.dtype == torch.float32 and x.stride(1) == 1:
        return ops.column_sums(x)
    k = (rows // 64) * 64
    if k == rows:
        return x.unflatten(0, (64, k // 64)).sum(1).sum(0)
    part = torch.empty((64 + rows - k, x.size(1)), dtype=x.dtype, device=x.device)
    torch.sum(x[:k].unflatten(0, (64, k // 64)), dim=1, out=part[:64])
    part[64:] = x[k:]           # the remainder rows join the second stage
    return part.sum(0)


class CrossEntropyLoss:
    """``torch.nn.CrossEntropyLoss()`` with its defaults (main.py:60): the fused HIP form
    (ops.softmax_cross_entropy) for fp32 logits on the GPU, torch's own elsewhere."""

    def __call__(self, logits: Tensor, target: Tensor) -> Tensor:
        if (logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 2
                and target.dtype == torch.int64 and target.dim() == 1 and logits.size(0) > 0):
            if logits.stride(1) != 1:
                logits = logits.contiguous()
            return ops.softmax_cross_entropy(logits, target.contiguous())
        return torch.nn.functional.cross_entropy(logits, target)


class _RowsLinear(torch.autograd.Function):
    """``x @ W^T + b`` over node rows with the weight gradient evaluated by :func:`_tn_product`."""

    @staticmethod
    def forward(ctx, x, weight, bias, gathered=True, push=None):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        if (push is not None and gathered and x.is_cuda and x.dim() == 2
                and x.dtype == torch.float32 and x.stride(1) == 1 and weight.stride(1) == 1
                and ops.rows_gemm_supported(x.size(1), weight.size(0))
                and push.fits(x.size(0), weight.size(0), x) and fused_layer_kernel_enabled()):
            # product, bias and the (activated) push of the result in one matrix-core launch
            out = ops.rows_gemm(x, weight, True, bias=bias, table=push.table,
                                table_row0=push.row0, relu_push=push.relu,
                                table_only=push.only and not torch.is_grad_enabled())
            push.done = True
            return out
        if not gathered:    # plain contiguous result (bias in the GEMM's epilogue)
            return x @ weight.t() if bias is None else torch.addmm(bias, x, weight.t())
        if bias is None:
            return _mm_rows(x, weight.t())
        if not x.is_cuda or x.dim() != 2:
            return torch.addmm(bias, x, weight.t())
        return torch.addmm(bias, x, weight.t(),
                           out=ops.rows_empty(x.size(0), weight.size(0), x.device, x.dtype))

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        if dy.stride(-1) != 1:      # (rows with a pitch of their own go to the GEMMs as they are)
            dy = dy.contiguous()
        dx = _mm_rows(dy, weight) if ctx.needs_input_grad[0] else None
        dw = _tn_product(dy, x) if ctx.needs_input_grad[1] else None
        db = _column_sums(dy) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db, None, None


class _EmbedRelu(torch.autograd.Function):
    """``x_0 = relu(x W^T + b)`` returned twice: all rows (the input of the first aggregation)
    and the first ``n_batch`` rows (the ``x_0`` term of every GCNII layer, models/gcn2.py:46-48).
    With the slice taken inside, the backward pass adds the two incoming gradients on the
    batch rows only, instead of materialising a zero-padded full-size copy of the small one and
    adding two full-size matrices."""

    @staticmethod
    def forward(ctx, x, weight, bias, n_batch, grad_gated=False):
        ctx.grad_gated = grad_gated
        if bias is not None and x.is_cuda:
            # bias and relu ride in the GEMM epilogue (hipBLASLt): 104 us instead of 145 us for
            # the 128k x 100 -> 256 embedding of a products batch
            # (into a contiguous matrix: with a strided output the epilogue fusion is lost --
            # GEMM + copy + clamp, 64 us more -- which costs more than the first aggregation
            # gains from a padded row pitch of its source)
            h = torch._addmm_activation(bias, x, weight.t(), use_gelu=False)
        else:
            h = (x @ weight.t() if bias is None else torch.addmm(bias, x, weight.t())).relu_()
        ctx.save_for_backward(x, weight, h)
        ctx.has_bias, ctx.n_batch = bias is not None, n_batch
        return h, h[:n_batch]

    @staticmethod
    def backward(ctx, d_all, d_batch):
        x, weight, h = ctx.saved_tensors
        nb = ctx.n_batch
        if d_all is None:
            g = torch.zeros_like(h)
        elif ctx.grad_gated:
            # the consumer (the first aggregation's backward) zeroed it where h <= 0 already
            # and hands over a buffer of its own: used in place
            g = d_all if d_all.stride(-1) == 1 else d_all.contiguous()
        else:
            # (a new tensor; a padded row pitch of d_all needs no copy in front)
            g = torch.ops.aten.threshold_backward(
                d_all if d_all.stride(-1) == 1 else d_all.contiguous(), h, 0)
            if g.stride(-1) != 1:
                g = g.contiguous()
        if d_batch is not None:
            if g.is_cuda and g.dtype == torch.float32 and d_batch.stride(-1) == 1:
                ops.gate_add_rows_(g[:nb], d_batch, h[:nb])     # one launch instead of two
            else:
                g[:nb] += torch.ops.aten.threshold_backward(d_batch.contiguous(), h[:nb], 0)
        dx = g @ weight if ctx.needs_input_grad[0] else None
        dw = _tn_product(g, x) if ctx.needs_input_grad[1] else None
        db = _column_sums(g) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db, None, None


def embed_relu(lin: Linear, x: Tensor, n_batch: int, grad_gated: bool = False):
    """``(relu(lin(x)), relu(lin(x))[:n_batch])`` through :class:`_EmbedRelu`.  ``grad_gated``:
    the gradient arriving for the first output is already zeroed where the output is 0 (its
    only consumer is an aggregation called with ``gate_grad=True``)."""
    return _EmbedRelu.apply(x, lin.weight, lin.bias, n_batch, grad_gated)


class PushTarget:
    """Where the History push of a layer's output goes when the kernel that PRODUCES the output
    does it: rows ``[row0, row0 + rows)`` of ``table`` receive the output (``relu``:
    ``max(0, output)``).  ``done`` is set by the producer that took the offer; otherwise the
    caller pushes as usual."""

    def __init__(self, table: Tensor, row0: int, relu: bool, only: bool = False):
        self.table, self.row0, self.relu, self.done = table, int(row0), bool(relu), False
        # the caller reads the result from the table alone (a refresh sweep outside autograd):
        # the producer need not write its own output matrix
        self.only = bool(only)

    def fits(self, rows: int, width: int, like: Tensor) -> bool:
        t = self.table
        return (t.is_cuda and t.device == like.device and t.dtype == torch.float32
                and t.size(1) == width and 0 <= self.row0 and self.row0 + rows <= t.size(0))


def fused_layer_kernel_enabled() -> bool:
    """``INCAGG_FUSED_LAYER=0`` keeps the dense update with hipBLASLt and the push as a copy."""
    import os
    return os.environ.get("INCAGG_FUSED_LAYER", "1") != "0"


class _TwoTermUpdate(torch.autograd.Function):
    """``h @ W1 + x0 @ W2`` (``x0`` / ``W2`` optional) with split-row weight gradients.
    ``term``: the second summand ready-made; the product is then accumulated INTO it (the GEMM
    reads and writes the same matrix: no copy of the summand in front of it) and it is the
    result.  ``push``: a :class:`PushTarget` -- with a ready-made summand the whole update then
    runs as ONE hand-written matrix-core launch (ops.rows_gemm) that also stores the result,
    activated, into the next layer's table."""

    @staticmethod
    def forward(ctx, h, w1, x0, w2, term=None, dw1_out=None, push=None):
        ctx.save_for_backward(h, w1, x0, w2)
        ctx.has_term = term is not None
        ctx.dw1_out = dw1_out       # where the caller wants the gradient of w1 (may be ignored)
        if term is not None:
            ctx.mark_dirty(term)
            if (push is not None and h.is_cuda and h.dtype == torch.float32 and h.dim() == 2
                    and h.stride(1) == 1 and w1.stride(1) == 1 and term.stride(1) == 1
                    and ops.rows_gemm_supported(h.size(1), w1.size(1))
                    and push.fits(h.size(0), w1.size(1), h) and fused_layer_kernel_enabled()):
                ops.rows_gemm(h, w1, False, out=term, accumulate=True, table=push.table,
                              table_row0=push.row0, relu_push=push.relu,
                              table_only=push.only and not torch.is_grad_enabled())
                push.done = True
                return term
            return term.addmm_(h, w1)
        if x0 is None:
            return _mm_rows(h, w1)
        return _mm_rows(x0, w2).addmm_(h, w1)

    @staticmethod
    def backward(ctx, dy):
        h, w1, x0, w2 = ctx.saved_tensors
        if dy.stride(-1) != 1:
            dy = dy.contiguous()
        dh = _mm_rows(dy, w1.t()) if ctx.needs_input_grad[0] else None
        dw1 = _tn_product(h, dy, ctx.dw1_out) if ctx.needs_input_grad[1] else None
        dx0 = dw2 = None
        if x0 is not None:
            dx0 = dy @ w2.t() if ctx.needs_input_grad[2] else None
            dw2 = _tn_product(x0, dy) if ctx.needs_input_grad[3] else None
        dterm = dy if ctx.has_term and ctx.needs_input_grad[4] else None
        return dh, dw1, dx0, dw2, dterm, None, None


class _BatchNormRows(torch.autograd.Function):
    """Training-mode ``BatchNorm1d`` over the rows of a tall matrix as streaming row passes
    (ops.batch_norm_forward / _backward); the input gradient overwrites the incoming gradient
    when that is a private matrix of the right layout."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps):
        y, mean, rstd = ops.batch_norm_forward(x, weight, bias, running_mean, running_var,
                                               momentum, eps)
        ctx.save_for_backward(x, weight, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, mean, rstd = ctx.saved_tensors
        if not ops.batch_norm_supported(dy):     # (an expanded or oddly strided gradient)
            dy = ops.rows_empty(dy.size(0), dy.size(1), dy.device).copy_(dy)
        dx, dweight, dbias = ops.batch_norm_backward(dy, x, weight, mean, rstd)
        return (dx if ctx.needs_input_grad[0] else None,
                dweight if weight is not None and ctx.needs_input_grad[1] else None,
                dbias if ctx.needs_input_grad[2] else None, None, None, None, None)


def batch_norm_rows(bn: torch.nn.BatchNorm1d, h: Tensor) -> Tensor:
    """``bn(h)``.  In training mode on a float32 device matrix this runs the package's streaming
    row kernels (torch's own kernel walks down the columns of the row-major matrix: 13 ms for the
    backward pass of a 169k x 256 matrix on MI355X against 0.2 ms); everything else -- evaluation
    mode, cumulative averaging (``momentum=None``), unsupported layouts -- goes to the module."""
    if (bn.training and bn.momentum is not None and bn.track_running_stats
            and ops.batch_norm_supported(h) and (bn.weight is None or bn.weight.is_cuda)):
        bn.num_batches_tracked.add_(1)
        return _BatchNormRows.apply(h, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                    float(bn.momentum), float(bn.eps))
    return bn(h)


class _Layers(torch.autograd.Function):
    """``stack.unbind(0)`` whose backward takes the gradients of the pieces where they are when
    their producers wrote them into ``grad_stack`` (a ``[L, ...]`` matrix, piece l in
    ``grad_stack[l]``) instead of copying every one of them into a fresh stack."""

    @staticmethod
    def forward(ctx, stack, grad_stack, deferred=None):
        ctx.grad_stack, ctx.deferred = grad_stack, deferred
        return tuple(t.data for t in stack.unbind(0))

    @staticmethod
    def backward(ctx, *grads):
        gs = ctx.grad_stack
        if ctx.deferred is not None:    # every layer's product has been started by now
            ctx.deferred.finish()
        for i, g in enumerate(grads):
            if g is None:
                gs[i].zero_()
            elif g.data_ptr() != gs[i].data_ptr() or g.stride() != gs[i].stride():
                gs[i].copy_(g)
        return gs, None, None


class _ReluInPlace(torch.autograd.Function):
    """``h.relu_()`` whose backward writes the gated gradient straight into ``grad_out`` (a
    matrix of h's shape, e.g. this layer's block of the side-by-side gradient stack) instead
    of into a fresh matrix that would then be copied there."""

    @staticmethod
    def forward(ctx, h, grad_out):
        ctx.mark_dirty(h)
        if h.is_cuda and h.dtype == torch.float32 and h.dim() == 2 and h.stride(1) == 1:
            ops.relu_rows_(h)       # (twice the rate of torch's in-place clamp on MI355X)
        else:
            h.relu_()
        ctx.save_for_backward(h)
        ctx.grad_out = grad_out
        return h

    @staticmethod
    def backward(ctx, g):
        h, = ctx.saved_tensors
        if (g.is_cuda and g.dtype == torch.float32 and g.dim() == 2 and g.stride(1) == 1
                and g.shape == h.shape):
            # one launch at full rate whatever the row pitches (torch's kernel for pitched
            # operands runs at half of it)
            return ops.gate_rows(g, h, out=ctx.grad_out), None
        if ctx.grad_out is None:
            return torch.ops.aten.threshold_backward(g, h, 0), None
        return torch.ops.aten.threshold_backward.grad_input(g, h, 0, grad_input=ctx.grad_out), None


def relu_into(h: Tensor, grad_out: Optional[Tensor] = None) -> Tensor:
    """In-place relu (the row kernel for fp32 matrices on the GPU, also with padded row
    pitches); ``grad_out``: where its backward should leave the gradient of ``h``."""
    rows_kernel = h.is_cuda and h.dtype == torch.float32 and h.dim() == 2 and h.stride(1) == 1
    if not (torch.is_grad_enabled() and h.requires_grad):
        return ops.relu_rows_(h) if rows_kernel else h.relu_()
    if grad_out is not None and grad_out.shape != h.shape:
        grad_out = None
    if grad_out is None and not rows_kernel:
        return h.relu_()
    return _ReluInPlace.apply(h, grad_out)


class _FoldedSideBySide(torch.autograd.Function):
    """``[C, L*C]`` matrix whose column block l is ``const_l + scale_l * stack[l]`` (the folded
    second weights of all layers laid out for ONE wide GEMM), from the ``[L, C, C]`` stack of
    the weights.  One launch forward; backward one launch that writes the gradient of the
    stack contiguously, so that the per-layer gradients are slices nobody has to copy
    (cutting column blocks out of a ``[C, L*C]`` gradient costs a copy per layer)."""

    @staticmethod
    def forward(ctx, stack, const_cat, scale_cat):
        n_layers, c, _ = stack.shape
        ctx.save_for_backward(scale_cat)
        out = torch.addcmul(const_cat.view(c, n_layers, c), stack.permute(1, 0, 2),
                            scale_cat.view(1, n_layers, c))
        return out.view(c, n_layers * c)

    @staticmethod
    def backward(ctx, d):
        scale_cat, = ctx.saved_tensors
        c = d.size(0)
        n_layers = d.size(1) // c
        out = torch.empty((n_layers, c, c), dtype=d.dtype, device=d.device)
        torch.mul(d.view(c, n_layers, c).permute(1, 0, 2),
                  scale_cat.view(n_layers, 1, c), out=out)
        return out, None, None


class _FoldedWeights(torch.autograd.Function):
    """Both folded weight sets of all GCNII layers from the stack ``[W1_0.., W2_0..]`` in one
    launch forward and one backward (ops.gcn2_fold_weights)."""

    @staticmethod
    def forward(ctx, ws, alpha, betas):
        ctx.alpha, ctx.betas = alpha, betas
        ctx.set_materialize_grads(False)
        return ops.gcn2_fold_weights(ws, alpha, betas)

    @staticmethod
    def backward(ctx, d1, d2):
        n_layers = len(ctx.betas)
        if d1 is None and d2 is None:
            return None, None, None
        if d1 is None:
            d1 = torch.zeros((n_layers, d2.size(0), d2.size(0)), dtype=d2.dtype, device=d2.device)
        if d2 is None:
            d2 = torch.zeros((d1.size(1), n_layers * d1.size(1)), dtype=d1.dtype, device=d1.device)
        return ops.gcn2_fold_weights_backward(d1.contiguous(), d2.contiguous(), ctx.alpha,
                                              ctx.betas), None, None


class _X0Terms(torch.autograd.Function):
    """``x_0 @ [W_1 | ... | W_L]`` handed out as the L column blocks ``[rows, C]`` of ONE
    ``[rows, L*C]`` matrix.  ``grad_stack``: a matrix of that shape whose column blocks the
    producers of the blocks' gradients may have written in place (see
    :func:`~torch_geometric_autoscale.aggregate.spmm`, ``grad_out``); gradients that arrive
    elsewhere are copied into their block, and the two wide backward GEMMs run on the stack
    without concatenating anything."""

    @staticmethod
    def forward(ctx, x_0, wcat, n_layers, grad_stack):
        # (row pitch: the blocks become layer outputs, which the next aggregation gathers from)
        buf = _mm_rows(x_0, wcat)
        c = wcat.size(1) // n_layers
        ctx.save_for_backward(x_0, wcat)
        ctx.grad_stack, ctx.c = grad_stack, c
        # .data: a version counter per block.  The blocks never overlap and every consumer
        # updates its own one in place; with the counter of `buf` shared among them, what
        # another consumer saved of ITS block would count as modified.
        return tuple(buf[:, i * c:(i + 1) * c].data for i in range(n_layers))

    @staticmethod
    def backward(ctx, *dterms):
        x_0, wcat = ctx.saved_tensors
        c, stack = ctx.c, ctx.grad_stack
        rows = x_0.size(0)
        if stack is not None and tuple(stack.shape) == (rows, c * len(dterms)):
            for i, g in enumerate(dterms):
                blk = stack[:, i * c:(i + 1) * c]
                if g is None:
                    blk.zero_()
                elif g.data_ptr() != blk.data_ptr() or g.stride() != blk.stride():
                    blk.copy_(g)
            d = stack
        else:
            d = torch.cat([x_0.new_zeros(rows, c) if g is None else g for g in dterms], dim=1)
        dx0 = d @ wcat.t() if ctx.needs_input_grad[0] else None
        dw = _tn_product(x_0, d) if ctx.needs_input_grad[1] else None
        return dx0, dw, None, None


class _ScaledFanOut(torch.autograd.Function):
    """``scale * x`` handed out ``n`` times, once per consumer (the same memory each time);
    backward: ``scale`` times the sum of the consumers' gradients in ONE launch -- instead of a
    multiplication per consumer and autograd's chain of additions."""

    @staticmethod
    def forward(ctx, x, scale, n):
        ctx.scale = scale
        y = ops.rows_sum([x], scale)
        return tuple(y.data for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        grads = [g if g.stride(-1) == 1 else g.contiguous() for g in grads if g is not None]
        if not grads:
            return None, None, None
        return ops.rows_sum(grads, ctx.scale), None, None


def scaled_fan_out(x: Tensor, scale: float, n: int):
    """``n`` tensors equal to ``scale * x`` for ``n`` consumers that each ADD theirs (APPNP's
    ``alpha * x_0``, models/appnp.py:85, consumed with ``add_scale = 1``); fp32 [rows, width]
    on the GPU, ``n <= 16``."""
    return _ScaledFanOut.apply(x if x.stride(-1) == 1 else x.contiguous(), float(scale), int(n))


class RowsLinear(Linear):
    """``torch.nn.Linear`` (same parameters, same state dict) for [rows, features] inputs.
    ``gathered=False``: nothing aggregates over the output (a model's last layer), so it gets
    no padded row pitch -- with a width such as 47 the padded form costs two extra copies."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True,
                 gathered: bool = True):
        super().__init__(in_features, out_features, bias=bias)
        self.gathered = gathered

    def forward(self, x: Tensor) -> Tensor:
        if x.dim() != 2 or not x.is_cuda:
            return super().forward(x)
        return _RowsLinear.apply(x, self.weight, self.bias, self.gathered)


def glorot_(w: Tensor) -> None:
    a = math.sqrt(6.0 / (w.size(-2) + w.size(-1)))
    with torch.no_grad():
        w.uniform_(-a, a)


class GCNConv(torch.nn.Module):
    """``out = A (x W^T) + b`` with a pre-normalised adjacency (``normalize=False``),
    as used in models/gcn.py:63.  The product is evaluated as ``(A x) W^T`` whenever the
    input is not wider than the output, so the gather runs at the narrower width and
    the GEMM only over the batch rows."""

    def __init__(self, in_channels: int, out_channels: int, normalize: bool = False,
                 bias: bool = True):
        super().__init__()
        if normalize:
            raise NotImplementedError("normalise the adjacency once up front (gcn_norm)")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin = RowsLinear(in_channels, out_channels, bias=False)
        self.bias = Parameter(torch.zeros(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        glorot_(self.lin.weight)
        if self.bias is not None:
            torch.nn.init.zeros_(self.bias)

    def update(self, agg: Tensor, push: Optional["PushTarget"] = None) -> Tensor:
        """Dense part applied to an aggregate of the layer INPUT (models/gcn.py:242-244).
        ``push``: see :class:`PushTarget`."""
        if self.bias is None or agg.dim() != 2 or not agg.is_cuda:
            h = self.lin(agg)
            return h if self.bias is None else h + self.bias
        # bias in the GEMM's epilogue, its gradient as streaming column sums
        return _RowsLinear.apply(agg, self.lin.weight, self.bias, True, push)

    def forward(self, x: Tensor, adj_t: SparseTensor) -> Tensor:
        if self.in_channels <= self.out_channels:
            return self.update(spmm(adj_t, x))
        h = spmm(adj_t, self.lin(x))
        return h if self.bias is None else h + self.bias


class GCN2Conv(torch.nn.Module):
    """GCNII layer with a pre-normalised adjacency (models/gcn2.py:46-48):
    ``p = (1-alpha) A x + alpha x_0``; shared weights: ``(1-beta) p + beta p W1``;
    otherwise the two summands get their own ``W1`` / ``W2``.  ``beta = log(theta/layer + 1)``."""

    def __init__(self, channels: int, alpha: float, theta: Optional[float] = None,
                 layer: Optional[int] = None, shared_weights: bool = True,
                 normalize: bool = False):
        super().__init__()
        if normalize:
            raise NotImplementedError("normalise the adjacency once up front (gcn_norm)")
        self.channels, self.alpha = channels, alpha
        self.beta = 1.0
        if theta is not None or layer is not None:
            assert theta is not None and layer is not None
            self.beta = math.log(theta / layer + 1)
        self.weight1 = Parameter(torch.empty(channels, channels))
        self.weight2 = None if shared_weights else Parameter(torch.empty(channels, channels))
        self._eye = None
        self.reset_parameters()

    def reset_parameters(self):
        glorot_(self.weight1)
        if self.weight2 is not None:
            glorot_(self.weight2)

    def aggregate(self, x: Tensor, x_0: Tensor, adj_t: SparseTensor) -> Tensor:
        """``A x`` (kept un-mixed so that :meth:`forward_after_propagate` takes it)."""
        return spmm(adj_t, x)

    def hold_folded_weights(self, hold: bool) -> None:
        """Between ``hold_folded_weights(True)`` and ``(False)`` the caller promises that the
        weights do not change (a refresh sweep visits every batch with the same weights):
        folded weights are then computed once and reused.  Both calls drop what was kept, so
        nothing survives from one such span to the next -- weights may have been updated in
        between by code this module never sees (a replayed HIP graph, an optimizer)."""
        self._hold_folds = bool(hold)
        self._held_folds = {}

    def _folded(self, weight: Tensor, scale: float) -> Tensor:
        """``scale * (beta * W + (1 - beta) * I)``: the identity part of
        ``(1 - beta) p + beta p W`` folded into the weight, so that the layer is one GEMM
        per term and no elementwise pass over the [rows, channels] activations."""
        eye = self._eye
        if eye is None or eye.device != weight.device:
            eye = self._eye = torch.eye(self.channels, device=weight.device, dtype=weight.dtype)
        held = getattr(self, "_hold_folds", False) and not torch.is_grad_enabled()
        if held:
            key = "w2" if weight is self.weight2 else "w1"
            hit = self._held_folds.get((key, scale))
            if hit is not None and hit.device == weight.device:
                return hit
        out = torch.add(eye * (scale * (1.0 - self.beta)), weight, alpha=scale * self.beta)
        if held:
            self._held_folds[(key, scale)] = out
        return out

    def forward_after_propagate(self, h: Tensor, x_0: Tensor, w1_folded: Optional[Tensor] = None,
                                x0_term: Optional[Tensor] = None,
                                dw1_out: Optional[Tensor] = None,
                                push: Optional[PushTarget] = None) -> Tensor:
        """Everything after the propagation, given ``h = A x`` (models/gcn2.py:257,306).
        ``w1_folded`` / ``x0_term``: this layer's slice of :func:`gcn2_batched_terms`, computed
        for all layers at once by the caller.  ``push``: see :class:`PushTarget`."""
        b = h.size(0)
        if x0_term is not None:
            return _TwoTermUpdate.apply(h, w1_folded, None, None, x0_term, dw1_out, push)
        x_0 = x_0 if x_0.size(0) == b else x_0[:b]
        if self.weight2 is None:
            p = torch.add(h * (1 - self.alpha), x_0, alpha=self.alpha)
            return _TwoTermUpdate.apply(p, self._folded(self.weight1, 1.0), None, None)
        return _TwoTermUpdate.apply(h, self._folded(self.weight1, 1.0 - self.alpha), x_0,
                                    self._folded(self.weight2, self.alpha))

    def update_mixed(self, p: Tensor) -> Tensor:
        """Shared weights, given ``p = (1 - alpha) A x + alpha x_0`` from the kernel epilogue."""
        return _TwoTermUpdate.apply(p, self._folded(self.weight1, 1.0), None, None)

    def forward_no_neighbor(self, x: Tensor, x_0: Tensor) -> Tensor:
        """The layer with the propagation replaced by the identity (models/gcn2.py:159,175)."""
        return self.forward_after_propagate(x, x_0)

    def forward(self, x: Tensor, x_0: Tensor, adj_t: SparseTensor) -> Tensor:
        b = adj_t.size(0)
        if self.weight2 is None:
            # the mix (1-alpha) A x + alpha x_0 rides in the aggregation epilogue
            return self.update_mixed(spmm(adj_t, x, scale=1 - self.alpha, add=x_0[:b],
                                          add_scale=self.alpha))
        return self.forward_after_propagate(spmm(adj_t, x), x_0)


def gcn2_batched_terms(convs, x_0: Tensor, grad_stack: Optional[Tensor] = None):
    """For GCNII layers with separate weights: the folded first weights of ALL layers (a tuple
    of L ``[C, C]`` matrices) and the ``x_0`` summands of all layers, a tuple of L ``[rows, C]``
    matrices cut from ONE ``[rows, C] x [C, L*C]`` product -- ``x_0`` is the same matrix in
    every layer (models/gcn2.py:46-48), so its L small GEMMs (and, in the backward pass, L
    weight-gradient GEMMs and L accumulations into ``d x_0``) collapse into one wide GEMM each.
    Folding ``scale * (beta W + (1 - beta) I)`` is one fused multiply-add per weight set over
    constants built once; the folded first weights are handed out with ``unbind`` so that their
    gradients come back as one stack instead of L zero-fill + copy + add rounds, the ``x_0``
    summands by :class:`_X0Terms` (``grad_stack``: see there).  Third result: where the layers
    may leave the gradients of their folded weights (``forward_after_propagate``'s ``dw1_out``),
    or ``None`` outside training."""
    n_layers, c = len(convs), convs[0].channels
    dev, dt = x_0.device, x_0.dtype
    alpha = convs[0].alpha
    if x_0.is_cuda and dt == torch.float32 and n_layers <= 64:
        ws = torch.stack([conv.weight1 for conv in convs] + [conv.weight2 for conv in convs])
        w1f, w2cat = _FoldedWeights.apply(ws, alpha, tuple(conv.beta for conv in convs))
        return _batched_terms(x_0, w1f, w2cat, n_layers, grad_stack)
    consts = convs[0].__dict__.setdefault("_batched_consts", {})
    key = (dev, dt, n_layers, alpha)
    if key not in consts:   # built once: no host-to-device copy inside a step
        beta = torch.tensor([conv.beta for conv in convs], device=dev, dtype=dt)
        eye = torch.eye(c, device=dev, dtype=dt)
        consts[key] = (
            ((1.0 - alpha) * (1.0 - beta)).view(-1, 1, 1) * eye,          # [L, C, C]
            ((1.0 - alpha) * beta).view(-1, 1, 1).contiguous(),           # [L, 1, 1]
            # second weight set laid out side by side, [C, L*C]: layer l in columns [lC, (l+1)C)
            torch.cat([alpha * (1.0 - b) * eye for b in beta.tolist()], dim=1),
            (alpha * beta).repeat_interleave(c).view(1, -1).contiguous())
    c1, b1, c2cat, b2cat = consts[key]
    w1f = torch.addcmul(c1, torch.stack([conv.weight1 for conv in convs]), b1)
    w2cat = _FoldedSideBySide.apply(torch.stack([conv.weight2 for conv in convs]), c2cat, b2cat)
    return _batched_terms(x_0, w1f, w2cat, n_layers, grad_stack)


def _batched_terms(x_0: Tensor, w1f: Tensor, w2cat: Tensor, n_layers: int,
                   grad_stack: Optional[Tensor]):
    terms = _X0Terms.apply(x_0, w2cat, n_layers, grad_stack)
    if not (torch.is_grad_enabled() and w1f.requires_grad):
        return w1f.unbind(0), terms, None
    # the layers' weight gradients are written side by side by their producers
    # (and finished together, by one launch, when the last layer's backward has run)
    dw1 = torch.empty_like(w1f)
    deferred = _DeferredProducts(dw1)
    return _Layers.apply(w1f, dw1, deferred), terms, [deferred.slot(i) for i in range(n_layers)]


class _SageUpdate(torch.autograd.Function):
    """``agg @ W_l^T + b_l + x_root @ W_r^T`` with the second GEMM accumulating into the result
    of the first (no addition pass, one matrix instead of three) and one backward pass over the
    incoming gradient for both terms."""

    @staticmethod
    def forward(ctx, agg, x_root, w_l, b_l, w_r):
        ctx.save_for_backward(agg, x_root, w_l, w_r)
        ctx.has_bias = b_l is not None
        out = ops.rows_empty(agg.size(0), w_l.size(0), agg.device, agg.dtype)
        if b_l is None:
            torch.mm(agg, w_l.t(), out=out)
        else:
            torch.addmm(b_l, agg, w_l.t(), out=out)
        return out.addmm_(x_root, w_r.t())

    @staticmethod
    def backward(ctx, dy):
        agg, x_root, w_l, w_r = ctx.saved_tensors
        if dy.stride(-1) != 1:
            dy = dy.contiguous()
        need = ctx.needs_input_grad
        d_agg = _mm_rows(dy, w_l) if need[0] else None
        d_root = _mm_rows(dy, w_r) if need[1] else None
        dw_l = _tn_product(dy, agg) if need[2] else None
        db = _column_sums(dy) if ctx.has_bias and need[3] else None
        dw_r = _tn_product(dy, x_root) if need[4] else None
        return d_agg, d_root, dw_l, db, dw_r


class SAGEConv(torch.nn.Module):
    """``lin_l(mean_j x_j) + lin_r(x_i)`` (models/graphsage.py:69, spelled out at :623-644)."""

    def __init__(self, in_channels: int, out_channels: int, aggr: str = "mean",
                 normalize: bool = False, root_weight: bool = True, bias: bool = True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.aggr, self.normalize, self.root_weight = aggr, normalize, root_weight
        self.project = False
        self.lin_l = RowsLinear(in_channels, out_channels, bias=bias)
        self.lin_r = RowsLinear(in_channels, out_channels, bias=False) if root_weight else None
        self.reset_parameters()

    def reset_parameters(self):
        self.lin_l.reset_parameters()
        if self.lin_r is not None:
            self.lin_r.reset_parameters()

    def update(self, agg: Tensor, x_root: Tensor) -> Tensor:
        if self.lin_r is not None and agg.is_cuda and agg.dim() == 2 \
                and agg.dtype == torch.float32 and agg.stride(1) == 1:
            x_root = x_root if x_root.size(0) == agg.size(0) else x_root[:agg.size(0)]
            out = _SageUpdate.apply(agg, x_root, self.lin_l.weight, self.lin_l.bias,
                                    self.lin_r.weight)
            return torch.nn.functional.normalize(out, p=2.0, dim=-1) if self.normalize else out
        out = self.lin_l(agg)
        if self.lin_r is not None:
            out = out + self.lin_r(x_root if x_root.size(0) == agg.size(0) else x_root[:agg.size(0)])
        if self.normalize:
            out = torch.nn.functional.normalize(out, p=2.0, dim=-1)
        return out

    def forward(self, x: Tensor, adj_t: SparseTensor) -> Tensor:
        return self.update(spmm(unweighted(adj_t), x, reduce=self.aggr), x)


def unweighted(adj_t: SparseTensor) -> SparseTensor:
    """``adj_t.set_value(None)`` (models/graphsage.py:631), cached on the adjacency so the
    transposed CSR of the value-less view is built once."""
    if not adj_t.has_value():
        return adj_t
    view = getattr(adj_t, "_unweighted_view", None)
    if view is None:
        view = adj_t.set_value(None, layout=None)
        adj_t._unweighted_view = view
    return view
