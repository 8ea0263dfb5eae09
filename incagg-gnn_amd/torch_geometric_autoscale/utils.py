"""The two helpers of the reference's utils.py that the training driver needs: the score
(utils.py:15-35) and random edge dropout (utils.py:62-73).  Dataset-split helpers of that
file are not part of this package."""
from typing import Optional

import torch
from torch import Tensor

from .sparse import SparseTensor


def compute_micro_f1(logits: Tensor, y: Tensor, mask: Optional[Tensor] = None) -> float:
    """Single-label targets (``y`` 1-D): share of rows whose arg-max equals the label.
    Multi-label targets: micro-averaged F1 of ``logits > 0`` against ``y > 0.5``, i.e.
    ``2 tp / (2 tp + fp + fn)``; 0 when nothing is predicted and nothing is true."""
    if mask is not None:
        logits, y = logits[mask], y[mask]
    if y.dim() == 1:
        if y.numel() == 0:
            return 0.0
        return float((logits.argmax(dim=-1) == y).sum()) / y.numel()
    pred, true = logits > 0, y > 0.5
    tp = int((pred & true).sum())
    errors = int((pred ^ true).sum())          # fp + fn
    if tp == 0:
        return 0.0
    return 2.0 * tp / (2.0 * tp + errors)


def dropout(adj_t: SparseTensor, p: float, training: bool = True) -> SparseTensor:
    """Random edge dropout on an adjacency (utils.py:62-73)."""
    if not training or p == 0.:
        return adj_t
    if adj_t.has_value():   # weights are dropped (and the survivors rescaled), structure kept
        value = torch.nn.functional.dropout(adj_t.storage.value(), p=p)
        return adj_t.set_value(value, layout='coo')
    mask = torch.rand(adj_t.nnz(), device=adj_t.device()) > p
    return adj_t.masked_select_nnz(mask, layout='coo')
