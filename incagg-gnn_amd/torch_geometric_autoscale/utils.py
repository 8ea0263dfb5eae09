"""The helpers of the reference's utils.py: ``index2mask`` (utils.py:9-12), the score
(utils.py:15-35), random per-class splits (utils.py:38-59) and random edge dropout
(utils.py:62-73)."""
from typing import Optional, Tuple

import torch
from torch import Tensor

from .sparse import SparseTensor


def index2mask(idx: Tensor, size: int) -> Tensor:
    """Boolean mask of length ``size`` that is set at ``idx`` (utils.py:9-12)."""
    mask = torch.zeros(size, dtype=torch.bool, device=idx.device)
    mask[idx] = True
    return mask


def gen_masks(y: Tensor, train_per_class: int = 20, val_per_class: int = 30,
              num_splits: int = 20) -> Tuple[Tensor, Tensor, Tensor]:
    """``num_splits`` random splits, one per column of the ``[N, num_splits]`` masks: of every
    class ``train_per_class`` training and ``val_per_class`` validation nodes, the rest is the
    test set (utils.py:38-59)."""
    n, num_classes = y.size(0), int(y.max()) + 1
    train_mask = torch.zeros(n, num_splits, dtype=torch.bool)
    val_mask = torch.zeros(n, num_splits, dtype=torch.bool)
    columns = torch.arange(num_splits)
    for c in range(num_classes):
        members = (y == c).nonzero(as_tuple=False).reshape(-1)
        # one shuffled copy of the class per split, side by side
        shuffled = torch.stack([members[torch.randperm(members.numel())]
                                for _ in range(num_splits)], dim=1)
        picked = shuffled[:train_per_class]
        train_mask[picked, columns.expand_as(picked)] = True
        picked = shuffled[train_per_class:train_per_class + val_per_class]
        val_mask[picked, columns.expand_as(picked)] = True
    return train_mask, val_mask, ~(train_mask | val_mask)


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
