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

from .aggregate import spmm
from .sparse import SparseTensor


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
        self.lin = Linear(in_channels, out_channels, bias=False)
        self.bias = Parameter(torch.zeros(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        glorot_(self.lin.weight)
        if self.bias is not None:
            torch.nn.init.zeros_(self.bias)

    def update(self, agg: Tensor) -> Tensor:
        """Dense part applied to an aggregate of the layer INPUT (models/gcn.py:242-244)."""
        h = self.lin(agg)
        return h if self.bias is None else h + self.bias

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
        self.reset_parameters()

    def reset_parameters(self):
        glorot_(self.weight1)
        if self.weight2 is not None:
            glorot_(self.weight2)

    def aggregate(self, x: Tensor, x_0: Tensor, adj_t: SparseTensor) -> Tensor:
        """``A x`` (kept un-mixed so that :meth:`forward_after_propagate` takes it)."""
        return spmm(adj_t, x)

    def forward_after_propagate(self, h: Tensor, x_0: Tensor) -> Tensor:
        """Everything after the propagation, given ``h = A x`` (models/gcn2.py:257,306)."""
        b = h.size(0)
        x_0 = x_0[:b]
        if self.weight2 is None:
            p = torch.add(h * (1 - self.alpha), x_0, alpha=self.alpha)
            return torch.addmm(p, p, self.weight1, beta=1.0 - self.beta, alpha=self.beta)
        ax = h * (1 - self.alpha)
        x0 = x_0 * self.alpha
        out = torch.addmm(ax, ax, self.weight1, beta=1.0 - self.beta, alpha=self.beta)
        return out + torch.addmm(x0, x0, self.weight2, beta=1.0 - self.beta, alpha=self.beta)

    def forward_no_neighbor(self, x: Tensor, x_0: Tensor) -> Tensor:
        """The layer with the propagation replaced by the identity (models/gcn2.py:159,175)."""
        return self.forward_after_propagate(x, x_0)

    def forward(self, x: Tensor, x_0: Tensor, adj_t: SparseTensor) -> Tensor:
        b = adj_t.size(0)
        if self.weight2 is None:
            # the mix (1-alpha) A x + alpha x_0 rides in the aggregation epilogue
            p = spmm(adj_t, x, scale=1 - self.alpha, add=x_0[:b], add_scale=self.alpha)
            return torch.addmm(p, p, self.weight1, beta=1.0 - self.beta, alpha=self.beta)
        return self.forward_after_propagate(spmm(adj_t, x), x_0)


class SAGEConv(torch.nn.Module):
    """``lin_l(mean_j x_j) + lin_r(x_i)`` (models/graphsage.py:69, spelled out at :623-644)."""

    def __init__(self, in_channels: int, out_channels: int, aggr: str = "mean",
                 normalize: bool = False, root_weight: bool = True, bias: bool = True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.aggr, self.normalize, self.root_weight = aggr, normalize, root_weight
        self.project = False
        self.lin_l = Linear(in_channels, out_channels, bias=bias)
        self.lin_r = Linear(in_channels, out_channels, bias=False) if root_weight else None
        self.reset_parameters()

    def reset_parameters(self):
        self.lin_l.reset_parameters()
        if self.lin_r is not None:
            self.lin_r.reset_parameters()

    def update(self, agg: Tensor, x_root: Tensor) -> Tensor:
        out = self.lin_l(agg)
        if self.lin_r is not None:
            out = out + self.lin_r(x_root[:agg.size(0)])
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
