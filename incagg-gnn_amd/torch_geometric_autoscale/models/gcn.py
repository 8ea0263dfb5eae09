"""GCN on historical embeddings (constructor and module layout of models/gcn.py:31-95)."""
from typing import Any, Dict, Optional

import torch.nn.functional as F
from torch import Tensor
from torch.nn import BatchNorm1d, ModuleList

from ..nn import GCNConv, RowsLinear, batch_norm_rows, relu_into
from .base import ScalableGNN


class GCN(ScalableGNN):
    raw_input_layer = True

    def __init__(self, num_nodes: int, in_channels, hidden_channels: int, out_channels: int,
                 num_layers: int, dropout: float = 0.0, drop_input: bool = True,
                 batch_norm: bool = False, residual: bool = False, linear: bool = False,
                 pool_size: Optional[int] = None, buffer_size: Optional[int] = None, device=None):
        super().__init__(num_nodes, hidden_channels, num_layers, pool_size, buffer_size, device,
                         in_channels=in_channels)
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.dropout = dropout
        self.drop_input = drop_input
        self.batch_norm = batch_norm
        self.residual = residual
        self.linear = linear

        self.lins = ModuleList()
        if linear:
            self.lins.append(RowsLinear(in_channels, hidden_channels))
            self.lins.append(RowsLinear(hidden_channels, out_channels, gathered=False))

        self.convs = ModuleList()
        for i in range(num_layers):
            in_dim = out_dim = hidden_channels
            if i == 0 and not linear:
                in_dim = in_channels
            if i == num_layers - 1 and not linear:
                out_dim = out_channels
            self.convs.append(self._make_conv(in_dim, out_dim))

        self.bns = ModuleList([BatchNorm1d(hidden_channels) for _ in range(num_layers)])

    def _make_conv(self, in_dim: int, out_dim: int):
        return GCNConv(in_dim, out_dim, normalize=False)

    @property
    def reg_modules(self):
        if self.linear:
            return ModuleList(list(self.convs) + list(self.bns))
        return ModuleList(list(self.convs[:-1]) + list(self.bns))

    @property
    def nonreg_modules(self):
        return self.lins if self.linear else self.convs[-1:]

    def reset_parameters(self):
        super().reset_parameters()
        for lin in self.lins:
            lin.reset_parameters()
        for conv in self.convs:
            conv.reset_parameters()
        for bn in self.bns:
            bn.reset_parameters()

    # ---- hooks (the per-layer arithmetic of models/gcn.py:283-332)
    def _embed(self, x: Tensor, state: Dict[str, Any], batch_size: int) -> Tensor:
        if self.drop_input:
            x = F.dropout(x, p=self.dropout, training=self.training)
        if self.linear:
            x = self.lins[0](x).relu_()
            x = F.dropout(x, p=self.dropout, training=self.training)
        return x

    def _defers_relu(self) -> bool:
        # the relu at the end of a layer feeds the next aggregation and the History push only:
        # no batch norm in front of it, no residual or root term reading the layer input (a
        # subclass with its own update, e.g. GraphSAGE, reads it), no dropout behind it
        return (type(self)._conv_update is GCN._conv_update and not self.batch_norm
                and not self.residual and not self.linear
                and not (self.training and self.dropout > 0))

    def _pre_aggregates(self, layer: int) -> bool:
        conv = self.convs[layer]
        return conv.out_channels < conv.in_channels

    def _pre_aggregate(self, layer: int, x: Tensor) -> Optional[Tensor]:
        return self.convs[layer].lin(x) if self._pre_aggregates(layer) else None

    def _conv_update(self, layer: int, agg: Tensor, x: Tensor, pre: bool, push=None) -> Tensor:
        conv = self.convs[layer]
        if pre:
            return agg if conv.bias is None else agg + conv.bias
        return conv.update(agg, push)

    def _update(self, layer: int, agg: Tensor, x: Tensor, state: Dict[str, Any],
                pre: bool = False) -> Tensor:
        # (the push of this layer's output may ride with its GEMM when nothing stands between
        # the two: no batch norm, no residual, the activation left to the push)
        push = state.get('push_into')
        if push is not None and (self.batch_norm or self.residual or self.linear
                                 or not state.get('defer_relu')
                                 or layer >= self.num_layers - 1):
            push = None
        h = self._conv_update(layer, agg, x, pre, push)
        if layer < self.num_layers - 1 or self.linear:
            if self.batch_norm:
                h = batch_norm_rows(self.bns[layer], h)
            if self.residual and h.size(-1) == x.size(-1):
                h = h + (x if x.size(0) == h.size(0) else x[:h.size(0)])
            if state.get('defer_relu') and not self.linear:
                state['relu_deferred'] = True      # applied where h is consumed
                return h
            h = relu_into(h)
        if layer == self.num_layers - 1 and self.linear:
            h = F.dropout(h, p=self.dropout, training=self.training)
            h = self.lins[1](h)
        return h
