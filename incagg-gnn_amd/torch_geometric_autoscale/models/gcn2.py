"""GCNII on historical embeddings (constructor and module layout of models/gcn2.py:22-76)."""
from typing import Any, Dict, Optional

import torch.nn.functional as F
from torch import Tensor
from torch.nn import BatchNorm1d, ModuleList

import torch

from .. import ops
from ..nn import (GCN2Conv, RowsLinear, batch_norm_rows, embed_relu, gcn2_batched_terms, relu_into,
                  scaled_fan_out)
from .base import ScalableGNN


class GCN2(ScalableGNN):
    def __init__(self, num_nodes: int, in_channels, hidden_channels: int, out_channels: int,
                 num_layers: int, alpha: float, theta: float, shared_weights: bool = True,
                 dropout: float = 0.0, drop_input: bool = True, batch_norm: bool = False,
                 residual: bool = False, pool_size: Optional[int] = None,
                 buffer_size: Optional[int] = None, device=None):
        super().__init__(num_nodes, hidden_channels, num_layers, pool_size, buffer_size, device,
                         in_channels=in_channels)
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.dropout = dropout
        self.drop_input = drop_input
        self.batch_norm = batch_norm
        self.residual = residual

        self.lins = ModuleList()
        self.lins.append(RowsLinear(in_channels, hidden_channels))
        self.lins.append(RowsLinear(hidden_channels, out_channels, gathered=False))

        self.convs = ModuleList([
            GCN2Conv(hidden_channels, alpha=alpha, theta=theta, layer=i + 1,
                     shared_weights=shared_weights, normalize=False)
            for i in range(num_layers)])
        self.bns = ModuleList([BatchNorm1d(hidden_channels) for _ in range(num_layers)])

    @property
    def reg_modules(self):
        return ModuleList(list(self.convs) + list(self.bns))

    @property
    def nonreg_modules(self):
        return self.lins

    def reset_parameters(self):
        super().reset_parameters()
        for lin in self.lins:
            lin.reset_parameters()
        for conv in self.convs:
            conv.reset_parameters()
        for bn in self.bns:
            bn.reset_parameters()

    # ---- hooks (the per-layer arithmetic of models/gcn2.py:326-346)
    def _defers_relu(self) -> bool:
        # the relu at the end of a layer feeds the next aggregation and the History push only
        # (no residual branch reads the layer input, no dropout sits in between)
        return not self.residual and not (self.training and self.dropout > 0)

    def _embed(self, x: Tensor, state: Dict[str, Any], batch_size: int) -> Tensor:
        if self.drop_input:
            x = F.dropout(x, p=self.dropout, training=self.training)
        # with deferred activations the first aggregation's backward also applies the relu
        # mask of the embedding to the gradient of all rows
        gated = bool(state.get('defer_relu')) and torch.is_grad_enabled()
        state['embed_grad_gated'] = gated
        x, state['x_0'] = embed_relu(self.lins[0], x, batch_size, grad_gated=gated)
        # The x_0 terms of every layer in one GEMM: in training steps (one batch, all layers
        # back to back) and in the sweeps whenever the terms of ALL batches fit in memory next
        # to the tables (they stay resident from a batch's first layer to its last).
        if (self.convs[0].weight2 is not None
                and (torch.is_grad_enabled() or state.get('keep_layer_terms'))
                and len({conv.alpha for conv in self.convs}) == 1):
            stack = None
            if state.get('defer_relu') and torch.is_grad_enabled():
                # gradients of the layer outputs land side by side (the next layer's aggregation
                # writes its input gradient straight into block `layer`): the backward GEMMs of
                # the x_0 terms then need no concatenation
                c = self.convs[0].channels
                stack = ops.rows_empty(batch_size, c * self.num_layers, x.device, x.dtype)
                state['grad_blocks'] = [stack[:, i * c:(i + 1) * c]
                                        for i in range(self.num_layers)]
            state['w1f'], state['x0_terms'], state['dw1'] = gcn2_batched_terms(
                self.convs, state['x_0'], stack)
        elif (self.convs[0].weight2 is None and torch.is_grad_enabled() and x.is_cuda
              and x.dtype == torch.float32 and self.num_layers <= 16
              and len({conv.alpha for conv in self.convs}) == 1):
            # shared weights: alpha * x_0 computed once for all layers (it rides in every
            # layer's aggregation epilogue), its gradient summed over the layers in one launch
            state['x_0_mix'] = scaled_fan_out(state['x_0'], self.convs[0].alpha, self.num_layers)
        return F.dropout(x, p=self.dropout, training=self.training)

    def _post(self, layer: int, h: Tensor, x: Tensor, state: Dict[str, Any] = None) -> Tensor:
        if self.batch_norm:
            h = batch_norm_rows(self.bns[layer], h)
        if self.residual and h.size(-1) == x.size(-1):
            h = h + (x if x.size(0) == h.size(0) else x[:h.size(0)])
        if state is not None and state.get('defer_relu') and layer < self.num_layers - 1:
            state['relu_deferred'] = True      # applied by the next aggregation and the push
            return h
        blocks = None if state is None else state.get('grad_blocks')
        # (the last layer's gradient goes where the wide x_0 GEMMs of the backward pass read it)
        h = relu_into(h, blocks[layer] if blocks and layer == self.num_layers - 1 else None)
        if layer == self.num_layers - 1:
            h = F.dropout(h, p=self.dropout, training=self.training)
            h = self.lins[1](h)
        return h

    def _update(self, layer: int, agg: Tensor, x: Tensor, state: Dict[str, Any],
                pre: bool = False) -> Tensor:
        if 'x0_terms' in state:
            # (the push of this layer's output may ride with its GEMM when nothing stands between
            # the GEMM and the push: no batch norm, no residual, the activation left to the push)
            push = state.get('push_into')
            if push is not None and (self.batch_norm or self.residual
                                     or not state.get('defer_relu')
                                     or layer >= self.num_layers - 1):
                push = None
            h = self.convs[layer].forward_after_propagate(
                agg, state['x_0'], state['w1f'][layer], state['x0_terms'][layer],
                None if state.get('dw1') is None else state['dw1'][layer], push)
        else:
            h = self.convs[layer].forward_after_propagate(agg, state['x_0'])
        return self._post(layer, h, x, state)

    def _mix(self, layer: int, state: Dict[str, Any], batch_size: int):
        conv = self.convs[layer]
        if conv.weight2 is not None:
            return None
        terms = state.get('x_0_mix')
        if terms is not None:
            return 1.0 - conv.alpha, terms[layer], 1.0
        return 1.0 - conv.alpha, state['x_0'], conv.alpha

    def _update_mixed(self, layer: int, mixed: Tensor, x: Tensor, state: Dict[str, Any]) -> Tensor:
        return self._post(layer, self.convs[layer].update_mixed(mixed), x, state)
