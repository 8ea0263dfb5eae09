"""APPNP on historical embeddings (constructor of models/appnp.py:19-42).  The History
tables are ``out_channels`` wide (models/appnp.py:24), in general not a multiple of 4."""
from typing import Any, Dict, Optional

import torch
import torch.nn.functional as F
from torch import Tensor
from torch.nn import ModuleList

from ..nn import RowsLinear, embed_relu, scaled_fan_out
from .base import ScalableGNN


class APPNP(ScalableGNN):
    _layer_dropout = False  # no dropout between propagation steps (models/appnp.py:84-89)

    def __init__(self, num_nodes: int, in_channels, hidden_channels: int, out_channels: int,
                 num_layers: int, alpha: float, dropout: float = 0.0,
                 pool_size: Optional[int] = None, buffer_size: Optional[int] = None, device=None):
        super().__init__(num_nodes, out_channels, num_layers, pool_size, buffer_size, device,
                         in_channels=in_channels)
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.alpha = alpha
        self.dropout = dropout

        self.lins = ModuleList()
        self.lins.append(RowsLinear(in_channels, hidden_channels))
        self.lins.append(RowsLinear(hidden_channels, out_channels))

        self.reg_modules = self.lins[:1]
        self.nonreg_modules = self.lins[1:]

    def reset_parameters(self):
        super().reset_parameters()
        for lin in self.lins:
            lin.reset_parameters()

    # ---- hooks (models/appnp.py:141-153)
    def _embed(self, x: Tensor, state: Dict[str, Any], batch_size: int) -> Tensor:
        x = F.dropout(x, p=self.dropout, training=self.training)
        on_gpu = x.is_cuda and x.dim() == 2 and x.dtype == torch.float32
        if on_gpu and self.lins[0].bias is not None:
            x = embed_relu(self.lins[0], x, 0)[0]      # bias and relu in the GEMM's epilogue
        else:
            x = self.lins[0](x).relu()
        x = F.dropout(x, p=self.dropout, training=self.training)
        x = self.lins[1](x)
        state['x_0'] = x[:batch_size]
        if on_gpu and self.num_layers <= 16:
            # alpha * x_0 computed once for all propagation steps, its gradient summed over
            # them in one launch
            state['x_0_terms'] = scaled_fan_out(state['x_0'], self.alpha, self.num_layers)
        return x

    def _update(self, layer: int, agg: Tensor, x: Tensor, state: Dict[str, Any],
                pre: bool = False) -> Tensor:
        return (1 - self.alpha) * agg + self.alpha * state['x_0'][:agg.size(0)]

    def _mix(self, layer: int, state: Dict[str, Any], batch_size: int):
        terms = state.get('x_0_terms')
        if terms is not None:
            return 1.0 - self.alpha, terms[layer], 1.0
        return 1.0 - self.alpha, state['x_0'], self.alpha

    def _update_mixed(self, layer: int, mixed: Tensor, x: Tensor, state: Dict[str, Any]) -> Tensor:
        return mixed
