"""GraphSAGE (mean aggregation) on historical embeddings (constructor and module layout of
models/graphsage.py:33-108; the layer arithmetic is spelled out at :623-644)."""
from typing import Optional

from torch import Tensor

from ..nn import SAGEConv
from .gcn import GCN


class GraphSAGE(GCN):
    agg_reduce = "mean"

    def _make_conv(self, in_dim: int, out_dim: int):
        return SAGEConv(in_dim, out_dim, normalize=False)

    def _pre_aggregates(self, layer: int) -> bool:
        return False

    def _pre_aggregate(self, layer: int, x: Tensor) -> Optional[Tensor]:
        return None  # lin_l follows the mean; the root term needs the untouched input

    def _conv_update(self, layer: int, agg: Tensor, x: Tensor, pre: bool, push=None) -> Tensor:
        return self.convs[layer].update(agg, x)
