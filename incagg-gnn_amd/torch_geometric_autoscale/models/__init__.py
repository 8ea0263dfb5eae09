from .base import ScalableGNN
from .gcn import GCN
from .gcn2 import GCN2
from .appnp import APPNP
from .graphsage import GraphSAGE

__all__ = ['ScalableGNN', 'GCN', 'GCN2', 'APPNP', 'GraphSAGE']
