"""MI355X-native drop-in for the ``torch_geometric_autoscale`` package of AMShek/IncAgg-GNN.

Importing the package loads ``libincagg_gnn.so`` (hand-written HIP kernels behind a C ABI)
and registers the reference's five native operators in
``torch.ops.torch_geometric_autoscale`` (torch_geometric_autoscale/__init__.py:8-10 loads
``_relabel`` / ``_async`` for the same purpose).
"""
import importlib.machinery
import os.path as _osp

import torch

from . import _lib
from . import ops

_lib.lib()  # fail loudly at import time when the native library has not been built

# The operator objects `_relabel` / `_async` are found next to this file and loaded exactly as
# the reference does (torch_geometric_autoscale/__init__.py:8-10); `csrc/build_torch_ops.py`
# builds them.  A tree where they have not been built registers the same five operators from
# Python instead (same schemas, same C ABI underneath).
_specs = [importlib.machinery.PathFinder().find_spec(_name, [_osp.dirname(__file__)])
          for _name in ('_relabel', '_async')]
OPERATOR_OBJECTS = [_s.origin for _s in _specs if _s is not None]
if len(OPERATOR_OBJECTS) == 2 and not ops.torch_ops_defined():
    for _origin in OPERATOR_OBJECTS:
        torch.ops.load_library(_origin)
else:
    OPERATOR_OBJECTS = []
    ops.register_torch_ops()

from .sparse import SparseTensor  # noqa: E402
from .data import Data, get_data, gcn_norm, synthetic_graph  # noqa: E402
from .history import History  # noqa: E402
from .pool import AsyncIOPool  # noqa: E402
from .metis import metis, permute  # noqa: E402
from .utils import compute_micro_f1, dropout, gen_masks, index2mask  # noqa: E402
from .loader import (SubgraphLoader, EvalSubgraphLoader, DeviceSubgraphLoader, SubData,  # noqa: E402
                     partition_batch, device_partition_batch)
from .aggregate import spmm, incagg_aggregate  # noqa: E402
from . import models  # noqa: E402
from . import distributed  # noqa: E402

__version__ = '0.1.0'

__all__ = [
    'SparseTensor', 'Data', 'get_data', 'gcn_norm', 'synthetic_graph', 'History', 'AsyncIOPool',
    'metis', 'permute', 'compute_micro_f1', 'dropout', 'gen_masks', 'index2mask',
    'SubgraphLoader', 'EvalSubgraphLoader', 'SubData', 'spmm', 'incagg_aggregate', 'models',
    '__version__',
]
