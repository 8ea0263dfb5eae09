"""A small ``Data`` container (the slice of ``torch_geometric.data.Data`` the loaders use,
loader.py:187-190) and seeded synthetic graphs shaped like the benchmark datasets.

Datasets are never downloaded: ``get_data`` builds a synthetic graph with the public
size statistics of the named dataset (optionally scaled down), planted in ``num_parts``
communities so that a partition-contiguous ``(perm, ptr)`` exists by construction.
"""
from typing import Dict, Iterator, Optional, Tuple

import torch
from torch import Tensor

from .sparse import SparseTensor


class Data:
    """Attribute bag; tensors whose first dimension equals ``num_nodes`` are node-level."""

    def __init__(self, **kwargs):
        self.__dict__['_store'] = {}
        for k, v in kwargs.items():
            self._store[k] = v

    def __getattr__(self, key):
        store = self.__dict__['_store']
        if key in store:
            return store[key]
        raise AttributeError(key)

    def __setattr__(self, key, value):
        self._store[key] = value

    def __getitem__(self, key):
        return self._store[key]

    def __setitem__(self, key, value):
        self._store[key] = value

    def __contains__(self, key):
        return key in self._store

    def __iter__(self) -> Iterator[Tuple[str, object]]:
        return iter(list(self._store.items()))

    def keys(self):
        return list(self._store.keys())

    @property
    def num_nodes(self) -> int:
        if 'num_nodes' in self._store:
            return self._store['num_nodes']
        if 'x' in self._store and self._store['x'] is not None:
            return self._store['x'].size(0)
        if 'adj_t' in self._store:
            return self._store['adj_t'].size(1)
        raise AttributeError('num_nodes')

    def to(self, device, *args, **kwargs) -> "Data":
        out = Data()
        for k, v in self._store.items():
            out._store[k] = v.to(device, *args, **kwargs) if hasattr(v, 'to') else v
        return out

    def __repr__(self):
        parts = []
        for k, v in self._store.items():
            parts.append(f"{k}={list(v.shape)}" if isinstance(v, Tensor) else f"{k}={v}")
        return f"Data({', '.join(parts)})"


# Public size statistics of the benchmark graphs (nodes, directed edges after
# symmetrisation, input width, classes).
DATASET_STATS: Dict[str, Tuple[int, int, int, int]] = {
    'flickr': (89_250, 899_756, 500, 7),
    'arxiv': (169_343, 2_315_598, 128, 40),
    'products': (2_449_029, 123_718_280, 100, 47),
    'reddit': (232_965, 23_213_838, 602, 41),
}


def synthetic_graph(num_nodes: int, num_edges: int, num_parts: int, cross_fraction: float = 0.15,
                    exponent: float = 2.1, seed: int = 0, max_degree_fraction: float = 0.02,
                    device="cpu", keep_on_device: bool = False) -> Tuple[SparseTensor, Tensor]:
    """Symmetric power-law graph planted in ``num_parts`` contiguous communities.

    Returns ``(adj_t, ptr)`` (on the host unless ``keep_on_device``): a value-less CSR without self loops or
    duplicates, and ``ptr[num_parts + 1]`` community boundaries (part ``p`` owns ids
    ``[ptr[p], ptr[p+1])``, the contract of metis.py:32-33 after ``permute``).
    Chung-Lu style: both endpoints of an edge are drawn with probability proportional to
    a power-law node weight; ``cross_fraction`` of the edges may leave the source's
    community.  ``device`` only selects where the sampling runs (a GPU makes the
    124M-edge ogbn-products-sized graph a matter of seconds); the random stream, and so
    the graph, depends on it.
    """
    device = torch.device(device)
    g = torch.Generator(device=device).manual_seed(seed)
    n = num_nodes

    def rand(k):
        return torch.rand(k, generator=g, device=device, dtype=torch.float64)

    bounds = (torch.arange(num_parts + 1, device=device, dtype=torch.int64) * n) // num_parts
    part_of = torch.repeat_interleave(torch.arange(num_parts, device=device), bounds[1:] - bounds[:-1])
    w = (1.0 - rand(n)).clamp_min(1e-12).pow(-1.0 / (exponent - 1.0))
    w = w.clamp(max=max(4.0, max_degree_fraction * n))
    cdf = torch.cumsum(w, 0)
    total = cdf[-1]
    def draw(half):
        src = torch.searchsorted(cdf, rand(half) * total).clamp(max=n - 1)
        cross = rand(half) < cross_fraction
        dst_any = torch.searchsorted(cdf, rand(half) * total).clamp(max=n - 1)
        p = part_of[src]
        lo, hi = bounds[p], bounds[p + 1]
        base = torch.where(lo > 0, cdf[(lo - 1).clamp(min=0)], torch.zeros_like(total))
        target = base + rand(half) * (cdf[hi - 1] - base)
        dst_local = torch.minimum(torch.maximum(torch.searchsorted(cdf, target), lo), hi - 1)
        dst = torch.where(cross, dst_any, dst_local)
        keep = src != dst
        src, dst = src[keep], dst[keep]
        return torch.unique(torch.cat([src * n + dst, dst * n + src]))

    # Duplicate draws (frequent between heavy nodes) are merged, so draw more pairs
    # until the de-duplicated edge count reaches the requested one.
    half = num_edges // 2
    key = draw(half)
    for _ in range(3):
        if key.numel() >= 0.97 * num_edges:
            break
        half = int(half * (num_edges / max(key.numel(), 1)) ** 1.25) + 1
        key = draw(half)
    row = torch.div(key, n, rounding_mode="floor")
    col = key - row * n
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    rowptr[1:] = torch.cumsum(torch.bincount(row, minlength=n), 0)
    if not keep_on_device:
        rowptr, col = rowptr.cpu(), col.cpu()
    adj_t = SparseTensor(rowptr=rowptr, col=col, sparse_sizes=(n, n), is_sorted=True)
    return adj_t, bounds.cpu()


def gcn_norm(adj_t: SparseTensor, add_self_loops: bool = False) -> SparseTensor:
    """``D^-1/2 A D^-1/2`` with ``deg`` = row sums and ``inf -> 0`` (main.py:150-151)."""
    if add_self_loops:
        adj_t = adj_t.set_diag()
    if not adj_t.has_value():
        adj_t = adj_t.fill_value(1.0)
    row, col, value = adj_t.coo()
    deg = torch.zeros(adj_t.size(0), dtype=value.dtype, device=value.device).index_add_(0, row, value)
    dis = deg.pow(-0.5)
    dis[dis == float('inf')] = 0
    return adj_t.set_value(dis[row] * value * dis[col])


def load_npz_dataset(path: str):
    """A graph stored as plain arrays: ``rowptr``, ``col`` (int64 CSR of the symmetric
    adjacency), ``x`` [N, F] float32, ``y`` [N] int64 (or [N, C] float for multi-label) and
    boolean ``train_mask`` / ``val_mask`` / ``test_mask``.  Nothing in the file is executed."""
    import numpy as np
    with np.load(path, allow_pickle=False) as z:
        arrays = {k: torch.from_numpy(z[k]) for k in z.files}
    for k in ("rowptr", "col", "x", "y"):
        if k not in arrays:
            raise ValueError(f"{path}: array '{k}' missing")
    n = arrays["rowptr"].numel() - 1
    adj_t = SparseTensor(rowptr=arrays["rowptr"].to(torch.int64), col=arrays["col"].to(torch.int64),
                         sparse_sizes=(n, n), is_sorted=True)
    data = Data(adj_t=adj_t, num_nodes=n, x=arrays["x"].to(torch.float32), y=arrays["y"])
    for k in ("train_mask", "val_mask", "test_mask"):
        if k in arrays:
            data[k] = arrays[k].to(torch.bool)
    y = data.y
    n_cls = int(y.max()) + 1 if y.dim() == 1 else y.size(1)
    return data, data.x.size(1), n_cls


def get_data(root: Optional[str], name: str, num_parts: int = 16, scale: float = 1.0,
             cross_fraction: float = 0.15, seed: int = 0, features: bool = True, device="cpu",
             keep_on_device: bool = False, planted_labels: bool = False):
    """``(data, in_channels, out_channels)`` like data.py:118-145.  If ``root/<name>.npz``
    exists it is loaded (:func:`load_npz_dataset`; datasets are never downloaded); otherwise
    a synthetic graph with the named dataset's statistics is generated (``scale`` shrinks
    nodes and edges together) and ``data.ptr`` holds the planted partition boundaries.
    ``planted_labels``: labels follow the planted communities and the features carry a noisy
    class signal, so that training has something to learn (default: random labels)."""
    import os
    if root is not None:
        path = os.path.join(root, f"{name}.npz")
        if os.path.isfile(path):
            return load_npz_dataset(path)
    key = name.lower().replace('ogbn-', '')
    if key not in DATASET_STATS:
        raise NotImplementedError(f"unknown dataset '{name}'")
    n, e, f_in, n_cls = DATASET_STATS[key]
    n, e = max(int(n * scale), num_parts * 4), max(int(e * scale), num_parts * 8)
    adj_t, ptr = synthetic_graph(n, e, num_parts, cross_fraction, seed=seed, device=device,
                                 keep_on_device=keep_on_device)
    g = torch.Generator().manual_seed(seed + 1)
    data = Data(adj_t=adj_t, num_nodes=n, ptr=ptr)
    if features:
        data.x = torch.randn(n, f_in, generator=g)
    data.y = torch.randint(0, n_cls, (n,), generator=g)
    if planted_labels:
        part = torch.repeat_interleave(torch.arange(num_parts), ptr[1:] - ptr[:-1])
        data.y = part % n_cls
        if features:
            centres = torch.randn(n_cls, f_in, generator=g)
            data.x = data.x + 0.5 * centres[data.y]
    split = torch.rand(n, generator=g)
    data.train_mask = split < 0.6
    data.val_mask = (split >= 0.6) & (split < 0.8)
    data.test_mask = split >= 0.8
    return data, f_in, n_cls
