"""Partition interface of metis.py:14-63: ``metis(adj_t, num_parts) -> (perm, ptr)`` and
``permute(data, perm)``.  The METIS library itself is not bundled; the contract that
matters downstream is kept -- after ``permute`` part ``p`` owns the contiguous id range
``[ptr[p], ptr[p+1])`` -- and partitions computed elsewhere can be supplied as a
cluster-assignment vector.
"""
from typing import Optional, Tuple

import torch
from torch import Tensor

from .data import Data
from .sparse import SparseTensor


def partition_from_clusters(cluster: Tensor, num_parts: int) -> Tuple[Tensor, Tensor]:
    """(perm, ptr) from a per-node part id, as metis.py:31-33 derives them."""
    cluster, perm = cluster.sort(stable=True)
    ptr = torch.zeros(num_parts + 1, dtype=torch.int64)
    ptr[1:] = torch.cumsum(torch.bincount(cluster, minlength=num_parts), 0)
    return perm, ptr


def bfs_clusters(adj_t: SparseTensor, num_parts: int, seed: int = 0) -> Tensor:
    """Deterministic stand-in partitioner: fills ``num_parts`` equal-sized blocks one after
    the other by breadth-first search (restarting from an unassigned node whenever the
    frontier dries up), so that blocks are unions of connected neighbourhoods."""
    from collections import deque
    rowptr, col, _ = adj_t.csr()
    rowptr, col = rowptr.tolist(), col.tolist()
    n = adj_t.size(0)
    target = -(-n // num_parts)
    cluster = [-1] * n
    seeds = torch.randperm(n, generator=torch.Generator().manual_seed(seed)).tolist()
    cursor, part, filled = 0, 0, 0
    queue = deque()
    for _ in range(n):
        if not queue:
            while cluster[seeds[cursor]] != -1:
                cursor += 1
            queue.append(seeds[cursor])
            cluster[seeds[cursor]] = -2  # queued
        v = queue.popleft()
        cluster[v] = part
        filled += 1
        if filled == target and part < num_parts - 1:
            for w in queue:          # un-queue: the next block starts afresh
                cluster[w] = -1
            queue.clear()
            part, filled = part + 1, 0
            continue
        for e in range(rowptr[v], rowptr[v + 1]):
            w = col[e]
            if cluster[w] == -1:
                cluster[w] = -2
                queue.append(w)
    return torch.tensor(cluster, dtype=torch.int64)


def metis(adj_t: SparseTensor, num_parts: int, recursive: bool = False, log: bool = True,
          cluster: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    r"""Computes the :obj:`(perm, ptr)` partitioning of the graph.  ``cluster`` lets a
    partition computed by an external METIS run be plugged in unchanged."""
    if num_parts <= 1:
        perm = torch.arange(adj_t.sparse_size(0))
        ptr = torch.tensor([0, adj_t.sparse_size(0)])
        return perm, ptr
    if cluster is None:
        cluster = bfs_clusters(adj_t, num_parts)
    return partition_from_clusters(cluster, num_parts)


def permute(data: Data, perm: Tensor, log: bool = True) -> Data:
    r"""Permutes a :obj:`data` object according to a given permutation (metis.py:43-63)."""
    out = Data()
    for key, value in data:
        if isinstance(value, Tensor) and value.size(0) == data.num_nodes:
            out[key] = value[perm]
        elif isinstance(value, SparseTensor) and value.sparse_sizes() == (data.num_nodes,
                                                                         data.num_nodes):
            out[key] = value.permute(perm)
        else:
            out[key] = value
    return out
