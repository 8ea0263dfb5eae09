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


def bfs_clusters(adj_t: SparseTensor, num_parts: int, seed: int = 0, refine_sweeps: int = 0,
                 imbalance: float = 1.03) -> Tensor:
    """Deterministic stand-in partitioner (native, host): fills ``num_parts`` equal-sized
    blocks one after the other by breadth-first search (restarting from an unassigned node
    whenever the frontier dries up), so that blocks are unions of connected neighbourhoods;
    ``refine_sweeps`` > 0 then moves boundary nodes to the part holding most of their
    neighbours while the edge cut drops and no part exceeds ``imbalance`` x the mean size."""
    from . import ops
    rowptr, col, _ = adj_t.csr()
    n = adj_t.size(0)
    seeds = torch.randperm(n, generator=torch.Generator().manual_seed(seed))
    cluster = ops.partition_bfs(rowptr, col, num_parts, seeds)
    if refine_sweeps > 0:
        limit = int(-(-n // num_parts) * imbalance) + 1
        cluster, _ = ops.partition_refine(rowptr, col, num_parts, cluster, limit, refine_sweeps)
    return cluster


def edge_cut(adj_t: SparseTensor, cluster: Tensor) -> int:
    """Number of stored edges whose endpoints lie in different parts."""
    row, col, _ = adj_t.coo()
    return int((cluster[row] != cluster[col]).sum())


# ---- files
def write_metis_graph(adj_t: SparseTensor, path: str) -> None:
    """Writes the (symmetric, loop-free part of the) adjacency in the METIS graph format
    (first line ``n m``; line ``i`` lists the 1-based neighbours of node ``i``), the input
    `gpmetis` takes, so that a real METIS run can produce the partition elsewhere."""
    rowptr, col, _ = adj_t.csr()
    n = adj_t.size(0)
    row = adj_t.storage.row()
    keep = row != col
    col_k, row_k = col[keep], row[keep]
    counts = torch.bincount(row_k, minlength=n)
    if int(keep.sum()) % 2:
        raise ValueError("the METIS format needs a symmetric adjacency (odd number of off-diagonal entries)")
    with open(path, "w") as f:
        f.write(f"{n} {int(keep.sum()) // 2}\n")
        pos = 0
        cols = (col_k + 1).tolist()
        for c in counts.tolist():
            f.write(" ".join(map(str, cols[pos:pos + c])) + "\n")
            pos += c


def read_metis_partition(path: str, num_nodes: Optional[int] = None) -> Tensor:
    """Reads a `gpmetis` output file (``<graph>.part.<k>``: one part id per line, node order)
    into the cluster vector that :func:`metis` accepts."""
    with open(path) as f:
        ids = [int(line) for line in f if line.strip()]
    cluster = torch.tensor(ids, dtype=torch.int64)
    if num_nodes is not None and cluster.numel() != num_nodes:
        raise ValueError(f"{path} lists {cluster.numel()} nodes, expected {num_nodes}")
    if cluster.numel() and int(cluster.min()) < 0:
        raise ValueError("negative part id")
    return cluster


def save_partition(path: str, perm: Tensor, ptr: Tensor) -> None:
    """Stores ``(perm, ptr)`` as a plain ``.npz`` (two int64 arrays, no pickled objects)."""
    import numpy as np
    with open(path, "wb") as f:
        np.savez(f, perm=perm.cpu().numpy().astype("int64"), ptr=ptr.cpu().numpy().astype("int64"))


def load_partition(path: str, num_nodes: Optional[int] = None) -> Tuple[Tensor, Tensor]:
    """Loads and validates a partition written by :func:`save_partition`: ``perm`` must be a
    permutation, ``ptr`` non-decreasing from 0 to the number of nodes."""
    import numpy as np
    with np.load(path, allow_pickle=False) as z:
        perm, ptr = torch.from_numpy(z["perm"]), torch.from_numpy(z["ptr"])
    n = perm.numel()
    if num_nodes is not None and n != num_nodes:
        raise ValueError(f"partition is for {n} nodes, expected {num_nodes}")
    if not torch.equal(torch.sort(perm).values, torch.arange(n)):
        raise ValueError("perm is not a permutation")
    if ptr.numel() < 2 or int(ptr[0]) != 0 or int(ptr[-1]) != n or bool((ptr[1:] < ptr[:-1]).any()):
        raise ValueError("ptr must rise from 0 to the number of nodes")
    return perm, ptr


def metis(adj_t: SparseTensor, num_parts: int, recursive: bool = False, log: bool = True,
          cluster: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    r"""Computes the :obj:`(perm, ptr)` partitioning of the graph.  ``cluster`` lets a
    partition computed by an external METIS run (:func:`read_metis_partition`) be plugged in
    unchanged; without it the native stand-in partitioner runs (BFS blocks + refinement)."""
    if num_parts <= 1:
        perm = torch.arange(adj_t.sparse_size(0))
        ptr = torch.tensor([0, adj_t.sparse_size(0)])
        return perm, ptr
    if cluster is None:
        cluster = bfs_clusters(adj_t, num_parts, refine_sweeps=4)
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
