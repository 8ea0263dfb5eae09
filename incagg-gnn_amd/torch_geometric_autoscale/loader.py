"""Partition mini-batch construction (interface of loader.py:95-284): every batch is a
union of whole partitions plus, for the push/pull and sweep paths, its one-hop halo.

A batch is the tuple ``SubData(data, batch_size, n_id, offset, count)``:
``data.adj_t`` is the local ``[B, M]`` (or ``[B, B]``) CSR produced by the native
relabelling, ``data.x = x[n_id]``, ``offset`` / ``count`` the contiguous global row
ranges of the batch's partitions.  The transposed CSR for the backward pass and the
in-batch / halo split are derived once per batch object, never per training step.
"""
import time
from typing import List, NamedTuple, Tuple

import torch
from torch import Tensor
from torch.utils.data import DataLoader

from . import ops
from .data import Data
from .sparse import SparseTensor


def _relabel_one_hop(*args):
    """Bound at call time to the registered operator (loader.py:23 of the reference)."""
    return torch.ops.torch_geometric_autoscale.relabel_one_hop(*args)


def _relabel_one_hop_within_batch(*args):
    return torch.ops.torch_geometric_autoscale.relabel_one_hop_within_batch(*args)


class SubData(NamedTuple):
    data: Data
    batch_size: int
    n_id: Tensor  # The indices of mini-batched nodes
    offset: Tensor  # The offset of contiguous mini-batched nodes
    count: Tensor  # The number of contiguous mini-batched nodes

    def to(self, *args, **kwargs):
        return SubData(self.data.to(*args, **kwargs), self.batch_size, self.n_id, self.offset,
                       self.count)


def sample_neighbors(src: SparseTensor, num_neighbors: int,
                     unsampled_n_id: Tensor = None) -> Tuple[SparseTensor, Tensor]:
    """Keeps at most ``num_neighbors`` random edges per row, columns in their original
    order (loader.py:32-93); ``num_neighbors < 0`` keeps everything.  The node ids are
    returned unchanged: out-of-batch nodes that lost all their edges keep their columns.
    Runs where ``src`` lives (a batch built on the GPU is sampled there, and the result is
    again a device-format adjacency)."""
    if num_neighbors < 0:
        return src, unsampled_n_id
    rowptr, col, value = src.csr()
    dev = col.device
    deg = rowptr[1:] - rowptr[:-1]
    # random key per edge; keep the `num_neighbors` smallest of every row, order preserved
    key = torch.rand(col.numel(), device=dev, dtype=torch.float64)
    row = src.storage.row()
    order = torch.argsort(row.to(torch.float64) + key * 0.999999, stable=True)
    rank = torch.empty_like(order)
    rank[order] = torch.arange(order.numel(), device=dev)
    keep = (rank - rowptr[row]) < num_neighbors
    new_deg = torch.minimum(deg, torch.full_like(deg, num_neighbors))
    new_rowptr = torch.zeros_like(rowptr)
    new_rowptr[1:] = torch.cumsum(new_deg, 0)
    new_val = None if value is None else value[keep]
    if dev.type == "cuda":
        out = SparseTensor.from_device_csr(new_rowptr.to(torch.int32), col[keep].to(torch.int32),
                                           new_val, src.sparse_sizes())
    else:
        out = SparseTensor(rowptr=new_rowptr, col=col[keep], value=new_val,
                           sparse_sizes=src.sparse_sizes(), is_sorted=True)
    return out, unsampled_n_id


def partition_batch(data: Data, ptr: Tensor, part_ids, in_batch_only: bool = False,
                    bipartite: bool = True) -> SubData:
    """The batch made of the partitions ``part_ids`` (ids into ``ptr``): their rows, and
    unless ``in_batch_only`` their one-hop halo.  What the loaders' collate functions
    produce, callable directly."""
    part_ids = [int(p) for p in part_ids]
    n_id = torch.cat([torch.arange(int(ptr[p]), int(ptr[p + 1])) for p in part_ids])
    batch_id = torch.tensor(part_ids)
    offset = ptr[batch_id]
    count = ptr[batch_id + 1] - offset
    rowptr, col, value = data.adj_t.csr()
    fn = _relabel_one_hop_within_batch if in_batch_only else _relabel_one_hop
    batch_size = n_id.numel()
    rowptr, col, value, n_id = fn(rowptr, col, value, n_id, bipartite)
    adj_t = SparseTensor(rowptr=rowptr, col=col, value=value,
                         sparse_sizes=(rowptr.numel() - 1, n_id.numel()), is_sorted=True)
    out = data.__class__(adj_t=adj_t)
    for k, v in data:
        if isinstance(v, Tensor) and v.dim() > 0 and v.size(0) == data.num_nodes:
            out[k] = v.index_select(0, n_id)
    return SubData(out, batch_size, n_id, offset, count)


def _device_graph(adj_t: SparseTensor, device):
    """The global CSR resident in HBM (int64 indices as the relabelling takes them, fp32
    values), uploaded once per graph and device."""
    cache = adj_t.__dict__.setdefault("_device_graph", {})
    device = torch.device(device)
    if device not in cache:
        rowptr, col, value = adj_t.csr()
        cache[device] = (rowptr.to(device), col.to(device),
                         None if value is None else value.to(device, torch.float32))
    return cache[device]


def device_partition_batch(data: Data, ptr: Tensor, part_ids, device,
                           in_batch_only: bool = False) -> SubData:
    """:func:`partition_batch` with the batch built on the GPU: the relabelling
    (csrc/cpu/relabel_cpu.cpp) runs as HIP kernels over the HBM-resident graph and emits the
    device-format adjacency directly; same ``n_id`` order, same local ids, same values as
    the host path, element for element.  ``n_id`` and ``data.adj_t`` live on ``device``;
    node-level tensors of ``data`` are indexed where they live."""
    device = torch.device(device)
    part_ids = [int(p) for p in part_ids]
    batch_id = torch.tensor(part_ids)
    offset = ptr[batch_id]
    count = ptr[batch_id + 1] - offset
    # the node ids of the partitions, one after another, in three device operations (an arange
    # per partition costs a launch each: 100 of them for a reddit-sized batch)
    total = int(count.sum())
    first = (offset - (count.cumsum(0) - count)).to(device)
    idx = torch.repeat_interleave(first, count.to(device), output_size=total) \
        + torch.arange(total, device=device)
    rowptr, col, value = _device_graph(data.adj_t, device)
    # the batch's rows are whole partitions, i.e. ranges of the global CSR: their edge count (the
    # capacity of the output) comes from the host copy of the row pointers -- no device pass, no wait
    bound, host_rowptr = None, data.adj_t.csr()[0]
    if not host_rowptr.is_cuda:
        bound = int(sum(int(host_rowptr[int(ptr[p + 1])]) - int(host_rowptr[int(ptr[p])])
                        for p in part_ids))
    rp32, c32, val, n_id, _ = ops.relabel_one_hop_device(rowptr, col, value, idx,
                                                         within_batch=in_batch_only,
                                                         edge_capacity=bound)
    batch_size = idx.numel()
    adj_t = SparseTensor.from_device_csr(rp32, c32, val, (batch_size, n_id.numel()))
    out = data.__class__(adj_t=adj_t)
    for k, v in data:
        if isinstance(v, Tensor) and v.dim() > 0 and v.size(0) == data.num_nodes:
            out[k] = v.index_select(0, n_id.to(v.device))
    return SubData(out, batch_size, n_id, offset, count)


class DeviceSubgraphLoader:
    """Partition batches built on the GPU (:func:`device_partition_batch`): the counterpart of
    :class:`SubgraphLoader` / :class:`EvalSubgraphLoader` for a graph that is resident in HBM.
    Iterating yields ``SubData`` tuples for consecutive groups of ``batch_size`` partitions, in
    a fresh random order per pass when ``shuffle``.  Batches of fixed groups (no shuffling, or
    one partition per batch) are built once and kept -- with their transposed CSR, row orders
    and in-place column ids, which hang off the adjacency object."""

    def __init__(self, data: Data, ptr: Tensor, device, batch_size: int = 1, shuffle: bool = False,
                 in_batch_only: bool = False, num_neighbors: int = -1):
        self.data, self.ptr, self.device = data, ptr, torch.device(device)
        self.batch_size, self.shuffle, self.in_batch_only = batch_size, shuffle, in_batch_only
        self.num_parts = ptr.numel() - 1
        # per-row neighbour sampling (loader.py:217-246): a fresh sample of the kept batch at
        # every visit, drawn on the GPU; such batches are never the same object twice
        self.num_neighbors = -1 if num_neighbors is None else int(num_neighbors)
        self._cache = {} if (not shuffle or batch_size == 1) else None

    def __len__(self) -> int:
        return -(-self.num_parts // self.batch_size)

    @property
    def static_batches(self) -> bool:
        """Whether a group of partitions always comes back as the SAME batch object (steps over
        such batches can be recorded once and replayed)."""
        return self._cache is not None and self.num_neighbors < 0

    def _sampled(self, sub: SubData) -> SubData:
        if self.num_neighbors < 0:
            return sub
        adj_t, n_id = sample_neighbors(sub.data.adj_t, self.num_neighbors, sub.n_id)
        out = sub.data.__class__(**{k: v for k, v in sub.data if k != "adj_t"})
        out.adj_t = adj_t
        return SubData(out, sub.batch_size, n_id, sub.offset, sub.count)

    def _build(self, parts) -> SubData:
        if self._cache is None:
            return device_partition_batch(self.data, self.ptr, parts, self.device, self.in_batch_only)
        key = tuple(parts)
        if key not in self._cache:
            self._cache[key] = device_partition_batch(self.data, self.ptr, parts, self.device,
                                                      self.in_batch_only)
        return self._cache[key]

    def __iter__(self):
        order = torch.randperm(self.num_parts).tolist() if self.shuffle else list(range(self.num_parts))
        for i in range(0, self.num_parts, self.batch_size):
            yield self._sampled(self._build(order[i:i + self.batch_size]))


class SubgraphLoader(DataLoader):
    r"""A simple subgraph loader that, given a pre-partioned :obj:`data` object,
    generates subgraphs from mini-batches in :obj:`ptr` (including their 1-hop
    neighbors)."""

    def __init__(self, data: Data, ptr: Tensor, batch_size: int = 1, bipartite: bool = True,
                 log: bool = True, num_neighbors=-1, type='eval', IB=False, **kwargs):
        self.data = data
        self.ptr = ptr
        self.bipartite = bipartite
        self.log = log
        self.num_neighbors = num_neighbors
        if 'shuffle' in kwargs:
            self.shuffle = kwargs['shuffle']
        self.shuffled_batch_id = []

        n_id = torch.arange(data.num_nodes)
        batches = n_id.split((ptr[1:] - ptr[:-1]).tolist())
        batches = [(i, batches[i]) for i in range(len(batches))]

        if type == 'train':
            collate = self.compute_subgraph_IB if IB else self.compute_subgraph_NS
            super().__init__(batches, batch_size=batch_size, collate_fn=collate, **kwargs)
        elif batch_size > 1:
            super().__init__(batches, batch_size=batch_size, collate_fn=self.compute_subgraph,
                             **kwargs)
        else:  # If `batch_size=1`, we pre-process the subgraph generation:
            if log:
                t = time.perf_counter()
                print('Pre-processing subgraphs...', end=' ', flush=True)
            data_list = list(DataLoader(batches, batch_size=batch_size,
                                        collate_fn=self.compute_subgraph, **kwargs))
            if log:
                print(f'Done! [{time.perf_counter() - t:.2f}s]')
            super().__init__(data_list, batch_size=batch_size, collate_fn=lambda x: x[0], **kwargs)

    def _batch_header(self, batches: List[Tuple[int, Tensor]]):
        batch_ids, n_ids = zip(*batches)
        n_id = torch.cat(n_ids, dim=0)
        batch_id = torch.tensor(batch_ids)
        offset = self.ptr[batch_id]
        count = self.ptr[batch_id + 1] - offset
        return n_id, n_id.numel(), offset, count

    def _finish(self, rowptr, col, value, n_id, batch_size, offset, count) -> SubData:
        adj_t = SparseTensor(rowptr=rowptr, col=col, value=value,
                             sparse_sizes=(rowptr.numel() - 1, n_id.numel()), is_sorted=True)
        data = self.data.__class__(adj_t=adj_t)
        for k, v in self.data:
            if isinstance(v, Tensor) and v.dim() > 0 and v.size(0) == self.data.num_nodes:
                data[k] = v.index_select(0, n_id)
        return SubData(data, batch_size, n_id, offset, count)

    def compute_subgraph(self, batches: List[Tuple[int, Tensor]]) -> SubData:
        """Batch rows plus their full one-hop halo (loader.py:172-192)."""
        n_id, batch_size, offset, count = self._batch_header(batches)
        rowptr, col, value = self.data.adj_t.csr()
        rowptr, col, value, n_id = _relabel_one_hop(rowptr, col, value, n_id, self.bipartite)
        return self._finish(rowptr, col, value, n_id, batch_size, offset, count)

    def compute_subgraph_IB(self, batches: List[Tuple[int, Tensor]]) -> SubData:
        """In-batch edges only -- the batches of the incremental-aggregation training
        path (loader.py:194-214)."""
        n_id, batch_size, offset, count = self._batch_header(batches)
        rowptr, col, value = self.data.adj_t.csr()
        rowptr, col, value, n_id = _relabel_one_hop_within_batch(rowptr, col, value, n_id,
                                                                    self.bipartite)
        return self._finish(rowptr, col, value, n_id, batch_size, offset, count)

    def compute_subgraph_NS(self, batches: List[Tuple[int, Tensor]]) -> SubData:
        """One-hop batch with optional per-row neighbour sampling (loader.py:217-246)."""
        n_id, batch_size, offset, count = self._batch_header(batches)
        rowptr, col, value = self.data.adj_t.csr()
        rowptr, col, value, n_id = _relabel_one_hop(rowptr, col, value, n_id, self.bipartite)
        if self.num_neighbors is not None and self.num_neighbors >= 0:
            adj_t = SparseTensor(rowptr=rowptr, col=col, value=value,
                                 sparse_sizes=(rowptr.numel() - 1, n_id.numel()), is_sorted=True)
            adj_t, n_id = sample_neighbors(adj_t, self.num_neighbors, n_id)
            rowptr, col, value = adj_t.csr()
        return self._finish(rowptr, col, value, n_id, batch_size, offset, count)

    def __repr__(self):
        return f'{self.__class__.__name__}()'


class EvalSubgraphLoader(SubgraphLoader):
    r"""Fixed-order loader over consecutive groups of ``batch_size`` partitions, with the
    subgraphs built once up front (loader.py:266-284)."""

    def __init__(self, data: Data, ptr: Tensor, batch_size: int = 1, bipartite: bool = True,
                 log: bool = True, **kwargs):
        ptr = ptr[::batch_size]
        if int(ptr[-1]) != data.num_nodes:
            ptr = torch.cat([ptr, torch.tensor([data.num_nodes])], dim=0)
        super().__init__(data=data, ptr=ptr, batch_size=1, bipartite=bipartite, log=log,
                         shuffle=False, num_workers=0, **kwargs)
