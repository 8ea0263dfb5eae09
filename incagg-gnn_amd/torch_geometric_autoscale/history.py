"""Per-layer table of historical node embeddings (interface of the reference's
``History``, history.py:9-74), with pull / push running as HIP row kernels.

Where the table lives decides the data path:

* ``device='cuda:i'``  (the MI355X default: 288 GB of HBM hold every table of an
  ogbn-products-scale model) -- pull is a row gather, push a row scatter / chunk
  copy, all inside HBM.
* ``device=None`` / ``'cpu'`` -- the table is pinned host memory, as in the reference.
  With the module on a GPU, the same kernels read / write the pinned table directly
  over the host link (no staging copy through pageable memory); with everything on
  the CPU the table is indexed with plain tensor ops (host plumbing, no GPU involved).
"""
from typing import Optional

import torch
from torch import Tensor

from . import ops


class History(torch.nn.Module):
    r"""A historical embedding storage module."""

    def __init__(self, num_embeddings: int, embedding_dim: int, device=None):
        super().__init__()
        self.num_embeddings = num_embeddings
        self.embedding_dim = embedding_dim

        on_host = device is None or str(device) == "cpu"
        pin_memory = on_host and torch.cuda.is_available()
        # Deliberately a plain attribute, not a buffer: it never enters a state_dict
        # and `.to()` does not move it (history.py:18,28-31).
        # and, in HBM, with the row pitch the aggregation kernels want for matrices they gather
        # from (ops.row_pitch): `emb` is then a [N, dim] view of a slightly wider allocation.
        if on_host:
            self.emb = torch.empty(num_embeddings, embedding_dim, device=device,
                                   pin_memory=pin_memory)
        else:
            self.emb = ops.rows_empty(num_embeddings, embedding_dim, device)
        # partition-parallel runs (reserve_tail): rows received from peers are parked right
        # behind the table's own rows, in the same allocation
        self.tail_rows = 0                 # rows per tail region
        self.tail_regions = 0
        self.emb_with_tail = self.emb      # [num_embeddings + tail_regions * tail_rows, dim]
        self._base = None                  # the contiguous [rows, pitch] matrix behind it
        self._device = torch.device("cpu")
        self.reset_parameters()

    @torch.no_grad()
    def reserve_tail(self, rows: int, regions: int = 1) -> None:
        """Room for ``regions`` x ``rows`` foreign rows behind the table's own (``emb`` keeps its
        contents and shape; it moves to a new allocation).  A partition-parallel step receives
        the out-of-batch rows its peers own there, so that the aggregation can read own and
        foreign out-of-batch rows from ONE matrix in place.  Several regions let the rows of the
        next step's batch arrive while the current step still reads its own
        (distributed.HaloPipeline); region ``r`` starts at row ``num_embeddings + r * rows``."""
        regions = max(1, int(regions))
        if rows <= self.tail_rows and regions <= self.tail_regions and self._base is not None:
            return
        rows, regions = max(rows, self.tail_rows), max(regions, self.tail_regions)
        n, dim = self.num_embeddings, self.embedding_dim
        if self.emb.is_cuda:
            full, base = ops.rows_empty(n + regions * rows, dim, self.emb.device, self.emb.dtype,
                                        with_base=True)
        else:
            full = base = torch.empty((n + regions * rows, dim), dtype=self.emb.dtype,
                                      pin_memory=self.emb.is_pinned())
        full[:n] = self.emb
        full[n:] = 0
        self.emb, self.emb_with_tail, self._base = full[:n], full, base
        self.tail_rows, self.tail_regions = rows, regions

    def reset_parameters(self):
        self.emb.fill_(0)

    def _apply(self, fn, *args, **kwargs):
        # Only re-target the compute device; `emb` stays where it was created.
        self._device = fn(torch.zeros(1)).device
        return self

    def _gpu_path(self, other_is_cuda: bool = False) -> bool:
        return self.emb.is_cuda or (self.emb.is_pinned() and
                                    (self._device.type == "cuda" or other_is_cuda))

    @torch.no_grad()
    def pull(self, n_id: Optional[Tensor] = None) -> Tensor:
        out = self.emb
        if n_id is None:
            return out.to(device=self._device)
        if self.emb.is_cuda:
            assert n_id.device == self.emb.device
            return ops.gather_rows(self.emb, n_id.contiguous()).to(device=self._device)
        if self._gpu_path(n_id.is_cuda):
            dev = self._device if self._device.type == "cuda" else n_id.device
            return ops.gather_rows(self.emb, n_id.to(dev).contiguous(), device=dev)
        assert n_id.device == self.emb.device
        return out.index_select(0, n_id).to(device=self._device)

    @torch.no_grad()
    def push(self, x, n_id: Optional[Tensor] = None, offset: Optional[Tensor] = None,
             count: Optional[Tensor] = None, relu: bool = False):
        """``relu``: ``x`` holds pre-activations and ``relu(x)`` is what gets stored (folded
        into the chunk copy on the GPU path)."""
        if relu and not (offset is not None and count is not None and x.is_cuda
                         and self._gpu_path(True)):
            x = torch.relu(x)
            relu = False
        if n_id is None and x.size(0) != self.num_embeddings:
            raise ValueError

        elif n_id is None and x.size(0) == self.num_embeddings:
            self.emb.copy_(x)

        elif offset is None or count is None:
            if x.is_cuda and self._gpu_path(True):
                ops.scatter_rows(self.emb, n_id.to(x.device).contiguous(), x.contiguous())
            else:
                assert n_id.device == self.emb.device
                self.emb[n_id] = x.to(self.emb.device)

        else:  # Push in chunks:
            if x.is_cuda and self._gpu_path(True):
                ops.copy_row_chunks(self.emb, x if x.stride(-1) == 1 else x.contiguous(), offset,
                                    count, packed_is_src=True, device=x.device, relu=relu)
            else:
                src_o = 0
                x = x.to(self.emb.device)
                for dst_o, c, in zip(offset.tolist(), count.tolist()):
                    self.emb[dst_o:dst_o + c] = x[src_o:src_o + c]
                    src_o += c

    @torch.no_grad()
    def pull_chunks(self, offset: Tensor, count: Tensor, out: Optional[Tensor] = None) -> Tensor:
        """Rows of whole partitions, packed: cat([emb[o:o+c] for o, c in zip(offset, count)])."""
        total = int(count.sum())
        dev = self._device if self._device.type == "cuda" else self.emb.device
        if dev.type != "cuda":
            return torch.cat([self.emb[o:o + c] for o, c in zip(offset.tolist(), count.tolist())])
        if out is None:     # at the row pitch the aggregation kernels want for their operands
            out = ops.rows_empty(total, self.embedding_dim, dev, self.emb.dtype)
        ops.copy_row_chunks(out, self.emb, offset, count, packed_is_src=False, device=dev)
        return out[:total]

    def forward(self, *args, **kwargs):
        """"""
        raise NotImplementedError

    def __repr__(self) -> str:
        return (f'{self.__class__.__name__}({self.num_embeddings}, '
                f'{self.embedding_dim}, emb_device={self.emb.device}, '
                f'device={self._device})')
