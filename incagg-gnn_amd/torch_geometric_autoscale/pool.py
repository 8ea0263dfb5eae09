"""Ring of staging slots between pinned-host History tables and HBM: the public surface and
the queue discipline of the reference's ``AsyncIOPool`` (pool.py:15-123).

A slot (:class:`_Slot`) owns a pull stream, a push stream, a pinned host bounce buffer and a
device buffer of ``[buffer_size, embedding_dim]``, all created on first use.  Pull requests
(:class:`_Pull`) are handed slots round-robin and wait in a FIFO; the first ``pool_size`` of
them are in flight, every ``free_pull`` starts the oldest one that is not.

How it moves the data differs from the reference (the rows delivered do not):

* nothing blocks the host on a stream -- consumers are ordered behind the copies with
  stream waits, and a slot's device buffer is refilled only behind the kernels that were
  queued on the compute stream while it was handed out;
* with ``zero_copy=True`` (default) the gathered part of a pull is a HIP kernel on the pull
  stream that reads the pinned table over the host link, instead of a single-threaded host
  gather into the bounce buffer followed by a copy (csrc/cuda/async_cuda.cu:95-110);
  contiguous chunks always go through the DMA engines (hipMemcpyAsync).
"""
from typing import Callable, List, NamedTuple, Optional

import torch
from torch import Tensor

from . import ops  # noqa: F401  (defines the operators when no operator object was loaded)


class _Native:
    """``torch.ops.torch_geometric_autoscale.{synchronize, read_async, write_async}``, bound at
    call time like pool.py:10-12 of the reference binds them: whichever object registered the
    operators (the loaded ``_async`` library, or the Python registration) is what runs."""

    def __getattr__(self, name):
        return getattr(torch.ops.torch_geometric_autoscale, name)


_native = _Native()


class _Pull(NamedTuple):
    slot: int
    src: Tensor
    offset: Optional[Tensor]
    count: Optional[Tensor]
    index: Tensor


class _Slot:
    """Resources of one ring entry; every member appears when it is first asked for."""

    def __init__(self, owner: "AsyncIOPool"):
        self._owner = owner
        self._streams = {}
        self._host: Optional[Tensor] = None
        self._dev: Optional[Tensor] = None
        self.pushed: Optional[Tensor] = None      # source of the push in flight, kept alive

    def stream(self, kind: str):
        if kind not in self._streams:
            dev = self._owner._device
            if dev.type != "cuda":
                raise RuntimeError("the staging pool moves rows to a GPU: call .to(device) first")
            self._streams[kind] = torch.cuda.Stream(dev)
        return self._streams[kind]

    def has_stream(self, kind: str) -> bool:
        return kind in self._streams

    def host_buffer(self) -> Tensor:
        if self._host is None:
            o = self._owner
            self._host = torch.empty(o.buffer_size, o.embedding_dim, pin_memory=True)
        return self._host

    def device_buffer(self) -> Tensor:
        if self._dev is None:
            o = self._owner
            if o._device.type != "cuda":
                raise RuntimeError("the staging pool moves rows to a GPU: call .to(device) first")
            self._dev = torch.empty(o.buffer_size, o.embedding_dim, device=o._device)
        return self._dev


class AsyncIOPool(torch.nn.Module):
    def __init__(self, pool_size: int, buffer_size: int, embedding_dim: int,
                 zero_copy: bool = True):
        super().__init__()
        self.pool_size = pool_size
        self.buffer_size = buffer_size
        self.embedding_dim = embedding_dim
        self.zero_copy = zero_copy
        self._device = torch.device('cpu')
        self._slots = [_Slot(self) for _ in range(pool_size)]
        self.pending: List[_Pull] = []      # FIFO of pulls issued and not yet freed
        self.last_pull_slot = -1            # round-robin cursors; -1 = ring idle
        self.last_push_slot = -1

    def _apply(self, fn: Callable, *args, **kwargs) -> None:
        self._device = fn(torch.zeros(1)).device
        return self

    # ------------------------------------------------------------------ pulls
    @torch.no_grad()
    def async_pull(self, src: Tensor, offset: Optional[Tensor], count: Optional[Tensor],
                   index: Tensor) -> None:
        """Queues the pull of ``src[offset_i : offset_i + count_i]`` followed by
        ``src[index]`` (pool.py:64-74); it starts right away while a slot is free."""
        self.last_pull_slot = (self.last_pull_slot + 1) % self.pool_size
        req = _Pull(self.last_pull_slot, src, offset, count, index)
        self.pending.append(req)
        if len(self.pending) <= self.pool_size:
            self._start_pull(req)

    @torch.no_grad()
    def _start_pull(self, req: _Pull) -> None:
        slot = self._slots[req.slot]
        stream, dst = slot.stream("pull"), slot.device_buffer()
        src, offset, count, index = req.src, req.offset, req.count, req.index
        # kernels already queued on the compute stream may still read this buffer
        stream.wait_stream(torch.cuda.current_stream(self._device))
        with torch.cuda.stream(stream):
            kernel_gather = (self.zero_copy and index.numel() > 0 and src.is_pinned()
                             and src.dtype == torch.float32)
            if not kernel_gather:
                _native.read_async(src, offset, count, index.cpu(), dst, slot.host_buffer())
                return
            n_chunk_rows = 0 if count is None else int(count.sum())
            if n_chunk_rows + index.numel() > self.buffer_size:
                raise RuntimeError("Buffer tensor size too small")
            _native.read_async(src, offset, count, torch.empty(0, dtype=torch.int64), dst,
                           slot.host_buffer())
            ids = index.to(self._device, non_blocking=True)
            ops.gather_rows(src, ids, out=dst[n_chunk_rows:n_chunk_rows + index.numel()],
                            device=self._device)

    @torch.no_grad()
    def synchronize_pull(self) -> Tensor:
        """The device buffer of the OLDEST pending pull, usable on the current stream
        (pool.py:83-88).  It stays valid until :meth:`free_pull`."""
        slot = self._slots[self.pending[0].slot]
        _native.synchronize()       # the worker has enqueued every copy of that pull
        torch.cuda.current_stream(self._device).wait_stream(slot.stream("pull"))
        return slot.device_buffer()

    @torch.no_grad()
    def free_pull(self) -> None:
        """Retires the oldest pull and starts the oldest one still waiting for a slot
        (pool.py:90-99)."""
        del self.pending[0]
        if len(self.pending) >= self.pool_size:
            self._start_pull(self.pending[self.pool_size - 1])
        elif not self.pending:
            self.last_pull_slot = -1

    # ------------------------------------------------------------------ pushes
    @torch.no_grad()
    def async_push(self, src: Tensor, offset: Tensor, count: Tensor, dst: Tensor) -> None:
        """Copies the rows of ``src`` to ``dst[offset_i : offset_i + count_i]`` on the next
        slot's push stream (pool.py:101-109)."""
        self.last_push_slot = (self.last_push_slot + 1) % self.pool_size
        self.synchronize_push(self.last_push_slot)
        slot = self._slots[self.last_push_slot]
        src = src if src.is_contiguous() else src.contiguous()
        slot.pushed = src
        stream = slot.stream("push")
        stream.wait_stream(torch.cuda.current_stream(self._device))   # src is complete first
        with torch.cuda.stream(stream):
            _native.write_async(src, offset, count, dst)
        src.record_stream(stream)

    @torch.no_grad()
    def synchronize_push(self, idx: Optional[int] = None) -> None:
        """Waits for the push of slot ``idx``, or for all of them (pool.py:111-123).  A real
        completion wait: the host reads the pinned table afterwards."""
        if idx is None:
            for i in range(self.pool_size):
                self.synchronize_push(i)
            self.last_push_slot = -1
            return
        slot = self._slots[idx]
        if slot.has_stream("push"):
            slot.stream("push").synchronize()
        slot.pushed = None

    def forward(self, *args, **kwargs):
        raise NotImplementedError("AsyncIOPool is driven through async_pull / async_push")

    def extra_repr(self) -> str:
        return (f'pool_size={self.pool_size}, buffer_size={self.buffer_size}, '
                f'embedding_dim={self.embedding_dim}, device={self._device}')
