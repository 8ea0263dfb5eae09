"""Ring of staging slots between pinned-host History tables and HBM (interface and queue
discipline of the reference's ``AsyncIOPool``, pool.py:15-123).

Each of the ``pool_size`` slots owns a pull stream, a push stream, a pinned host
buffer and a device buffer of ``[buffer_size, embedding_dim]``.  Pull requests enter
a FIFO; at most ``pool_size`` are in flight, the rest start as slots are freed.

Differences in mechanism (not in results):

* no host-blocking stream synchronisation -- consumers are ordered behind the copies
  with stream waits, and a slot's device buffer is only refilled after the compute
  stream has passed the point where it was freed;
* with ``zero_copy=True`` (default) the gathered part of a pull is a HIP kernel on
  the pull stream that reads the pinned table directly over the host link, instead
  of a single-threaded host gather into the pinned bounce buffer followed by a copy
  (csrc/cuda/async_cuda.cu:95-110).  Contiguous chunks always go through the DMA
  engines (hipMemcpyAsync).
"""
from typing import Callable, Optional

import torch
from torch import Tensor

from . import ops


class AsyncIOPool(torch.nn.Module):
    def __init__(self, pool_size: int, buffer_size: int, embedding_dim: int,
                 zero_copy: bool = True):
        super().__init__()

        self.pool_size = pool_size
        self.buffer_size = buffer_size
        self.embedding_dim = embedding_dim
        self.zero_copy = zero_copy

        self._device = torch.device('cpu')
        self._pull_queue = []
        self._push_cache = [None] * pool_size
        self._push_streams = [None] * pool_size
        self._pull_streams = [None] * pool_size
        self._cpu_buffers = [None] * pool_size
        self._cuda_buffers = [None] * pool_size
        self._pull_index = -1
        self._push_index = -1

    def _apply(self, fn: Callable, *args, **kwargs) -> None:
        self._device = fn(torch.zeros(1)).device
        return self

    # ---- lazily created per-slot resources (pool.py:37-62)
    def _pull_stream(self, idx: int):
        if self._pull_streams[idx] is None:
            assert str(self._device)[:4] == 'cuda'
            self._pull_streams[idx] = torch.cuda.Stream(self._device)
        return self._pull_streams[idx]

    def _push_stream(self, idx: int):
        if self._push_streams[idx] is None:
            assert str(self._device)[:4] == 'cuda'
            self._push_streams[idx] = torch.cuda.Stream(self._device)
        return self._push_streams[idx]

    def _cpu_buffer(self, idx: int) -> Tensor:
        if self._cpu_buffers[idx] is None:
            self._cpu_buffers[idx] = torch.empty(self.buffer_size, self.embedding_dim,
                                                 pin_memory=True)
        return self._cpu_buffers[idx]

    def _cuda_buffer(self, idx: int) -> Tensor:
        if self._cuda_buffers[idx] is None:
            assert str(self._device)[:4] == 'cuda'
            self._cuda_buffers[idx] = torch.empty(self.buffer_size, self.embedding_dim,
                                                  device=self._device)
        return self._cuda_buffers[idx]

    # ---- pull side
    @torch.no_grad()
    def async_pull(self, src: Tensor, offset: Optional[Tensor], count: Optional[Tensor],
                   index: Tensor) -> None:
        # Start pulling `src` at ([offset, count] and index positions:
        self._pull_index = (self._pull_index + 1) % self.pool_size
        data = (self._pull_index, src, offset, count, index)
        self._pull_queue.append(data)
        if len(self._pull_queue) <= self.pool_size:
            self._async_pull(self._pull_index, src, offset, count, index)

    @torch.no_grad()
    def _async_pull(self, idx: int, src: Tensor, offset: Optional[Tensor],
                    count: Optional[Tensor], index: Tensor) -> None:
        stream = self._pull_stream(idx)
        dst = self._cuda_buffer(idx)
        # The slot's device buffer may still be read by kernels already queued on the
        # compute stream: refill it only behind them.
        stream.wait_stream(torch.cuda.current_stream(self._device))
        with torch.cuda.stream(stream):
            if self.zero_copy and index.numel() > 0 and src.is_pinned() \
                    and src.dtype == torch.float32:
                numel = 0 if count is None else int(count.sum())
                if numel + index.numel() > self.buffer_size:
                    raise RuntimeError("Buffer tensor size too small")
                no_index = torch.empty(0, dtype=torch.int64)     # (the staging op wants host ids)
                ops.read_async(src, offset, count, no_index, dst, self._cpu_buffer(idx))
                idx_dev = index.to(self._device, non_blocking=True)
                ops.gather_rows(src, idx_dev, out=dst[numel:numel + index.numel()],
                                device=self._device)
            else:
                ops.read_async(src, offset, count, index.cpu(), dst, self._cpu_buffer(idx))

    @torch.no_grad()
    def synchronize_pull(self) -> Tensor:
        idx = self._pull_queue[0][0]
        ops.synchronize()  # every copy of the oldest pull has been enqueued on its stream
        torch.cuda.current_stream(self._device).wait_stream(self._pull_stream(idx))
        return self._cuda_buffer(idx)

    @torch.no_grad()
    def free_pull(self) -> None:
        # Free the buffer space and start pulling from remaining queue:
        self._pull_queue.pop(0)
        if len(self._pull_queue) >= self.pool_size:
            data = self._pull_queue[self.pool_size - 1]
            idx, src, offset, count, index = data
            self._async_pull(idx, src, offset, count, index)
        elif len(self._pull_queue) == 0:
            self._pull_index = -1

    # ---- push side
    @torch.no_grad()
    def async_push(self, src: Tensor, offset: Tensor, count: Tensor, dst: Tensor) -> None:
        # Start pushing `src` to ([offset, count] and index positions to `dst`:
        self._push_index = (self._push_index + 1) % self.pool_size
        self.synchronize_push(self._push_index)
        if not src.is_contiguous():
            src = src.contiguous()
        self._push_cache[self._push_index] = src
        stream = self._push_stream(self._push_index)
        # `src` is produced on the compute stream: copy it out only once it is complete.
        stream.wait_stream(torch.cuda.current_stream(self._device))
        with torch.cuda.stream(stream):
            ops.write_async(src, offset, count, dst)
        src.record_stream(stream)

    @torch.no_grad()
    def synchronize_push(self, idx: Optional[int] = None) -> None:
        # Synchronize the push command of stream `idx` or all commands:
        if idx is None:
            for idx in range(self.pool_size):
                self.synchronize_push(idx)
            self._push_index = -1
        else:
            # The pinned table is read by the HOST afterwards (the next pull gathers
            # from it), so this one has to be a real completion wait.
            if self._push_streams[idx] is not None:
                self._push_streams[idx].synchronize()
            self._push_cache[idx] = None

    def forward(self, *args, **kwargs):
        """"""
        raise NotImplementedError

    def __repr__(self):
        return (f'{self.__class__.__name__}(pool_size={self.pool_size}, '
                f'buffer_size={self.buffer_size}, '
                f'embedding_dim={self.embedding_dim}, '
                f'device={self._device})')
