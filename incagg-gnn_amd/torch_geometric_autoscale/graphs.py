"""HIP-graph capture of whole training steps.

A history-based training step is ~110 kernel launches (aggregations, History pushes, GEMMs,
element-wise glue, the optimizer) issued from Python; on MI355X the launches of one step take
the host longer than the GPU needs to run them.  A step over a given partition batch is the
same launch sequence every epoch -- same shapes, same addresses once its inputs are kept in
static tensors -- so it is recorded once into a HIP graph and replayed with a single call.

All graphs of a :class:`StepGraphs` share one memory pool and one capture stream (they are
replayed one after the other, never concurrently).  The aggregation and History-row entry
points allocate nothing and keep no state of their own between launches, so they record as plain
kernel nodes; :meth:`StepGraphs.warm` runs the step eagerly on the capture stream first, which
builds every lazily built per-batch structure (transposed CSR, work plans, in-place column ids)
outside the recording.  Tensors a recorded step reads -- batch arrays, plans and their combine
buffers, History tables, parameters -- must stay alive and keep their addresses.

Known limitation (observed on ROCm 7.0 / torch 2.10 with an earlier build of the library; cause
not established, and not re-tested since): replaying recorded steps AFTER other work had been
issued eagerly on the device since the recording -- a refresh sweep between two epochs of replays
-- ended in an illegal memory access, while the same graphs replay indefinitely when nothing
eager runs in between (warm everything, record everything, then only replay: what ``bench.py``
does), and recording / replaying interleaved with eager *training steps* is fine too.  Loops
that alternate training epochs with eager sweeps (``train.run``) therefore issue their steps
eagerly.
"""
from typing import Callable, Dict, Hashable

import torch


class StepGraphs:
    def __init__(self, device):
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(self.device)
        self.pool = torch.cuda.graph_pool_handle()
        self.graphs: Dict[Hashable, torch.cuda.CUDAGraph] = {}

    def warm(self, fn: Callable[[], None], times: int = 2) -> None:
        """Runs ``fn`` eagerly on the capture stream (lazy initialisation happens here)."""
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.stream):
            for _ in range(times):
                fn()
        torch.cuda.current_stream(self.device).wait_stream(self.stream)

    def capture(self, key: Hashable, fn: Callable[[], None],
                capture_error_mode: str = "global") -> None:
        """Records one execution of ``fn`` (all tensors it reads must stay alive and keep
        their addresses; whatever it allocates lives in the shared pool).
        ``capture_error_mode="thread_local"``: other threads may keep calling the HIP runtime
        while this one records -- needed next to a ``torch.distributed`` process group, whose
        watchdog thread polls the events of earlier collectives."""
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, pool=self.pool, stream=self.stream,
                              capture_error_mode=capture_error_mode):
            try:
                fn()
            except BaseException:
                # ending a broken recording can itself fail (or crash): show the cause first
                import sys
                import traceback
                traceback.print_exc(file=sys.stderr)
                sys.stderr.flush()
                raise
        self.graphs[key] = graph

    def replay(self, key: Hashable) -> None:
        self.graphs[key].replay()

    def __contains__(self, key: Hashable) -> bool:
        return key in self.graphs
