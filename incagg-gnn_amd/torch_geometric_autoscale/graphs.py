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

Envelope the recordings are used in (and tested in, tests/test_gpu_train_graphs.py): once
steps have been recorded, the device only sees replays (steps and the recorded refresh sweep,
:class:`SweepGraph`), eagerly issued *training steps*, and small host<->device transfers through
pinned memory.  An earlier build of the library showed an illegal memory access when recorded
steps were replayed after an EAGERLY issued refresh sweep; the cause was never established (the
library kept mutable device state across launches then, which it no longer does), so that
sequence is refused instead of run: :meth:`StepGraphs.replay` raises when an eager sweep has been
issued since the step was recorded.  Loops that alternate training epochs with refresh sweeps
(``train.run``) record the sweep too, which also makes the sweep host-free.
"""
import logging
from typing import Callable, Dict, Hashable, Optional

import torch

log = logging.getLogger("incagg.graphs")

_eager_sweeps = 0


def note_eager_sweep() -> None:
    """Called by the models whenever a refresh sweep is issued eagerly (not being recorded)."""
    global _eager_sweeps
    _eager_sweeps += 1


class StepGraphs:
    def __init__(self, device):
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(self.device)
        self.pool = torch.cuda.graph_pool_handle()
        self.graphs: Dict[Hashable, torch.cuda.CUDAGraph] = {}
        self._sweeps_at_capture: Dict[Hashable, int] = {}

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
        self._sweeps_at_capture[key] = _eager_sweeps

    def replay(self, key: Hashable) -> None:
        if self._sweeps_at_capture[key] != _eager_sweeps:
            raise RuntimeError(
                "a refresh sweep has been issued eagerly since this step was recorded; replaying "
                "recorded steps after an eager sweep is not supported (record the sweep with "
                "graphs.SweepGraph, or record the steps again)")
        self.graphs[key].replay()

    def __contains__(self, key: Hashable) -> bool:
        return key in self.graphs


class SweepGraph:
    """A whole refresh sweep -- ``mini_inference`` / ``mini_inference_vr`` over a FIXED list of
    evaluation batches (models/gcn2.py:433-507) -- recorded once as one HIP graph and replayed
    with a single call per epoch.  A sweep is ``batches x layers`` rounds of aggregation, dense
    update and one or two History pushes, 6-10 launches each, issued from Python: on MI355X the
    host needs longer to issue a round than the GPU to run it.  Evaluation batches are static
    (loader.py:266-284), the tables and parameters keep their addresses, so the launch sequence
    is the same every epoch.

    The first call runs the sweep eagerly on the capture stream (that IS the first sweep: tables
    and output are filled; every lazily built per-batch structure and the device copies of the
    batches' features come into being) and then records it; later calls replay.  Only for
    models whose tables live where they compute, without staging pools and outside a partition
    context -- anything else, and any failure to record, leaves the sweep to be issued eagerly
    (``recorded`` tells which).  The model is switched to evaluation mode for the sweep and
    back afterwards."""

    def __init__(self, model, loader, with_aggregates: bool = True, use_aggregation: bool = True,
                 graphs: Optional[StepGraphs] = None):
        self.model = model
        self.batches = list(loader)
        self.with_aggregates, self.use_aggregation = with_aggregates, use_aggregation
        self.graphs = graphs
        self.key = ("sweep", id(self))
        self.recorded = False
        self.calls = 0
        dev = torch.device(model.device)
        self.supported = (
            dev.type == "cuda" and model.pool is None and model.partition_ctx is None
            and model.emb_device.type == "cuda"
            and (model.emb_device.index or 0) == (dev.index or torch.cuda.current_device()))

    def _sweep(self):
        return self.model._sweep(self.batches, self.use_aggregation, self.with_aggregates)

    def _resident(self) -> bool:
        """Whether every batch's features are in HBM (a recording cannot copy from the host)."""
        dev = torch.device(self.model.device)
        for sub in self.batches:
            data = sub[0]
            x = data.x
            if x.is_cuda:
                continue
            kept = data.__dict__.get("_x_on_device", {})
            if not any(d.type == "cuda" and (d.index or 0) == (dev.index or 0) for d in kept):
                return False
        return True

    def __call__(self):
        model = self.model
        was_training = model.training
        model.eval()
        try:
            self.calls += 1
            if self.recorded:
                self.graphs.replay(self.key)
                return model._out
            if not self.supported:
                return self._sweep()
            if self.graphs is None:
                self.graphs = StepGraphs(model.device)
            out = None

            def eager():
                nonlocal out
                out = self._sweep()
            self.graphs.warm(eager, times=1)
            if not self._resident():
                self.supported = False
                return out
            try:
                self.graphs.capture(self.key, self._sweep)
                self.recorded = True
            except Exception as exc:    # the sweep stays eager
                self.supported = False
                log.warning("recording the refresh sweep failed (%r); it is issued eagerly", exc)
            return out
        finally:
            model.train(was_training)
