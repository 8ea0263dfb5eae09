"""Partition-parallel execution over the GPUs of one node: one process per GPU,
``torch.distributed`` (RCCL over xGMI on the GPU box, gloo in the CPU tests).

The reference is single-process (no collective anywhere in its tree); this module adds
the sharding its data layout already implies:

* METIS-style partitions are contiguous id ranges (metis.py:32-33) and batches are unions
  of whole partitions (loader.py:135-138), so rank ``r`` of ``G`` owns partitions
  ``[r P / G, (r+1) P / G)`` = the contiguous History rows ``[row_lo, row_hi)`` of every
  table.  Tables are row-sharded, never replicated; pushes are always local.
* A batch's out-of-batch rows may live on peers.  They are fetched with ONE all-to-all per
  layer (each GPU serves every peer directly, so all xGMI links carry traffic at once;
  nothing funnels through a ring): the owner gathers the requested rows out of its shard
  with the History gather kernel, ``all_to_all_single`` moves them, and they land
  directly in the tail of the halo buffer the aggregation kernel reads -- the batch's
  out-of-batch ids are re-ordered at build time to ``[own rank | peer 0 | peer 1 | ...]``
  precisely so that no scatter is needed on arrival.
* The incremental-aggregation training step touches in-batch rows only
  (loader.py:194-214), so it needs no halo traffic at all; weight gradients are summed
  with one flat all-reduce (a few MB).
"""
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist
from torch import Tensor

from . import ops
from .data import Data
from .loader import SubData, partition_batch
from .sparse import SparseTensor


class PartitionContext:
    """Who owns which partitions / History rows."""

    def __init__(self, ptr: Tensor, group=None, rank: Optional[int] = None,
                 world: Optional[int] = None, exercise_collectives: bool = False):
        self.group = group
        # issue the all-to-all / all-reduce calls even when this is the only rank (they then
        # move nothing): the communication path of a multi-GPU run on a single GPU
        self.exercise_collectives = exercise_collectives
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.ptr = ptr
        n_parts = ptr.numel() - 1
        cuts = [r * n_parts // self.world for r in range(self.world + 1)]
        self.part_bounds = cuts
        self.row_bounds = ptr[torch.tensor(cuts)].clone()          # [world + 1] global row ids
        self.part_lo, self.part_hi = cuts[self.rank], cuts[self.rank + 1]
        self.row_lo, self.row_hi = int(self.row_bounds[self.rank]), int(self.row_bounds[self.rank + 1])

    @property
    def local_rows(self) -> int:
        return self.row_hi - self.row_lo

    @property
    def local_parts(self) -> List[int]:
        return list(range(self.part_lo, self.part_hi))

    def owner_of(self, n_id: Tensor) -> Tensor:
        return torch.bucketize(n_id, self.row_bounds[1:].to(n_id.device), right=True)

    def max_over_ranks(self, value: int) -> int:
        if self.world == 1:
            return value
        t = torch.tensor([value], dtype=torch.int64, device=self._comm_device())
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return int(t)

    def _comm_device(self):
        backend = dist.get_backend(self.group)
        return torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")


class HaloPlan:
    """The exchange pattern of one batch: which of its out-of-batch rows come from which
    rank, and which local rows this rank serves to each peer while the batch is in flight.
    Built collectively (every rank calls :func:`build_halo_plan` the same number of times)."""

    def __init__(self, ctx: PartitionContext, own_local: Tensor, recv_counts: List[int],
                 serve_local: Tensor, send_counts: List[int]):
        self.ctx = ctx
        self.own_local = own_local          # local row ids of out-of-batch rows this rank owns
        self.recv_counts = recv_counts      # rows arriving from each rank (0 for itself)
        self.serve_local = serve_local      # local row ids to send, grouped by destination rank
        self.send_counts = send_counts
        self.n_halo = own_local.numel() + sum(recv_counts)
        # which tail region of the tables receives this batch's foreign rows (History.reserve_tail;
        # fixed before the batch's first in-place exchange: the row ids handed to the
        # aggregation -- and whatever was recorded with them -- point into that region)
        self.region = 0
        self._dev = {}

    def _on(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = (self.own_local.to(device), self.serve_local.to(device))
        return self._dev[key]

    def start_exchange(self, shard: Tensor, width: Optional[int] = None,
                       out: Optional[Tensor] = None) -> "PendingHalo":
        """Starts fetching the batch's out-of-batch rows of one (row-sharded) table: the rows
        this rank owns and the rows its peers asked for are gathered now, on the current
        stream; the all-to-all that moves the latter runs asynchronously (on the communication
        library's own stream) and :meth:`PendingHalo.wait` orders the consumer behind it.
        Collective: every rank of the group calls it once per (round, layer), in the same
        order."""
        device = out.device if out is not None else (
            shard.device if shard.is_cuda or not torch.cuda.is_available()
            else torch.device("cuda", torch.cuda.current_device()))
        own_local, serve_local = self._on(device)
        f = shard.size(1)
        if out is None:
            out = torch.empty((self.n_halo, f), dtype=shard.dtype, device=device)
        n_own = own_local.numel()
        if n_own:
            ops.gather_rows(shard, own_local, out=out[:n_own], device=device)
        work = send = None
        if self.ctx.world > 1 or self.ctx.exercise_collectives:
            send = ops.gather_rows(shard, serve_local, device=device) if serve_local.numel() \
                else torch.empty((0, f), dtype=shard.dtype, device=device)
            recv = out[n_own:self.n_halo]
            work = dist.all_to_all_single(recv, send, self.recv_counts, self.send_counts,
                                          group=self.ctx.group, async_op=True)
        rows = None
        if self.n_halo:
            rows = out[:self.n_halo] if width is None else out[:self.n_halo, :width]
        return PendingHalo(work, rows, send)

    @property
    def n_remote(self) -> int:
        """Out-of-batch rows of the batch that peers own."""
        return sum(self.recv_counts)

    def in_place_ids(self, history) -> Tensor:
        """Row ids of the batch's out-of-batch rows in a table whose tail holds the received
        rows (History.reserve_tail): own rows where they are, row ``tail start + k`` for the
        k-th received one (tail start: the first row of this plan's region) -- what the
        aggregation takes as ``halo_ids``."""
        n_local, device = history.num_embeddings, history.emb.device
        start = n_local + self.region * history.tail_rows
        key = ("ids", str(device), start)
        if key not in self._dev:
            self._dev[key] = torch.cat([
                self.own_local, start + torch.arange(self.n_remote, dtype=torch.int64)]
            ).to(device)
        return self._dev[key]

    def start_exchange_in_place(self, history) -> "PendingHalo":
        """:meth:`start_exchange` without a halo buffer: rows this rank owns are not moved at
        all, rows of peers land in the tail of ``history``'s own allocation, in this plan's
        region (whole rows at the table's pitch, so that both ends of the transfer are
        contiguous).  Collective."""
        base, n_local = history._base, history.num_embeddings
        if base is None or history.tail_rows < self.n_remote or self.region >= history.tail_regions:
            raise RuntimeError(f"the table has {history.tail_regions} region(s) of "
                               f"{history.tail_rows} foreign rows, the batch needs {self.n_remote} "
                               f"in region {self.region} (History.reserve_tail)")
        work = send = None
        if self.ctx.world > 1 or self.ctx.exercise_collectives:
            device = base.device if base.is_cuda or not torch.cuda.is_available() \
                else torch.device("cuda", torch.cuda.current_device())
            _, serve_local = self._on(device)
            send = ops.gather_rows(base[:n_local], serve_local, device=device) \
                if serve_local.numel() else torch.empty((0, base.size(1)), dtype=base.dtype,
                                                        device=device)
            start = n_local + self.region * history.tail_rows
            recv = base[start:start + self.n_remote]
            work = dist.all_to_all_single(recv, send, self.recv_counts, self.send_counts,
                                          group=self.ctx.group, async_op=True)
        return PendingHalo(work, None, send)

    def exchange(self, shard: Tensor, width: Optional[int] = None,
                 out: Optional[Tensor] = None) -> Optional[Tensor]:
        """:meth:`start_exchange` and wait: the rows, in plan order."""
        return self.start_exchange(shard, width, out).wait()


class PendingHalo:
    """Out-of-batch rows on their way (see :meth:`HaloPlan.start_exchange`)."""

    def __init__(self, work, rows: Optional[Tensor], send: Optional[Tensor]):
        self._work, self._rows, self._send = work, rows, send   # `send` stays alive until waited

    def wait(self) -> Optional[Tensor]:
        if self._work is not None:
            self._work.wait()       # RCCL: orders the current stream behind the transfer
            self._work = self._send = None
        return self._rows


def build_halo_plan(ctx: PartitionContext, halo_ids: Tensor) -> Tuple[HaloPlan, Tensor]:
    """``halo_ids``: global ids of a batch's out-of-batch nodes (any order; may be empty).
    Returns the plan and the permutation ``order`` such that ``halo_ids[order]`` is the
    order in which :meth:`HaloPlan.exchange` delivers the rows: own rank first, then the
    peers in rank order.  Collective."""
    owner = ctx.owner_of(halo_ids)
    # sort key: own rank -> -1, others keep their rank
    key = torch.where(owner == ctx.rank, torch.full_like(owner, -1), owner)
    order = torch.argsort(key, stable=True)
    sorted_ids, sorted_owner = halo_ids[order], owner[order]
    own_mask = sorted_owner == ctx.rank
    own_local = sorted_ids[own_mask] - ctx.row_lo
    remote_ids = sorted_ids[~own_mask]
    recv_counts = torch.bincount(sorted_owner[~own_mask], minlength=ctx.world).tolist()
    recv_counts[ctx.rank] = 0
    if ctx.world == 1:
        return HaloPlan(ctx, own_local, [0], torch.zeros(0, dtype=torch.int64), [0]), order
    dev = ctx._comm_device()
    want = torch.tensor(recv_counts, dtype=torch.int64, device=dev)
    serve = torch.empty_like(want)
    dist.all_to_all_single(serve, want, group=ctx.group)
    send_counts = serve.tolist()
    req = remote_ids.to(dev)
    asked = torch.empty(int(sum(send_counts)), dtype=torch.int64, device=dev)
    dist.all_to_all_single(asked, req, send_counts, recv_counts, group=ctx.group)
    serve_local = asked.cpu() - ctx.row_lo
    if serve_local.numel() and (int(serve_local.min()) < 0 or int(serve_local.max()) >= ctx.local_rows):
        raise RuntimeError("a peer requested rows this rank does not own")
    return HaloPlan(ctx, own_local, recv_counts, serve_local, send_counts), order


def sharded_partition_batch(data: Data, ptr: Tensor, part_ids, ctx: PartitionContext,
                            in_batch_only: bool = False) -> SubData:
    """:func:`partition_batch` for partition-parallel runs.  The out-of-batch ids are
    re-ordered to the plan's delivery order (and the adjacency's columns relabelled to
    match); the plan rides on ``adj_t._halo_plan``.  ``part_ids`` may be empty (a rank
    that has run out of batches still has to serve its peers).  Collective unless
    ``in_batch_only``."""
    if len(part_ids) == 0:
        if in_batch_only:
            return None
        plan, _ = build_halo_plan(ctx, torch.zeros(0, dtype=torch.int64))
        ghost = SparseTensor(rowptr=torch.zeros(1, dtype=torch.int64),
                             col=torch.zeros(0, dtype=torch.int64), sparse_sizes=(0, 0))
        ghost._halo_plan = plan
        empty = torch.zeros(0, dtype=torch.int64)
        return SubData(data.__class__(adj_t=ghost), 0, empty, empty, empty)
    sub = partition_batch(data, ptr, part_ids, in_batch_only=in_batch_only)
    if in_batch_only:
        return sub
    b, n_id = sub.batch_size, sub.n_id
    plan, order = build_halo_plan(ctx, n_id[b:])
    inv = torch.empty_like(order)
    inv[order] = torch.arange(order.numel())
    rowptr, col, value = sub.data.adj_t.csr()
    new_col = torch.where(col < b, col, b + inv[(col - b).clamp(min=0)])
    adj_t = SparseTensor(rowptr=rowptr, col=new_col, value=value,
                         sparse_sizes=sub.data.adj_t.sparse_sizes(), is_sorted=True)
    adj_t._halo_plan = plan
    new_n_id = torch.cat([n_id[:b], n_id[b:][order]])
    out = data.__class__(adj_t=adj_t)
    for k, v in sub.data:
        if k != "adj_t" and isinstance(v, Tensor) and v.dim() > 0 and v.size(0) == n_id.numel():
            out[k] = torch.cat([v[:b], v[b:][order]])
    return SubData(out, b, new_n_id, sub.offset, sub.count)


def sharded_eval_batches(data: Data, ptr: Tensor, ctx: PartitionContext,
                         batch_size: int = 1) -> List[SubData]:
    """This rank's share of the evaluation batches (groups of ``batch_size`` consecutive own
    partitions), padded with empty batches so that every rank runs the same number of
    exchange rounds.  Collective."""
    parts = ctx.local_parts
    groups = [parts[i:i + batch_size] for i in range(0, len(parts), batch_size)]
    rounds = ctx.max_over_ranks(len(groups))
    groups += [[] for _ in range(rounds - len(groups))]
    return [sharded_partition_batch(data, ptr, g, ctx) for g in groups]


def _plan_of(sub) -> Optional[HaloPlan]:
    """The halo plan of a SubData, an adjacency or a plan itself."""
    if isinstance(sub, HaloPlan):
        return sub
    return getattr(getattr(getattr(sub, "data", None), "adj_t", sub), "_halo_plan", None)


def reserve_halo_tails(model, batches, regions: int = 1) -> int:
    """Gives every layer-input table of ``model`` room (``regions`` times over) for the largest
    number of foreign out-of-batch rows any of ``batches`` (SubData from
    :func:`sharded_partition_batch`) needs; the push/pull step and the sweeps then read
    out-of-batch rows in place.  Returns that number.  Collective."""
    need, ctx = 0, None
    for sub in batches:     # SubData, adjacencies or the plans themselves
        plan = _plan_of(sub)
        if plan is not None:
            need, ctx = max(need, plan.n_remote), plan.ctx
    if ctx is not None:     # every rank reserves (and so takes the in-place form of the exchange)
        need = ctx.max_over_ranks(need)
    for h in model.histories:
        h.reserve_tail(need, regions)
    return need


class HaloPipeline:
    """Push/pull training steps over a fixed cyclic sequence of batches with the exchange of
    out-of-batch rows running ONE STEP AHEAD: while step ``i`` computes, the rows that step
    ``i + 1`` will read travel over the links into another region of the table tails.

    What a step reads is then well defined and the same on every run: the out-of-batch rows
    of step ``i + 1`` are gathered on their owners at the top of step ``i``, i.e. they hold the
    tables as the pushes of steps ``< i`` left them -- one step older than with an exchange at
    the top of step ``i + 1`` itself.  (Ranks run their batches side by side, so "the value a
    peer holds right now" is no better defined in either form; historical rows are stale by
    construction, history.py:9-74.)

    Regions: consecutive batches of the cycle must not share one.  Batch ``k`` gets region
    ``k % 2``; with an odd number of batches the last one gets a third region (it is followed
    by batch 0).  Every rank must bring the same number of batches: the exchange plan of a
    rank's k-th batch was built together with its peers' k-th batches
    (:func:`build_halo_plan`), and that is the round they have to be exchanged in.  With a
    single batch per rank there is nothing to look ahead to, and the pipeline degenerates to an
    exchange at the top of every step.

    Use: ``pipe = HaloPipeline(model, adjs)`` once (collective; assigns regions and reserves
    the tails), then per step ``pipe.ready(k)`` followed by ``model(..., halo_ready=True)`` for
    batch ``k``; ``pipe.drain()`` before anything that must not overlap a collective (recording
    a HIP graph)."""

    def __init__(self, model, adjs, ahead: bool = True):
        self.model, self.adjs = model, list(adjs)
        plans = [_plan_of(a) for a in self.adjs]
        if any(p is None for p in plans):
            raise RuntimeError("every batch needs a halo plan (sharded_partition_batch)")
        ctx = plans[0].ctx
        n = len(plans)
        if ctx.max_over_ranks(n) != n or -ctx.max_over_ranks(-n) != n:
            raise RuntimeError("every rank must bring the same number of batches")
        self.ahead = bool(ahead) and n >= 2
        regions = 1
        if self.ahead:
            for k, p in enumerate(plans):
                p.region = 2 if (n % 2 == 1 and k == n - 1) else k % 2
            regions = 3 if n % 2 else 2
        reserve_halo_tails(model, plans, regions)
        self._pending = {}      # batch index -> its exchanges in flight, one per layer

    def ready(self, k: int) -> None:
        """Orders the current stream behind the arrival of batch ``k``'s out-of-batch rows and
        starts the exchange for the batch that follows it in the cycle.  Collective."""
        cur = self._pending.pop(k, None)
        if self._pending:       # the look-ahead was for another batch: those rows are dropped
            self.drain()        # (their region may be written again before they are read)
        if cur is None:
            cur = self.model.start_halo_exchanges(self.adjs[k])
        for pending in cur:
            pending.wait()
        if self.ahead:
            nxt = (k + 1) % len(self.adjs)
            if nxt not in self._pending:
                self._pending[nxt] = self.model.start_halo_exchanges(self.adjs[nxt])

    def drain(self) -> None:
        """Waits for every exchange in flight (their rows are dropped: the next
        :meth:`ready` fetches afresh)."""
        for k in sorted(self._pending):
            for pending in self._pending[k]:
                pending.wait()
        self._pending = {}


def allreduce_gradients(module: torch.nn.Module, ctx: PartitionContext) -> None:
    """Mean of the weight gradients over the ranks, as ONE flat all-reduce (the models here
    have a few MB of parameters: a single bucket is the right granularity)."""
    if ctx.world == 1 and not ctx.exercise_collectives:
        return
    grads = [p.grad for p in module.parameters() if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, group=ctx.group)
    flat.div_(ctx.world)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


def enable_collective_records(entries: int = 2000) -> None:
    """Turns on torch.distributed's record of issued collectives (its "flight recorder"), which
    :func:`quiesce_collectives` reads.  Call BEFORE ``init_process_group``; an explicit setting in
    the environment is left alone."""
    import os
    os.environ.setdefault("TORCH_FR_BUFFER_SIZE", str(int(entries)))


def quiesce_collectives(timeout: float = 10.0) -> str:
    """Returns once every collective issued so far has completed AND the process group's
    watchdog thread has retired its bookkeeping for it (the thread polls the collectives'
    events a few times per second, and an event poll while a stream records a HIP graph is an
    error on some builds): call it between the last eager collective and a graph recording.

    The condition is read from torch.distributed's record of collectives (every entry marked
    ``retired``; :func:`enable_collective_records`) -- returns ``"retired"``.  Where that record
    is not kept, the call falls back to waiting 1.5 s, fifteen polling periods of the watchdog,
    and returns ``"slept"``."""
    import pickle
    import time
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    try:
        from torch._C._distributed_c10d import _dump_nccl_trace
    except ImportError:
        _dump_nccl_trace = None
    if _dump_nccl_trace is not None and dist.is_initialized():
        deadline = time.monotonic() + timeout
        while True:
            try:
                # (a dict of plain values written by this process's own torch)
                entries = pickle.loads(_dump_nccl_trace()).get("entries", [])
            except Exception:
                break
            if not entries:
                break                       # no record is kept (or nothing was ever issued)
            if all(e.get("retired", False) for e in entries):
                return "retired"
            if time.monotonic() > deadline:
                break               # (a record that never settles: wait it out below instead)
            time.sleep(0.005)
    time.sleep(1.5)
    return "slept"


def flatten_gradients(module: torch.nn.Module) -> Tensor:
    """Makes every parameter's ``.grad`` a view into ONE flat buffer (zero-filled) and returns
    the buffer: the gradient all-reduce is then a single collective on it, with no
    concatenation before and no copies after.  Use ``zero_grad(set_to_none=False)`` from then
    on, so that backward passes keep accumulating into the views."""
    params = [p for p in module.parameters() if p.requires_grad]
    if not params:
        return torch.zeros(0)
    flat = torch.zeros(sum(p.numel() for p in params), dtype=params[0].dtype,
                       device=params[0].device)
    off = 0
    for p in params:
        p.grad = flat[off:off + p.numel()].view_as(p)
        off += p.numel()
    return flat


def allreduce_flat_gradients(flat: Tensor, ctx: PartitionContext) -> None:
    """Mean over the ranks of a buffer from :func:`flatten_gradients`."""
    if ctx.world == 1 and not ctx.exercise_collectives:
        return
    dist.all_reduce(flat, group=ctx.group)
    if ctx.world > 1:
        flat.div_(ctx.world)


def broadcast_parameters(module: torch.nn.Module, ctx: PartitionContext, src: int = 0) -> None:
    """Replicates rank ``src``'s weights (and batch-norm statistics) on every rank."""
    if ctx.world == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src, group=ctx.group)
