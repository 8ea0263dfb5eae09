"""``ScalableGNN``: history-based mini-batch execution of a GNN (interface of the
reference's models/base.py:26-709) over the HIP aggregation and History-row kernels.

One indexing convention is used throughout (the reference mixes two, see
models/base.py:544-579 against models/gcn2.py:129-138):

* ``histories[i]``    holds the INPUT rows of layer ``i`` (``i = 0``: the embedded
  features, zero-padded to the table width) -- "M_in";
* ``histories_ag[i]`` holds the full one-hop aggregate of ``histories[i]`` -- "M_ag";

exactly what the refresh sweep writes (models/gcn.py:367-369,387,390).  The classic
push/pull forward therefore pushes the output of layer ``i`` to ``histories[i + 1]``
and pulls the out-of-batch rows of ``histories[i + 1]``; it issues as many pulls as it
consumes.

Subclasses provide three small hooks -- :meth:`_embed`, :meth:`_update`,
:attr:`agg_reduce` -- from which this class builds ``forward`` (push/pull),
``VR_forward`` (incremental aggregation), ``forward_layer`` and both refresh sweeps.
There are no device-wide synchronisations anywhere in these paths.
"""
import warnings
from typing import Any, Callable, Dict, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

from .. import ops
from ..aggregate import incagg_aggregate, spmm
from ..history import History
from ..nn import PushTarget, relu_into, unweighted
from ..pool import AsyncIOPool
from ..sparse import SparseTensor


class ScalableGNN(torch.nn.Module):
    r"""An abstract class for implementing scalable GNNs via historical embeddings.

    Args:
        num_nodes (int): The number of nodes in the graph.
        hidden_channels (int): Width of every History table.
        num_layers (int): The number of layers of the model.
        pool_size (int, optional): Number of staging slots for host-resident tables.
        buffer_size (int, optional): Rows per staging slot.
        device: Where the History tables live (``None`` / ``'cpu'`` = pinned host
            memory, ``'cuda:i'`` = HBM).
    """

    def __init__(self, num_nodes: int, hidden_channels: int, num_layers: int,
                 pool_size: Optional[int] = None, buffer_size: Optional[int] = None,
                 device=None, in_channels=None):
        super().__init__()
        self.num_nodes = num_nodes
        self.hidden_channels = hidden_channels
        self.num_layers = num_layers
        self.pool_size = num_layers - 1 if pool_size is None else pool_size
        self.buffer_size = buffer_size

        # Layer 0 of some models aggregates the raw features, which may be wider than the hidden
        # width (the reference only handles the narrower case, models/gcn.py:355-365).  Device
        # tables of layer 0 are then made wide enough; host tables keep one width for all
        # layers because the staging slots are shared between them.
        w0 = hidden_channels
        if self.raw_input_layer and in_channels is not None and device is not None:
            w0 = max(hidden_channels, int(in_channels))
        widths = [w0] + [hidden_channels] * (num_layers - 1)
        self.histories = torch.nn.ModuleList(
            [History(num_nodes, w, device) for w in widths])
        self.pool: Optional[AsyncIOPool] = None
        self.histories_ag = torch.nn.ModuleList(
            [History(num_nodes, w, device) for w in widths])
        self.pool_ag: Optional[AsyncIOPool] = None

        self._async = False
        self.__out: Optional[Tensor] = None
        # A (x - M_in) + M_ag in one launch that gathers both operands per edge (True), or a
        # streaming subtract first and M_ag in the aggregation's epilogue (False).  Measured on
        # MI355X at width 256 (products batch): the gathers are the bottleneck, the two-pass form
        # halves them and wins, 3.31 vs 3.49 ms per training step.
        self.fused_incagg = False
        # partition-parallel runs: a distributed.PartitionContext; the tables then hold only
        # this rank's rows (construct the model with num_nodes = ctx.local_rows)
        self.partition_ctx = None
        self._current_plan = None
        self._halo_pending = []   # partition-parallel: exchanges started at the top of a step
        # push/pull forward and refresh sweeps: let the aggregation read out-of-batch rows
        # straight from the HBM-resident table (no gathered copy) whenever nothing has to
        # touch those rows first (no dropout on them, no per-row transform before aggregating)
        self.halo_in_place = True

    # ------------------------------------------------------------------ placement
    @property
    def emb_device(self):
        return self.histories[0].emb.device

    @property
    def device(self):
        return self.histories[0]._device

    def _apply(self, fn: Callable, *args, **kwargs) -> None:
        super()._apply(fn, *args, **kwargs)
        # Staging pools exist only when the tables are on the host and the model on a GPU.
        if (str(self.emb_device) == 'cpu' and str(self.device)[:4] == 'cuda'
                and self.pool_size is not None and self.buffer_size is not None):
            self.pool = AsyncIOPool(self.pool_size, self.buffer_size,
                                    self.histories[0].embedding_dim)
            self.pool.to(self.device)
            self.pool_ag = AsyncIOPool(self.pool_size, self.buffer_size,
                                       self.histories_ag[0].embedding_dim)
            self.pool_ag.to(self.device)
        return self

    def reset_parameters(self):
        for history in self.histories:
            history.reset_parameters()
        for history in self.histories_ag:
            history.reset_parameters()

    def set_partition_context(self, ctx) -> "ScalableGNN":
        """Switches to partition-parallel mode (see ``distributed.py``): History row ids
        become ``global id - ctx.row_lo`` and out-of-batch rows owned by peers arrive
        through the batch's halo plan."""
        if ctx is not None and ctx.local_rows != self.num_nodes:
            raise ValueError(f"the tables hold {self.num_nodes} rows but this rank owns "
                             f"{ctx.local_rows}")
        self.partition_ctx = ctx
        return self

    def start_halo_exchanges(self, adj_t: SparseTensor) -> list:
        """Partition-parallel, tables with tails: starts the in-place exchanges of every layer
        for the batch ``adj_t`` belongs to and returns them (``PendingHalo``; ``wait()`` orders
        the current stream behind the transfer).  Collective.  Together with
        ``model(..., halo_ready=True)`` this splits a step into its collectives and a
        collective-free part."""
        plan = getattr(adj_t, "_halo_plan", None)
        if self.partition_ctx is None or plan is None:
            raise RuntimeError("needs a partition context and a batch from sharded_partition_batch")
        return [plan.start_exchange_in_place(h) for h in self.histories[1:]]

    def _local(self, offset: Optional[Tensor]) -> Optional[Tensor]:
        ctx = self.partition_ctx
        return offset if ctx is None or offset is None else offset - ctx.row_lo

    # ------------------------------------------------------------------ hooks
    agg_reduce = "sum"
    raw_input_layer = False   # layer 0 aggregates the raw features (no embedding in front)

    def _embed(self, x: Tensor, state: Dict[str, Any], batch_size: int) -> Tensor:
        """Layer-0 input rows from raw features (may fill ``state``)."""
        raise NotImplementedError

    def _update(self, layer: int, agg: Tensor, x: Tensor, state: Dict[str, Any],
                pre: bool = False) -> Tensor:
        """Everything of layer ``layer`` after the aggregation of its input ``x``
        (``pre``: the aggregate is of :meth:`_pre_aggregate`'s output instead)."""
        raise NotImplementedError

    def _pre_aggregate(self, layer: int, x: Tensor) -> Optional[Tensor]:
        """Optionally a narrower, linearly transformed ``x`` to aggregate instead (only the
        push/pull forward uses it; the history tables always hold un-transformed rows)."""
        return None

    def _pre_aggregates(self, layer: int) -> bool:
        """Whether :meth:`_pre_aggregate` returns a matrix for ``layer`` (its out-of-batch rows
        must then be pulled and transformed too: they cannot be read in place from the table)."""
        return type(self)._pre_aggregate is not ScalableGNN._pre_aggregate

    def _mix(self, layer: int, state: Dict[str, Any], batch_size: int):
        """Optional ``(scale, add, add_scale)`` to fold into the aggregation epilogue, i.e.
        the layer consumes ``scale * agg + add_scale * add`` (models/appnp.py:85);
        :meth:`_update_mixed` then receives that instead of the raw aggregate."""
        return None

    def _update_mixed(self, layer: int, mixed: Tensor, x: Tensor, state: Dict[str, Any]) -> Tensor:
        raise NotImplementedError

    _layer_dropout = True  # dropout on the input of every layer after the first

    def _adj(self, adj_t: SparseTensor) -> SparseTensor:
        return unweighted(adj_t) if self.agg_reduce == "mean" else adj_t

    def _aggregate(self, adj_t: SparseTensor, x: Tensor, halo: Optional[Tensor] = None,
                   halo_ids: Optional[Tensor] = None, out: Optional[Tensor] = None) -> Tensor:
        return spmm(self._adj(adj_t), x, reduce=self.agg_reduce, x_halo=halo, halo_ids=halo_ids,
                    out=out)

    def _drop(self, x: Tensor) -> Tensor:
        p = getattr(self, "dropout", 0.0)
        return F.dropout(x, p=p, training=self.training) if p > 0 else x

    def _layer_input(self, layer: int, x: Tensor) -> Tensor:
        return self._drop(x) if layer > 0 and self._layer_dropout else x

    # ------------------------------------------------------------------ calls
    def __call__(self, x: Optional[Tensor] = None, adj_t: Optional[SparseTensor] = None,
                 batch_size: Optional[int] = None, n_id: Optional[Tensor] = None,
                 offset: Optional[Tensor] = None, count: Optional[Tensor] = None,
                 loader=None, drift_norm: int = 2, aggregate_combined: bool = True,
                 use_aggregation: bool = True, halo_ready: bool = False, **kwargs):
        r"""Push/pull forward (models/base.py:126-240): starts the pulls of the
        out-of-batch rows of every layer, runs :meth:`forward`, waits for the pushes.
        Returns a dict whose ``'out'`` entry is the model output.

        ``halo_ready`` (partition-parallel, tables with tails): the caller has run
        :meth:`start_halo_exchanges` for this batch and waited for them; the call then
        contains no collective and can be recorded into a HIP graph."""
        if loader is not None:
            return self.mini_inference(loader, use_aggregation)
        self._halo_ready = bool(halo_ready)

        self._async = (self.pool is not None and batch_size is not None and n_id is not None
                       and offset is not None and count is not None)
        if (batch_size is not None and not self._async and str(self.emb_device) == 'cpu'
                and str(self.device)[:4] == 'cuda' and self.partition_ctx is None):
            warnings.warn('Asynchronous I/O disabled, although history and '
                          'model sit on different devices.')

        self._current_plan = getattr(adj_t, "_halo_plan", None)
        if self.partition_ctx is not None:
            # Partition-parallel: a rank's table shard may also live in its pinned host memory.
            # Rows then move by the row kernels reading / writing the pinned shard directly
            # over the host link (own pushes, own out-of-batch rows, rows served to peers in
            # front of the all-to-all); the staging ring is not used.
            self._async = False
            if use_aggregation and self._current_plan is None and n_id is not None \
                    and n_id.numel() > (batch_size or 0):
                raise RuntimeError("partition-parallel batches must come from "
                                   "distributed.sharded_partition_batch (no halo plan attached)")
        if self._async and use_aggregation:
            for hist in self.histories[1:]:
                self.pool.async_pull(hist.emb, None, None, n_id[batch_size:])
        if self._halo_ready:
            if self.partition_ctx is None or self._current_plan is None or not all(
                    self._in_place_ok(h, h.emb[:0].to(self.device)) for h in self.histories[1:]):
                raise RuntimeError("halo_ready needs a partition context, a batch with a halo plan "
                                   "and tables that receive foreign rows in place (reserve_tail)")
        elif (self.partition_ctx is not None and self._current_plan is not None and use_aggregation
                and batch_size is not None):
            # Like the pool's pulls above (models/base.py:186-188), the exchanges of ALL layers
            # start now: out-of-batch rows are historical values either way, and the transfers
            # over xGMI then overlap with the layers' compute instead of preceding each of them.
            # (Rows a peer asks for are read before this step's own pushes.)
            # (tables with room for the foreign rows receive them in place, see _in_place_ok;
            # the width probe stands for the layer input, which is as wide as its table)
            self._halo_pending = [
                self._current_plan.start_exchange_in_place(h)
                if self._in_place_ok(h, h.emb[:0].to(self.device))
                else self._current_plan.start_exchange(h.emb)
                for h in self.histories[1:]]

        try:
            out, t_move = self.forward(x, adj_t, drift_norm, aggregate_combined, use_aggregation,
                                       batch_size, n_id, offset, count, **kwargs)
        finally:
            for pending in self._halo_pending:     # keep the collectives matched on every path
                pending.wait()
            self._halo_pending = []

        if self._async:
            self.pool.synchronize_push()
        self._async = False
        return {'out': out, 'time_pool_pull': 0, 'time_pool_push': 0,
                'time_forward_histories_movement': t_move, 'time_forward_total': 0}

    def VR_call(self, x: Optional[Tensor] = None, adj_t: Optional[SparseTensor] = None,
                batch_size: Optional[int] = None, n_id: Optional[Tensor] = None,
                offset: Optional[Tensor] = None, count: Optional[Tensor] = None,
                loader=None, debug_flag: bool = False, drift_norm: int = 2, epoch: int = 0,
                batch_idx: int = 0, **kwargs):
        r"""Incremental-aggregation forward (models/base.py:242-378): pulls the batch's
        own rows of ``M_in`` / ``M_ag`` for every layer and runs :meth:`VR_forward`.
        Nothing is pushed; the tables change only in :meth:`mini_inference_vr`."""
        if loader is not None:
            return self.mini_inference(loader)

        self._async = (self.pool is not None and batch_size is not None and n_id is not None
                       and offset is not None and count is not None
                       and self.partition_ctx is None)
        if (batch_size is not None and not self._async and str(self.emb_device) == 'cpu'
                and str(self.device)[:4] == 'cuda' and self.partition_ctx is None):
            warnings.warn('Asynchronous I/O disabled, although history and '
                          'model sit on different devices.')

        if self._async:
            empty = torch.empty(0, dtype=torch.int64)
            for i in range(len(self.histories)):
                self.pool.async_pull(self.histories[i].emb, offset, count, empty)
                self.pool_ag.async_pull(self.histories_ag[i].emb, offset, count, empty)

        out, t_move, n_ib, n_ob = self.VR_forward(x, adj_t, drift_norm, epoch, batch_idx,
                                                  batch_size, n_id, offset, count, **kwargs)
        self._async = False
        return {'out': out, 'time_pool_pull': 0, 'time_pool_push': 0,
                'time_forward_histories_movement': t_move, 'time_forward_total': 0,
                'num_in_batch_neighbors': n_ib, 'num_out_batch_neighbors': n_ob}

    # ------------------------------------------------------------------ push / pull primitives
    def push_and_pull(self, history, x: Tensor, batch_size: Optional[int] = None,
                      n_id: Optional[Tensor] = None, offset: Optional[Tensor] = None,
                      count: Optional[Tensor] = None) -> Tuple[Tensor, float]:
        r"""Pushes the in-batch rows of :obj:`x` to :obj:`history` and replaces the
        out-of-batch rows by their historical values (models/base.py:380-456)."""
        if n_id is None and x.size(0) != self.num_nodes:
            return x, 0.0  # Do nothing...

        if n_id is None and x.size(0) == self.num_nodes:
            history.push(x)
            return x, 0.0

        assert n_id is not None

        if batch_size is None:
            history.push(x, n_id)
            return x, 0.0

        if not self._async:
            history.push(x[:batch_size].detach(), n_id[:batch_size], offset, count)
            h = history.pull(n_id[batch_size:])
            return torch.cat([x[:batch_size], h], dim=0), 0.0

        out = self.pool.synchronize_pull()[:n_id.numel() - batch_size]
        self.pool.async_push(x[:batch_size].detach(), offset, count, history.emb)
        out = torch.cat([x[:batch_size], out], dim=0)
        self.pool.free_pull()
        return out, 0.0

    def _in_place_ok(self, history, x: Tensor) -> bool:
        if not self.halo_in_place or self._async:
            return False
        if self.partition_ctx is not None:
            # rows of peers must have a place behind the table's own rows (reserve_tail)
            # (decided the same way on every rank: the exchange is collective and rows travel
            # at the table's pitch in this form; the capacity itself is checked by the exchange)
            if self._current_plan is None or history._base is None:
                return False
        if not _same_device(history.emb.device, x.device):
            return False
        layers = [k for k, h in enumerate(self.histories) if h is history]
        if not layers or self._pre_aggregates(layers[0]):   # (the layer this table feeds)
            return False
        drops = self.training and self._layer_dropout and getattr(self, "dropout", 0.0) > 0
        return not drops and history.emb.size(1) >= x.size(1)

    def supports_halo_ready(self) -> bool:
        """Whether ``model(..., halo_ready=True)`` is available: partition-parallel, every
        layer-input table on the compute device with a tail for foreign rows, and a model whose
        out-of-batch rows need nothing done to them before they are aggregated."""
        if self.partition_ctx is None or not self.halo_in_place or self.pool is not None:
            return False
        if any(self._pre_aggregates(k) for k in range(1, self.num_layers)):
            return False
        if self.training and self._layer_dropout and getattr(self, "dropout", 0.0) > 0:
            return False
        return all(h._base is not None and _same_device(h.emb.device, self.device)
                   for h in self.histories[1:])

    def _push_target(self, history, like: Tensor, batch_size: Optional[int],
                     offset: Optional[Tensor], count: Optional[Tensor]) -> Optional[PushTarget]:
        """Where a layer's producer may store the (activated) output itself: the batch is ONE
        range of rows of a table that lives where the model computes, and the push is an
        ordinary local one (no staging pool)."""
        if (batch_size is None or offset is None or count is None or self._async
                or not like.is_cuda or not history.emb.is_cuda
                or not _same_device(history.emb.device, like.device)):
            return None
        lo = self._batch_range(batch_size, offset, count)
        if lo is None or lo + batch_size > history.emb.size(0):
            return None
        return PushTarget(history.emb, lo, relu=True)

    def _push_and_pull_split(self, history, x: Tensor, batch_size: Optional[int],
                             n_id: Optional[Tensor], offset: Optional[Tensor],
                             count: Optional[Tensor], relu: bool = False, pushed: bool = False
                             ) -> Tuple[Tensor, Optional[Tensor], Optional[Tensor]]:
        """:meth:`push_and_pull` without the concatenation: returns the fresh in-batch rows
        and, separately, the out-of-batch rows (``None`` when there are none) plus, when
        those are to be read in place, their row ids: ``(x, table, ids)`` instead of
        ``(x, pulled_rows, None)``.  The aggregation kernel reads its source rows from the
        two pieces, and gradients only ever flow into the first.  With staging pools the
        second piece is a view into slot memory that stays valid until
        :meth:`_release_halo`."""
        if n_id is None or batch_size is None:
            assert not relu
            out, _ = self.push_and_pull(history, x, batch_size, n_id, offset, count)
            return out, None, None
        x = x if x.size(0) == batch_size else x[:batch_size]
        if self.partition_ctx is not None:
            # the local push; the out-of-batch rows were requested at the top of the step
            # (or are fetched now, when nothing was started: direct calls of this method)
            if not pushed:
                history.push(x.detach(), n_id[:batch_size], self._local(offset), count, relu=relu)
            plan = self._current_plan
            if plan is not None and self._in_place_ok(history, x):
                if not getattr(self, "_halo_ready", False):
                    pending = (self._halo_pending.pop(0) if self._halo_pending
                               else plan.start_exchange_in_place(history))
                    pending.wait()
                if plan.n_halo == 0:
                    return x, None, None
                return (x, history.emb_with_tail[:, :x.size(1)],
                        plan.in_place_ids(history))
            if self._halo_pending:
                return x, self._halo_pending.pop(0).wait(), None
            return x, (None if plan is None else plan.exchange(history.emb)), None
        if not self._async:
            if not pushed:
                history.push(x.detach(), n_id[:batch_size], offset, count, relu=relu)
            if n_id.numel() <= batch_size:
                return x, None, None
            if self._in_place_ok(history, x):
                return x, history.emb[:, :x.size(1)], n_id[batch_size:]
            return x, history.pull(n_id[batch_size:]), None
        halo = self.pool.synchronize_pull()[:n_id.numel() - batch_size]
        self.pool.async_push(torch.relu(x.detach()) if relu else x.detach(), offset, count,
                             history.emb)
        self._halo_in_slot = True
        return x, halo, None

    def _release_halo(self) -> None:
        """The kernels reading the pulled rows have been launched: hand the staging slot back
        (its refill is ordered behind them on the device)."""
        if getattr(self, "_halo_in_slot", False):
            self.pool.free_pull()
            self._halo_in_slot = False

    def push_only(self, history, x: Tensor, batch_size: Optional[int] = None,
                  n_id: Optional[Tensor] = None, offset: Optional[Tensor] = None,
                  count: Optional[Tensor] = None) -> Tuple[Tensor, float]:
        """Pushes without pulling (models/base.py:458-499)."""
        if n_id is None and x.size(0) != self.num_nodes:
            return x, 0.0
        if n_id is None and x.size(0) == self.num_nodes:
            history.push(x)
            return x, 0.0
        assert n_id is not None
        if batch_size is None:
            history.push(x, n_id)
            return x, 0.0
        if not self._async:
            history.push(x[:batch_size].detach(), n_id[:batch_size], offset, count)
        else:
            self.pool.async_push(x[:batch_size].detach(), offset, count, history.emb)
        return x[:batch_size], 0.0

    def _pull_batch_rows(self, layer: int, batch_size: int, n_id: Tensor, offset: Tensor,
                         count: Tensor, width: int) -> Tuple[Tensor, Tensor]:
        """The batch's own rows of M_in / M_ag of ``layer`` (what VR_forward consumes,
        models/gcn2.py:246-247).  Views into slot memory when pools are in use."""
        if self._async:
            m_in = self.pool.synchronize_pull()[:batch_size, :width]
            m_ag = self.pool_ag.synchronize_pull()[:batch_size, :width]
            return m_in, m_ag
        if offset is not None and count is not None:
            t_in, t_ag = self.histories[layer], self.histories_ag[layer]
            # one range of rows (partitions that are neighbours in the table form one) of
            # tables that live where the model computes: read where they are (nothing writes
            # the tables during an incremental-aggregation step)
            lo = self._batch_range(batch_size, offset, count)
            if (lo is not None and t_in.emb.is_cuda and t_ag.emb.is_cuda
                    and lo + batch_size <= min(t_in.emb.size(0), t_ag.emb.size(0))
                    and _same_device(t_in.emb.device, self.device)
                    and _same_device(t_ag.emb.device, self.device)):
                return (t_in.emb[lo:lo + batch_size, :width],
                        t_ag.emb[lo:lo + batch_size, :width])
            m_in = t_in.pull_chunks(self._local(offset), count)
            m_ag = t_ag.pull_chunks(self._local(offset), count)
        elif self.partition_ctx is not None:
            raise RuntimeError("partition-parallel batches address their rows by (offset, count)")
        else:
            idx = n_id[:batch_size].to(self.emb_device)
            m_in = self.histories[layer].pull(idx)
            m_ag = self.histories_ag[layer].pull(idx)
        return m_in[:batch_size, :width], m_ag[:batch_size, :width]

    def _batch_range(self, batch_size: int, offset: Tensor, count: Tensor) -> Optional[int]:
        """First table-local row of a batch whose rows are ONE range of the tables, else None
        (worked out once per (offset, count) pair: every layer of a step asks)."""
        key = (id(offset), id(count), batch_size)
        memo = getattr(self, "_batch_range_memo", None)
        if memo is not None and memo[0] == key and memo[1] is offset and memo[2] is count:
            return memo[3]
        lo, rows = ops.merge_row_chunks(self._local(offset), count)
        first = int(lo[0]) if lo.numel() == 1 and int(rows[0]) == batch_size else None
        if first is not None and first < 0:
            first = None
        self._batch_range_memo = (key, offset, count, first)
        return first

    def _free_batch_rows(self) -> None:
        if self._async:
            self.pool.free_pull()
            self.pool_ag.free_pull()

    # ------------------------------------------------------------------ generic forward passes
    def forward(self, x: Tensor, adj_t: SparseTensor, drift_norm: int = 2,
                aggregate_combined: bool = True, use_aggregation: bool = True,
                *args) -> Tuple[Tensor, float]:
        """Push/pull training forward (e.g. models/gcn.py:97-205).  ``args`` =
        (batch_size, n_id, offset, count)."""
        batch_size, n_id, offset, count = (tuple(args) + (None,) * 4)[:4]
        b = adj_t.size(0) if batch_size is None else batch_size
        state: Dict[str, Any] = {}
        # Activation between two layers applied where its output is consumed (models that
        # can: :meth:`_defers_relu`): `x` then holds pre-activations, the aggregation reads
        # max(0, x), the History push stores max(0, x), and the aggregation's backward returns
        # the gradient w.r.t. the pre-activations -- no pass of its own over the rows in
        # either direction.
        state['defer_relu'] = bool(use_aggregation and batch_size is not None
                                   and n_id is not None and self._defers_relu())
        x = self._embed(x, state, b)
        if use_aggregation and not aggregate_combined:
            adj_t = in_batch_columns(adj_t, b)
        if not use_aggregation:
            x = x[:b]
        adj = self._adj(adj_t)
        # out-of-batch rows of the current layer input (never concatenated): pulled rows, or
        # the whole table plus the row ids when they are read in place
        halo = halo_ids = None
        # node ids where the tables live: one upload per batch object, not one per layer and step
        nid = None if n_id is None else _ids_on(adj_t, n_id, self.emb_device)
        relu_pending = False
        for i in range(self.num_layers):
            if relu_pending and self._pre_aggregates(i):
                # this layer aggregates a transformed copy of x: nothing downstream can fold
                # the activation in (the out-of-batch rows were stored activated)
                x, relu_pending = relu_into(x), False
            x = self._layer_input(i, x)
            if not use_aggregation:
                h = self._update(i, x[:b], x, state)
            else:
                if halo is not None and halo_ids is None:
                    halo = self._layer_input(i, halo)
                mix = self._mix(i, state, b)
                xa = self._pre_aggregate(i, x)
                fuse = dict(relu_x=relu_pending,
                            gate_grad=i == 0 and state.get('embed_grad_gated', False))
                if relu_pending and 'grad_blocks' in state:
                    # x is the previous layer's output itself: its gradient is this
                    # aggregation's input gradient, written where the model wants it
                    fuse['grad_out'] = state['grad_blocks'][i - 1]
                if mix is not None:
                    agg = spmm(adj, x, self.agg_reduce, *mix, x_halo=halo, halo_ids=halo_ids,
                               **fuse)
                    h = self._update_mixed(i, agg, x, state)
                elif xa is not None:
                    ha = None if halo is None else self._pre_aggregate(i, halo)
                    agg = spmm(adj, xa, reduce=self.agg_reduce, x_halo=ha)
                    h = self._update(i, agg, x, state, pre=True)
                else:
                    agg = spmm(adj, x, reduce=self.agg_reduce, x_halo=halo, halo_ids=halo_ids,
                               **fuse)
                    if i < self.num_layers - 1 and state['defer_relu']:
                        state['push_into'] = self._push_target(self.histories[i + 1], agg,
                                                               batch_size, offset, count)
                    h = self._update(i, agg, x, state)
                self._release_halo()
            offered = state.pop('push_into', None)
            if i < self.num_layers - 1:
                if use_aggregation:
                    relu_pending = state.pop('relu_deferred', False)
                    pushed = offered is not None and offered.done
                    if pushed and not relu_pending:
                        raise RuntimeError("a layer took the fused push without leaving its "
                                           "activation to it")
                    x, halo, halo_ids = self._push_and_pull_split(
                        self.histories[i + 1], h, batch_size, nid, offset, count,
                        relu=relu_pending, pushed=pushed)
                else:
                    x, _ = self.push_only(self.histories[i + 1], h, batch_size, nid, offset, count)
            else:
                x = h
        return x, 0.0

    def _defers_relu(self) -> bool:
        """Models whose layers end in a relu that nothing but the next aggregation and the
        History push consume may leave it to those two (see :meth:`forward`): when
        ``state['defer_relu']`` is set they skip it in :meth:`_update` and set
        ``state['relu_deferred']`` instead; an embedding whose relu backward the first
        aggregation takes over sets ``state['embed_grad_gated']``."""
        return False

    def VR_forward(self, x: Tensor, adj_t: SparseTensor, drift_norm: int = 2, epoch: int = 0,
                   batch_idx: int = 0, *args):
        """Incremental-aggregation training forward (e.g. models/gcn2.py:187-323):
        every layer computes ``A_ib (x - M_in) + M_ag`` over in-batch edges only."""
        batch_size, n_id, offset, count = (tuple(args) + (None,) * 4)[:4]
        b = batch_size
        state: Dict[str, Any] = {}
        # Activations between two layers applied where they are consumed, as in :meth:`forward`
        # (two-pass form of the incremental aggregate only): `x` then holds pre-activations,
        # the subtraction reads max(0, x), and the aggregation's backward returns the gradient
        # w.r.t. the pre-activations.
        state['defer_relu'] = bool(b is not None and not self.fused_incagg
                                   and self._defers_relu())
        x = self._embed(x, state, b)
        x = x if x.size(0) == b else x[:b]
        adj = self._adj(adj_t)
        relu_pending = False
        for i in range(self.num_layers):
            x = self._layer_input(i, x)
            m_in, m_ag = self._pull_batch_rows(i, b, n_id, offset, count, x.size(1))
            mix = self._mix(i, state, b)
            fuse = {}
            if state['defer_relu']:
                fuse = dict(relu_x=relu_pending,
                            gate_grad=i == 0 and state.get('embed_grad_gated', False))
                if relu_pending and 'grad_blocks' in state:
                    fuse['grad_out'] = state['grad_blocks'][i - 1]
            if mix is None:
                agg = incagg_aggregate(adj, x, m_in, m_ag, reduce=self.agg_reduce,
                                       fused=self.fused_incagg, **fuse)
                x = self._update(i, agg, x, state)
            else:
                mixed = incagg_aggregate(adj, x, m_in, m_ag, self.agg_reduce, *mix,
                                         fused=self.fused_incagg, **fuse)
                x = self._update_mixed(i, mixed, x, state)
            relu_pending = state.pop('relu_deferred', False)
            self._free_batch_rows()
        return x, 0.0, 0, 0

    @torch.no_grad()
    def forward_layer(self, layer: int, x: Tensor, adj_t: SparseTensor, state: Dict[str, Any],
                      use_aggregation: bool = True) -> Tensor:
        """One layer of the layer-wise evaluation (e.g. models/gcn.py:283-332)."""
        b = adj_t.size(0)
        if layer == 0:
            x = self._embed(x, state, b)
        if not use_aggregation:
            x = x[:b]
        x = self._layer_input(layer, x)
        agg = self._aggregate(adj_t, x) if use_aggregation else x
        return self._update(layer, agg, x, state)

    # ------------------------------------------------------------------ refresh sweeps
    @property
    def _out(self):
        if self.__out is None:
            on_host = str(self.emb_device) == 'cpu'
            self.__out = torch.empty(self.num_nodes, self.out_channels, device=self.emb_device,
                                     pin_memory=on_host and torch.cuda.is_available())
        return self.__out

    def _push_rows(self, table: Tensor, rows: Tensor, offset: Tensor, count: Tensor, pool,
                   relu: bool = False) -> None:
        """``relu``: ``rows`` holds pre-activations, ``max(0, rows)`` is what gets stored (on the
        GPU path by the copy kernel itself)."""
        if relu and not (pool is None and rows.is_cuda):
            rows = torch.relu(rows)
            relu = False
        if pool is not None:
            if rows.size(1) < table.size(1):  # pad to the table width (models/gcn.py:355-365)
                rows = F.pad(rows, (0, table.size(1) - rows.size(1)))
            pool.async_push(rows, offset, count, table)
        elif rows.is_cuda:
            ops.copy_row_chunks(table, rows, offset, count, packed_is_src=True, device=rows.device,
                                relu=relu)
        else:
            src = 0
            for o, c in zip(offset.tolist(), count.tolist()):
                table[o:o + c, :rows.size(1)] = rows[src:src + c]
                table[o:o + c, rows.size(1):] = 0
                src += c

    def _pull_layer_input(self, history: History, batch_size: int, n_id: Tensor, offset: Tensor,
                          count: Tensor, plan=None) -> Tensor:
        """[in-batch rows by chunks ; out-of-batch rows by index] of one table, on the
        compute device (what models/gcn.py:378-385 gets from the pool)."""
        m = n_id.numel()
        if self.device.type != 'cuda' and self.partition_ctx is None:
            emb = history.emb
            ib = torch.cat([emb[o:o + c] for o, c in zip(offset.tolist(), count.tolist())])
            return torch.cat([ib, emb.index_select(0, n_id[batch_size:])])[:m]
        out = torch.empty((m, history.embedding_dim), dtype=history.emb.dtype, device=self.device)
        ops.copy_row_chunks(out, history.emb, self._local(offset), count, packed_is_src=False,
                            device=self.device)
        if self.partition_ctx is not None:
            plan.exchange(history.emb, out=out[batch_size:])
        elif m > batch_size:
            ops.gather_rows(history.emb, n_id[batch_size:].to(self.device), out=out[batch_size:],
                            device=self.device)
        return out

    @torch.no_grad()
    def _sweep(self, loader, use_aggregation: bool, with_aggregates: bool) -> Tensor:
        if self.device.type == 'cuda' and not torch.cuda.is_current_stream_capturing():
            from ..graphs import note_eager_sweep
            note_eager_sweep()      # (recorded steps are not replayed after an eager sweep)
        holders = [m for m in self.modules() if hasattr(m, "hold_folded_weights")]
        for m in holders:       # weights are constant for the duration of a sweep
            m.hold_folded_weights(True)
        try:
            return self._sweep_layers(loader, use_aggregation, with_aggregates)
        finally:
            for m in holders:
                m.hold_folded_weights(False)

    def _sweep_layers(self, loader, use_aggregation: bool, with_aggregates: bool) -> Tensor:
        batches = [tuple(sub) + ({}, ) for sub in loader]
        sharded = self.partition_ctx is not None
        use_pool = self.pool is not None and not sharded   # (sharded: see __call__)
        pool = self.pool if use_pool else None
        pool_ag = self.pool_ag if use_pool else None
        last = self.num_layers - 1
        in_place = (self.halo_in_place and not use_pool
                    and _same_device(self.emb_device, self.device))

        defer = self._defers_relu()
        # per-batch results that every layer of the sweep reuses (e.g. GCNII's x_0 summands of
        # all layers) are kept from a batch's first layer to its last when they fit: one
        # [rows, layers * width] matrix per batch, i.e. about one more set of tables
        keep_terms = False
        if self.device.type == 'cuda' and not use_pool:
            if torch.cuda.is_current_stream_capturing():
                # (a recording repeats the decision of the eager sweep that preceded it: the
                # driver is not asked for its free memory inside a recording)
                keep_terms = bool(getattr(self, "_sweep_keeps_terms", False))
            else:
                rows = sum(b[1] for b in batches)
                need = rows * self.num_layers * self.histories[-1].embedding_dim * 4
                keep_terms = need < 0.4 * torch.cuda.mem_get_info(self.device)[0]
                self._sweep_keeps_terms = keep_terms

        def ids_on_device(data, n_id):     # one upload per (static) evaluation batch
            return _ids_on(data.adj_t, n_id, self.device)

        def run_layer(i, x, adj_t, batch_size, offset, count, state, halo=None, halo_ids=None):
            # (partitions that are neighbours in the table count as one range)
            offset, count = ops.merge_row_chunks(self._local(offset), count)
            # a batch that is ONE range of table rows: the aggregation writes its result where
            # the table of aggregates keeps it (no push of its own)
            dest = None
            if (with_aggregates and use_aggregation and pool_ag is None and count.numel() == 1
                    and x.is_cuda and x.dtype == torch.float32):
                tab = self.histories_ag[i].emb
                if _same_device(tab.device, x.device) and tab.size(1) == x.size(1) \
                        and tab.dtype == x.dtype:
                    o = int(offset[0])
                    dest = tab[o:o + batch_size]
            agg = (self._aggregate(adj_t, x, halo, halo_ids, out=dest) if use_aggregation
                   else x[:batch_size])
            if with_aggregates and dest is None:
                self._push_rows(self.histories_ag[i].emb, agg, offset, count, pool_ag)
            # a relu that ends the layer is left to the push where the model allows it (every
            # later reader of these rows -- the next layer of the sweep, training steps --
            # reads them from the table)
            state['defer_relu'] = defer and i < last
            if state['defer_relu'] and pool is None and count.numel() == 1 and use_aggregation:
                hist = self.histories[i + 1]
                if (agg.is_cuda and hist.emb.is_cuda and _same_device(hist.emb.device, agg.device)
                        and int(offset[0]) + batch_size <= hist.emb.size(0)):
                    state['push_into'] = PushTarget(hist.emb, int(offset[0]), relu=True, only=True)
            out = self._update(i, agg, x, state)[:batch_size]
            offered = state.pop('push_into', None)
            relu = state.pop('relu_deferred', False)
            if offered is not None and offered.done:
                if not relu:
                    raise RuntimeError("a layer took the fused push without leaving its "
                                       "activation to it")
                return
            target = self._out if i == last else self.histories[i + 1].emb
            self._push_rows(target, out, offset, count, pool, relu=relu)

        # Models that embed their features row by row in front of layer 0 (GCNII, APPNP): with
        # the tables in HBM every batch embeds only its OWN rows and stores them in
        # ``histories[0]``; layer 0 then aggregates like every other layer, its out-of-batch
        # rows read in place from that table -- instead of embedding the M >> B rows of every
        # batch's neighbourhood over again (the union of the batches covers every node once; the
        # values are the same either way).
        # (only when the batches cover every node: a sweep over part of the graph must not read
        # layer-0 rows that nobody wrote)
        embed_once = (in_place and not sharded and use_aggregation and not self.raw_input_layer
                      and self.device.type == 'cuda'
                      and all(b[1] > 0 for b in batches)
                      and sum(b[1] for b in batches) == self.num_nodes)
        for data, batch_size, n_id, offset, count, state in batches:
            if batch_size == 0:      # padding batch of a partition-parallel run
                continue
            adj_t = data.adj_t.to(self.device)
            state['keep_layer_terms'] = keep_terms
            feats = _features_on(data, self.device)
            x = self._embed(feats[:batch_size] if embed_once else feats, state, batch_size)
            if with_aggregates or embed_once:
                if x.size(1) > self.histories[0].embedding_dim:
                    raise ValueError(
                        f"layer-0 rows are {x.size(1)} wide but the History tables only "
                        f"{self.histories[0].embedding_dim}: keep the tables on the GPU "
                        "(device=...) or choose hidden_channels >= in_channels")
                self._push_rows(self.histories[0].emb, x[:batch_size], self._local(offset), count,
                                pool)
            if not embed_once:
                run_layer(0, x, adj_t, batch_size, offset, count, state)
        if use_pool:
            pool.synchronize_push()
            pool_ag.synchronize_push()

        for i in range(0 if embed_once else 1, self.num_layers):
            if use_pool:  # prime the queue for the whole layer (models/gcn.py:378-379)
                for _, batch_size, n_id, offset, count, _ in batches:
                    pool.async_pull(self.histories[i].emb, offset, count, n_id[batch_size:])
            for data, batch_size, n_id, offset, count, state in batches:
                plan = getattr(data.adj_t, "_halo_plan", None) if sharded else None
                # (every rank takes the same form of the exchange: rows travel at the table's
                # pitch when they are received in place, at its width otherwise)
                tail_ok = plan is not None and in_place and use_aggregation \
                    and self.histories[i]._base is not None
                if batch_size == 0:  # nothing of our own left, but peers still need serving
                    if tail_ok:
                        plan.start_exchange_in_place(self.histories[i]).wait()
                    else:
                        plan.exchange(self.histories[i].emb)
                    continue
                adj_t = data.adj_t.to(self.device)
                halo = halo_ids = None
                if use_pool:
                    x = pool.synchronize_pull()[:n_id.numel()]
                elif tail_ok or (plan is None and in_place and use_aggregation
                                 and n_id.numel() > batch_size):
                    # in-batch rows by chunks (the update needs them); out-of-batch rows are
                    # read from the table by the aggregation itself
                    hist = self.histories[i]
                    loc, cnt = ops.merge_row_chunks(self._local(offset), count)
                    if cnt.numel() == 1:        # one contiguous range: the table rows themselves
                        o = int(loc[0])
                        x = hist.emb[o:o + batch_size]
                    else:
                        x = torch.empty((batch_size, hist.embedding_dim), dtype=hist.emb.dtype,
                                        device=self.device)
                        ops.copy_row_chunks(x, hist.emb, loc, cnt, packed_is_src=False,
                                            device=self.device)
                    if plan is None:
                        halo = hist.emb
                        halo_ids = ids_on_device(data, n_id)[batch_size:]
                    else:   # partition-parallel: peers' rows arrive behind the table's own
                        plan.start_exchange_in_place(hist).wait()
                        if plan.n_halo:
                            halo = hist.emb_with_tail
                            halo_ids = plan.in_place_ids(hist)
                else:
                    x = self._pull_layer_input(self.histories[i], batch_size, n_id, offset, count,
                                               plan)
                run_layer(i, x, adj_t, batch_size, offset, count, state, halo, halo_ids)
                if use_pool:
                    pool.free_pull()
            if use_pool:
                pool.synchronize_push()
                pool_ag.synchronize_push()
        return self._out

    @torch.no_grad()
    def mini_inference(self, loader, use_aggregation: bool = True) -> Tensor:
        """Layer-wise evaluation that refreshes ``histories`` (models/base.py:509-603)."""
        return self._sweep(loader, use_aggregation, with_aggregates=False)

    @torch.no_grad()
    def mini_inference_vr(self, loader, use_aggregation: bool = True) -> Tensor:
        """Layer-wise evaluation that refreshes both ``histories`` (M_in) and
        ``histories_ag`` (M_ag) (e.g. models/gcn2.py:433-507).  The aggregate of a
        layer's input is computed once and used for M_ag and for the layer itself."""
        return self._sweep(loader, use_aggregation, with_aggregates=True)


def _ids_on(holder, n_id: Tensor, device) -> Tensor:
    """``n_id`` on ``device``, uploaded once per (holder, id tensor) and kept on the holder --
    the batch's adjacency -- together with the tensor it was copied from, so that every step
    over the same batch object hands the kernels the same device array."""
    device = torch.device(device)
    if n_id.device == device or (n_id.is_cuda and device.type == "cuda" and device.index is None):
        return n_id
    kept = holder.__dict__.setdefault("_ids_on_device", {})
    hit = kept.get(device)
    if hit is None or hit[0] is not n_id or hit[2] != n_id._version:
        hit = kept[device] = (n_id, n_id.to(device), n_id._version)
    return hit[1]


def _features_on(data, device) -> Tensor:
    """``data.x`` on ``device``.  Features of a (static) evaluation batch that live on the host
    are uploaded once, through pinned memory, and kept on the batch -- every later sweep over the
    same batch object reads the resident copy (no host-to-device copy per sweep, and nothing a
    recorded sweep would have to copy from the host) -- as long as HBM has ample room for them;
    otherwise they are uploaded per use."""
    x = data.x
    device = torch.device(device)
    if x.device == device or device.type != "cuda" or (x.is_cuda and device.index is None):
        return x.to(device)
    kept = data.__dict__.setdefault("_x_on_device", {})
    hit = kept.get(device)
    if hit is not None and hit[0] is x and hit[2] == x._version:
        return hit[1]
    staged = x if x.is_cuda or x.is_pinned() else x.pin_memory()
    # (torch's pinned-memory allocator keeps the staging block until the transfer is over)
    on_dev = staged.to(device, non_blocking=True)
    if x.numel() * x.element_size() * 16 < torch.cuda.mem_get_info(device)[0]:
        kept[device] = (x, on_dev, x._version)
    return on_dev


def _same_device(a, b) -> bool:
    a, b = torch.device(a), torch.device(b)
    return a.type == b.type and (a.index or 0) == (b.index or 0)


def in_batch_columns(adj_t: SparseTensor, batch_size: int) -> SparseTensor:
    """Edges whose column is an in-batch node, same sparse sizes (the ``in_batch_adj`` of
    models/gcn.py:117-126), built once per adjacency instead of once per step."""
    cache = getattr(adj_t, "_in_batch_view", None)
    if cache is not None and cache[0] == batch_size:
        return cache[1]
    host = adj_t._host if adj_t._host is not None else adj_t
    row, col, value = host.coo()
    keep = col < batch_size
    view = SparseTensor(row=row[keep], col=col[keep],
                        value=None if value is None else value[keep],
                        sparse_sizes=host.sparse_sizes(), is_sorted=True).to(adj_t.device())
    adj_t._in_batch_view = (batch_size, view)
    return view
