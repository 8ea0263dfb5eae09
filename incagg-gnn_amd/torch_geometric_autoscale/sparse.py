"""A minimal CSR container with the part of the ``torch_sparse.SparseTensor`` surface the
reference touches (loader.py:180-185, models/gcn.py:117-135, models/graphsage.py:631,
main.py:62,148,182, metis.py:58, data.py:59, utils.py:62-73).

On the host it holds int64 ``rowptr`` / ``col`` (the reference's layout).  On the GPU
the primary storage is int32 (halves index traffic in the aggregation kernels); int64
views are materialised only if somebody asks for them.  The transposed CSR needed by
the backward aggregation is built once per adjacency and cached.
"""
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import ops


class _Storage:
    """``adj.storage.row() / col() / value() / rowptr() / rowcount()``."""

    def __init__(self, owner: "SparseTensor"):
        self._o = owner

    def row(self) -> Tensor:
        return self._o._coo_row()

    def col(self) -> Tensor:
        return self._o._col64()

    def value(self) -> Optional[Tensor]:
        return self._o._value

    def rowptr(self) -> Tensor:
        return self._o._rowptr64()

    def rowcount(self) -> Tensor:
        rp = self._o._rowptr64()
        return rp[1:] - rp[:-1]

    def sparse_sizes(self) -> Tuple[int, int]:
        return self._o._sizes


class SparseTensor:
    def __init__(self, row: Optional[Tensor] = None, rowptr: Optional[Tensor] = None,
                 col: Optional[Tensor] = None, value: Optional[Tensor] = None,
                 sparse_sizes: Optional[Tuple[int, int]] = None, is_sorted: bool = False,
                 trust_data: bool = False):
        if col is None:
            raise ValueError("col is required")
        if rowptr is None and row is None:
            raise ValueError("either row or rowptr is required")
        col = col.to(torch.int64)
        if rowptr is None:
            row = row.to(torch.int64)
            n_rows = sparse_sizes[0] if sparse_sizes is not None else (
                int(row.max()) + 1 if row.numel() else 0)
            if not is_sorted and row.numel():
                n_c = (sparse_sizes[1] if sparse_sizes is not None else int(col.max()) + 1)
                perm = torch.argsort(row * max(n_c, 1) + col, stable=True)
                row, col = row[perm], col[perm]
                value = None if value is None else value[perm]
            rowptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=col.device)
            if row.numel():
                rowptr[1:] = torch.cumsum(torch.bincount(row, minlength=n_rows), 0)
        else:
            rowptr = rowptr.to(torch.int64)
        n_rows = rowptr.numel() - 1
        if sparse_sizes is None:
            n_cols = int(col.max()) + 1 if col.numel() else 0
            sparse_sizes = (n_rows, n_cols)
        self._sizes = (int(sparse_sizes[0]), int(sparse_sizes[1]))
        self._device = col.device
        self._rowptr_i64: Optional[Tensor] = rowptr
        self._col_i64: Optional[Tensor] = col
        self._rowptr_i32: Optional[Tensor] = None
        self._col_i32: Optional[Tensor] = None
        self._value = value
        self._row: Optional[Tensor] = None
        self._host: Optional["SparseTensor"] = None   # CPU original of a GPU copy
        self._transposed = {}                         # reduce -> (rowptr32, col32, val)
        self._t_shared = {}                           # structure of the transposed CSR (see set_value)
        self._max_row_nnz: Optional[int] = None
        self.storage = _Storage(self)

    @classmethod
    def from_device_csr(cls, rowptr32: Tensor, col32: Tensor, value: Optional[Tensor],
                        sparse_sizes: Tuple[int, int]) -> "SparseTensor":
        """Wraps a device-format CSR (int32 indices, fp32 values, already on the GPU) as built
        by the device batch builder: no host copy exists, the transposed CSR and the row
        orders are derived on the device."""
        if rowptr32.dtype != torch.int32 or col32.dtype != torch.int32 or not rowptr32.is_cuda:
            raise TypeError("expected int32 device tensors")
        out = cls.__new__(cls)
        out._sizes = (int(sparse_sizes[0]), int(sparse_sizes[1]))
        out._device = rowptr32.device
        out._rowptr_i64 = out._col_i64 = None
        out._rowptr_i32, out._col_i32 = rowptr32, col32
        out._value = value
        out._row = None
        out._host = None
        out._transposed = {}
        out._t_shared = {}
        out._max_row_nnz = None
        out.storage = _Storage(out)
        return out

    # ------------------------------------------------------------------ accessors
    def _rowptr64(self) -> Tensor:
        if self._rowptr_i64 is None:
            self._rowptr_i64 = self._rowptr_i32.to(torch.int64)
        return self._rowptr_i64

    def _col64(self) -> Tensor:
        if self._col_i64 is None:
            self._col_i64 = self._col_i32.to(torch.int64)
        return self._col_i64

    def _rowptr32(self) -> Tensor:
        if self._rowptr_i32 is None:
            if self.nnz() >= 2 ** 31 - 1:
                raise ValueError("adjacency too large for int32 device indices")
            self._rowptr_i32 = self._rowptr_i64.to(torch.int32)
        return self._rowptr_i32

    def _col32(self) -> Tensor:
        if self._col_i32 is None:
            self._col_i32 = self._col_i64.to(torch.int32)
        return self._col_i32

    def _coo_row(self) -> Tensor:
        if self._row is None:
            rp = self._rowptr64()
            self._row = torch.repeat_interleave(
                torch.arange(self._sizes[0], device=rp.device), rp[1:] - rp[:-1])
        return self._row

    def csr(self) -> Tuple[Tensor, Tensor, Optional[Tensor]]:
        return self._rowptr64(), self._col64(), self._value

    def coo(self) -> Tuple[Tensor, Tensor, Optional[Tensor]]:
        return self._coo_row(), self._col64(), self._value

    def size(self, dim: int) -> int:
        return self._sizes[dim]

    def sizes(self):
        return list(self._sizes)

    def sparse_sizes(self) -> Tuple[int, int]:
        return self._sizes

    def sparse_size(self, dim: int) -> int:
        return self._sizes[dim]

    def nnz(self) -> int:
        c = self._col_i64 if self._col_i64 is not None else self._col_i32
        return int(c.numel())

    def has_value(self) -> bool:
        return self._value is not None

    def device(self):
        return self._device

    def max_row_nnz(self) -> int:
        if self._max_row_nnz is None:
            if self._host is not None:
                self._max_row_nnz = self._host.max_row_nnz()
            else:
                rc = self.storage.rowcount()
                self._max_row_nnz = int(rc.max()) if rc.numel() else 0
        return self._max_row_nnz

    # ------------------------------------------------------------------ movement
    def to(self, device=None, *args, non_blocking: bool = False, **kwargs) -> "SparseTensor":
        if device is None or isinstance(device, torch.dtype):
            return self
        device = torch.device(device)
        if device == self._device or (device.type == "cuda" and self._device.type == "cuda"
                                       and device.index is None):
            return self
        if device.type == "cuda":
            if device.index is None:
                device = torch.device("cuda", torch.cuda.current_device())
            # One upload per adjacency: evaluation batches are static and are revisited
            # once per layer and per epoch, and HBM has room for all of them.
            copies = self.__dict__.setdefault("_device_copies", {})
            if device in copies:
                return copies[device]
            # Convert to the device format on the host, then upload 4-byte indices.
            out = SparseTensor.__new__(SparseTensor)
            out._sizes = self._sizes
            out._device = device if device.index is not None else torch.device(
                "cuda", torch.cuda.current_device())
            out._rowptr_i64 = out._col_i64 = None
            out._rowptr_i32 = self._rowptr32().to(out._device, non_blocking=non_blocking)
            out._col_i32 = self._col32().to(out._device, non_blocking=non_blocking)
            out._value = None if self._value is None else self._value.to(
                out._device, non_blocking=non_blocking)
            out._row = None
            out._host = self if self._device.type == "cpu" else self._host
            out._transposed = {}
            out._t_shared = {}
            out._max_row_nnz = self.max_row_nnz()
            # the aggregation's work plan is part of batch construction: built on the host
            # from the row pointers, uploaded with them
            out._plans = {"rows": ops.SpmmPlan.build(self._rowptr32(), self.nnz()).to(
                out._device, non_blocking=non_blocking)}
            out.storage = _Storage(out)
            if hasattr(self, "_halo_plan"):
                out._halo_plan = self._halo_plan
            copies[device] = out
            return out
        rowptr, col, value = self.csr()
        out = SparseTensor(rowptr=rowptr.to(device), col=col.to(device),
                           value=None if value is None else value.to(device),
                           sparse_sizes=self._sizes, is_sorted=True)
        return out

    def cpu(self):
        return self.to("cpu")

    def cuda(self, device=None):
        return self.to("cuda" if device is None else device)

    # ------------------------------------------------------------------ edits (host-side graph prep)
    # What a copy with other edge values may share with its source: sizes, index arrays and
    # everything derived from the STRUCTURE alone -- the work plans, the in-place column ids, and
    # the index arrays of the transposed CSR together with the permutation that carries edge
    # values into its order (``_t_shared``).  Everything that holds values (transposed values,
    # packed edge lists, uploaded copies, the in-batch view, the value-less view) is rebuilt on
    # demand: edge dropout on a weighted adjacency (utils.py:62-68, applied at every training
    # step, main.py:68) then costs one gather of the new values and the re-packing of the edge
    # lists, not a transposition and not a plan.
    _STRUCTURE_KEYS = ("_sizes", "_device", "_rowptr_i64", "_col_i64", "_rowptr_i32", "_col_i32",
                       "_row", "_max_row_nnz", "_plans", "_in_place_cols", "_halo_plan",
                       "_ids_on_device", "_t_shared")

    def set_value(self, value: Optional[Tensor], layout: Optional[str] = None) -> "SparseTensor":
        out = SparseTensor.__new__(SparseTensor)
        for k in self._STRUCTURE_KEYS:
            if k in self.__dict__:
                out.__dict__[k] = self.__dict__[k]
        out._max_row_nnz = self.max_row_nnz() if self._host is not None else self._max_row_nnz
        out._value = value
        out._transposed = {}
        # a device copy with new values no longer mirrors its host original: its own arrays
        # are the source of everything derived from now on (no device-to-host copy here)
        out._host = None
        out.storage = _Storage(out)
        return out

    def set_value_(self, value: Optional[Tensor], layout: Optional[str] = None) -> "SparseTensor":
        if self._host is not None:
            self._max_row_nnz = self.max_row_nnz()
        self._value = value
        self._transposed = {}
        self._host = None
        for k in ("_in_batch_view", "_device_copies", "_unweighted_view", "_packed_edges",
                  "_plan_twins"):
            self.__dict__.pop(k, None)
        return self

    def set_diag(self, values: Optional[Tensor] = None) -> "SparseTensor":
        """Replaces the diagonal by ones (or ``values``): main.py:148."""
        row, col, value = self.coo()
        n = min(self._sizes)
        keep = row != col
        diag = torch.arange(n, device=row.device)
        new_row = torch.cat([row[keep], diag])
        new_col = torch.cat([col[keep], diag])
        new_val = None
        if value is not None:
            dv = torch.ones(n, dtype=value.dtype, device=value.device) if values is None else values
            new_val = torch.cat([value[keep], dv])
        return SparseTensor(row=new_row, col=new_col, value=new_val, sparse_sizes=self._sizes)

    def fill_value(self, fill: float, dtype=torch.float32) -> "SparseTensor":
        return self.set_value(torch.full((self.nnz(),), fill, dtype=dtype, device=self._device))

    def permute(self, perm: Tensor) -> "SparseTensor":
        """Symmetric permutation: out[i, j] = self[perm[i], perm[j]] (metis.py:58)."""
        row, col, value = self.coo()
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(perm.numel(), device=perm.device)
        return SparseTensor(row=inv[row], col=inv[col], value=value, sparse_sizes=self._sizes)

    def t(self) -> "SparseTensor":
        row, col, value = self.coo()
        return SparseTensor(row=col, col=row, value=value,
                            sparse_sizes=(self._sizes[1], self._sizes[0]))

    def to_symmetric(self) -> "SparseTensor":
        """Union with the transpose, duplicates merged (data.py:59)."""
        row, col, value = self.coo()
        n = max(self._sizes)
        r = torch.cat([row, col])
        c = torch.cat([col, row])
        key = torch.unique(r * n + c)
        if value is not None:
            raise NotImplementedError("to_symmetric with values")
        return SparseTensor(row=key // n, col=key % n, sparse_sizes=(n, n), is_sorted=True)

    def masked_select_nnz(self, mask: Tensor, layout: Optional[str] = None) -> "SparseTensor":
        row, col, value = self.coo()
        return SparseTensor(row=row[mask], col=col[mask],
                            value=None if value is None else value[mask],
                            sparse_sizes=self._sizes, is_sorted=True)

    def sum(self, dim: int) -> Tensor:
        row, col, value = self.coo()
        idx = row if dim == 1 else col
        w = torch.ones(idx.numel(), device=idx.device) if value is None else value
        return torch.zeros(self._sizes[1 - dim], dtype=w.dtype, device=w.device).index_add_(0, idx, w)

    # ------------------------------------------------------------------ device-format views
    def device_csr(self):
        """(rowptr int32, col int32, value fp32 | None) on the GPU."""
        if self._device.type != "cuda":
            raise RuntimeError("aggregation runs on the GPU only: move the adjacency with .to(device)")
        v = self._value
        if v is not None and v.dtype != torch.float32:
            v = v.to(torch.float32)
        return self._rowptr32(), self._col32(), v

    def col_in_place(self, halo_ids: Tensor, n_in_batch: int) -> Tensor:
        """Column ids for reading out-of-batch source rows straight from a History table:
        columns below ``n_in_batch`` unchanged, column ``c`` above it becomes
        ``n_in_batch + halo_ids[c - n_in_batch]`` (int32, device; built once per adjacency
        and id vector)."""
        # ONE array per (id count, batch size), kept for the life of the adjacency and only
        # ever rewritten in place: a recorded HIP graph holds the address of the array it was
        # recorded with, so the array is never freed and never replaced.  The id vector it was
        # filled from is kept too (its address then cannot be handed to another tensor); a call
        # with another id vector refills the same array.
        key = (halo_ids.numel(), n_in_batch)
        cache = self.__dict__.setdefault("_in_place_cols", {})
        hit = cache.get(key)
        stamp = (halo_ids.data_ptr(), halo_ids._version)
        if hit is not None and hit[1] == stamp:
            return hit[0]
        if halo_ids.numel() < self._sizes[1] - n_in_batch:
            raise ValueError("halo_ids does not cover the out-of-batch columns")
        col = self._col32().to(torch.int64)
        ids = halo_ids.to(device=col.device, dtype=torch.int64)
        moved = n_in_batch + ids[(col - n_in_batch).clamp_(min=0)] if ids.numel() else col
        fresh = torch.where(col < n_in_batch, col, moved).to(torch.int32)
        if hit is None:
            out = fresh
        else:
            out = hit[0]
            out.copy_(fresh)
        cache[key] = (out, stamp, halo_ids)
        return out

    def plan(self) -> "ops.SpmmPlan":
        """Work plan of the aggregation over this adjacency (rows by descending length, long
        rows cut into workgroup items), on the device.  Built on the host when the adjacency
        is uploaded (part of batch construction), else on first use, on the device."""
        cache = self.__dict__.setdefault("_plans", {})
        if "rows" not in cache:
            cache["rows"] = ops.SpmmPlan.build(self._rowptr32(), self.nnz())
        return cache["rows"]

    def _with_packed_edges(self, base: "ops.SpmmPlan", tag, twin_key, col: Tensor,
                           val: Optional[Tensor], stamp=None) -> "ops.SpmmPlan":
        """``base`` carrying the packed edge list of (``col``, ``val``).  The packed arrays hold
        edge VALUES, so they live on this object (not with the structure shared between
        copies that differ in their values) and, like every array a recorded HIP graph may
        hold, are allocated once per ``tag`` and refilled in place when ``stamp`` changes."""
        if not col.is_cuda:
            return base
        packed = self.__dict__.setdefault("_packed_edges", {})
        hit = packed.get(tag)
        if hit is None or hit[1] != stamp:
            packed[tag] = hit = (ops.pack_edges(col, val, None if hit is None else hit[0]), stamp)
        twins = self.__dict__.setdefault("_plan_twins", {})
        twin = twins.get(twin_key)
        if twin is None or twin.edges is not hit[0] or twin.order is not base.order:
            twin = twins[twin_key] = base.with_edges(hit[0])
        return twin

    def plan_edges(self) -> "ops.SpmmPlan":
        """:meth:`plan` with the packed edge list of :meth:`device_csr`."""
        _, col, val = self.device_csr()
        return self._with_packed_edges(self.plan(), "rows", "rows", col, val)

    def plan_edges_in_place(self, halo_ids: Tensor, n_in_batch: int) -> "ops.SpmmPlan":
        """:meth:`plan` with the packed edge list of :meth:`col_in_place`'s column ids."""
        col = self.col_in_place(halo_ids, n_in_batch)
        key = ("in_place", halo_ids.numel(), n_in_batch)
        stamp = (halo_ids.data_ptr(), halo_ids._version)
        return self._with_packed_edges(self.plan(), key, key, col, self.device_csr()[2], stamp)

    def plan_t(self, reduce: str, lo: int, hi: int) -> "ops.SpmmPlan":
        """The same for rows [lo, hi) of the transposed CSR (device_csr_t prepares the two
        windows the backward pass uses: all columns, and the in-batch columns), with the
        packed edge list of the transposed arrays."""
        cache = self.__dict__.setdefault("_plans", {})
        key = ("t", lo, hi)
        t_rowptr, t_col, t_val = self.device_csr_t(reduce)
        if key not in cache:
            cache[key] = ops.SpmmPlan.build(t_rowptr[lo:hi + 1], self.nnz())
        red = "mean" if reduce == "mean" else "sum"
        return self._with_packed_edges(cache[key], ("t", red), ("t", red, lo, hi), t_col, t_val)

    def device_csr_t(self, reduce: str = "sum"):
        """Transposed CSR for dX = A^T dY, with 1/len(row) folded in for 'mean'."""
        key = "mean" if reduce == "mean" else "sum"
        if key not in self._transposed:
            n_r, n_c = self._sizes
            shared = self.__dict__.setdefault("_t_shared", {})
            if key == "sum" and shared.get("csr") is not None and self._device.type == "cuda":
                # an adjacency that shares its structure with one transposed before (a copy with
                # other edge values): only the values are carried over, in the transposed order
                # (edges sorted by column, ties in storage order)
                t_rowptr, t_col = shared["csr"]
                if self._value is None:
                    t_val = torch.ones(t_col.numel(), dtype=torch.float32, device=self._device)
                else:
                    if "perm" not in shared:
                        shared["perm"] = torch.argsort(self._col32(), stable=True)
                    t_val = self._value.to(torch.float32).index_select(0, shared["perm"])
                self._transposed[key] = (t_rowptr, t_col, t_val)
                return self._transposed[key]
            windows = {(0, n_c), (0, min(n_r, n_c))}
            plans = self.__dict__.setdefault("_plans", {})
            if self._host is None and self._device.type == "cuda":
                # built on the device: transposed there too (same element order)
                t_rowptr, t_col, t_val = ops.csr_transpose_device(
                    self._rowptr32(), self._col32(), self._value, n_r, n_c, key)
                self._transposed[key] = (t_rowptr, t_col, t_val)
                if key == "sum":
                    shared["csr"] = (t_rowptr, t_col)
                for lo, hi in windows:
                    plans[("t", lo, hi)] = ops.SpmmPlan.build(t_rowptr[lo:hi + 1], self.nnz())
                return self._transposed[key]
            src = self._host if self._host is not None else self
            rowptr, col, value = src.csr()
            t_rowptr, t_col, t_val = ops.csr_transpose(rowptr, col, value, n_r, n_c, key)
            dev = self._device
            self._transposed[key] = (t_rowptr.to(dev), t_col.to(dev), t_val.to(dev))
            if key == "sum" and dev.type == "cuda":
                shared["csr"] = self._transposed[key][:2]
            if dev.type == "cuda":
                for lo, hi in windows:
                    plans[("t", lo, hi)] = ops.SpmmPlan.build(t_rowptr[lo:hi + 1],
                                                              self.nnz()).to(dev)
        return self._transposed[key]

    # ------------------------------------------------------------------ aggregation
    def matmul(self, other: Tensor, reduce: str = "sum") -> Tensor:
        from .aggregate import spmm
        return spmm(self, other, reduce)

    def __matmul__(self, other: Tensor) -> Tensor:
        return self.matmul(other, "sum")

    def spmm(self, other: Tensor, reduce: str = "sum") -> Tensor:
        return self.matmul(other, reduce)

    def __repr__(self) -> str:
        return (f"SparseTensor(sparse_sizes={self._sizes}, nnz={self.nnz()}, "
                f"has_value={self._value is not None}, device={self._device})")
