"""ctypes binding of ``libincagg_gnn.so`` (the C ABI declared in ``include/incagg_gnn.h``).

The library is the only implementation of the hot path: there is no Python or
PyTorch fallback.  If the shared object is missing every entry point raises.
"""
import ctypes
import os

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get(
    "INCAGG_GNN_LIB", os.path.join(os.path.dirname(_PKG_DIR), "lib", "libincagg_gnn.so"))

_c = ctypes
_i64, _i32, _f32, _vp, _int = _c.c_int64, _c.c_int32, _c.c_float, _c.c_void_p, _c.c_int

# name -> (restype, argtypes); mirrors include/incagg_gnn.h one to one.
SIGNATURES = {
    "incagg_last_error": (_c.c_char_p, []),
    "incagg_abi_version": (_int, []),
    "incagg_device_count": (_int, []),
    "incagg_check_device_errors": (_int, []),
    "incagg_spmm_csr_f32": (_int, [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _i64,
                                   _int, _f32, _vp, _i64, _f32, _i64, _vp, _vp, _vp, _i64, _vp]),
    "incagg_spmm_csr_incagg_f32": (_int, [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _i64,
                                          _i64, _int, _f32, _vp, _i64, _f32, _i64, _vp, _vp,
                                          _i64, _vp]),
    "incagg_spmm_plan_defaults": (_int, [_vp, _vp]),
    "incagg_spmm_plan_chunk_rows": (_i32, [_i64]),
    "incagg_spmm_plan_items_bound": (_i64, [_i64, _i64, _i32, _i32]),
    "incagg_spmm_plan_host": (_int, [_vp, _i64, _i32, _i32, _vp, _vp, _i64, _vp, _i64, _vp]),
    "incagg_spmm_plan_device_scratch_bytes": (_i64, [_i64]),
    "incagg_spmm_plan_device": (_int, [_vp, _i64, _i32, _i32, _vp, _i64, _vp, _vp, _i64, _vp,
                                       _i64, _vp, _vp]),
    "incagg_spmm_plan_device_static_max_rows": (_i64, []),
    "incagg_spmm_plan_device_static": (_int, [_vp, _i64, _i32, _i32, _vp, _vp, _i64, _vp, _i64,
                                              _vp, _vp]),
    "incagg_set_tuning": (_int, [_c.c_char_p, _int]),
    "incagg_rows_sub_f32": (_int, [_vp, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _vp]),
    "incagg_rows_sum_f32": (_int, [_vp, _vp, _i64, _f32, _i64, _i64, _vp, _i64, _vp]),
    "incagg_rows_sub_relu_f32": (_int, [_vp, _i64, _vp, _i64, _i64, _i64, _int, _vp, _i64, _vp]),
    "incagg_tn_finish_f32": (_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _vp, _i64, _vp]),
    "incagg_tn_finish_batched_f32": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "incagg_relu_rows_f32": (_int, [_vp, _i64, _i64, _i64, _vp]),
    "incagg_batch_norm_supported": (_int, [_i64, _i64]),
    "incagg_batch_norm_ws_floats": (_i64, [_i64, _i64]),
    "incagg_batch_norm_forward_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _f32, _f32, _vp, _vp,
                                             _vp, _i64, _vp, _vp, _vp, _i64, _vp]),
    "incagg_batch_norm_backward_f32": (_int, [_vp, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp,
                                              _i64, _vp, _vp, _vp, _i64, _vp]),
    "incagg_rows_gemm_supported": (_int, [_i64, _i64]),
    "incagg_rows_gemm_f32": (_int, [_vp, _i64, _vp, _i64, _int, _vp, _i64, _i64, _i64, _int, _vp,
                                    _i64, _vp, _i64, _i64, _i64, _int, _int, _vp]),
    "incagg_gate_add_rows_f32": (_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _vp]),
    "incagg_gate_rows_f32": (_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _vp]),
    "incagg_gcn2_fold_weights_f32": (_int, [_vp, _i64, _i64, _f32, _vp, _vp, _vp, _vp]),
    "incagg_gcn2_fold_weights_backward_f32": (_int, [_vp, _vp, _i64, _i64, _f32, _vp, _vp, _vp]),
    "incagg_column_sums_ws_floats": (_i64, [_i64, _i64]),
    "incagg_column_sums_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _i64, _vp, _vp]),
    "incagg_softmax_xent_ws_floats": (_i64, [_i64]),
    "incagg_softmax_xent_forward_f32": (_int, [_vp, _i64, _vp, _i64, _i64, _vp, _i64, _vp, _vp]),
    "incagg_softmax_xent_backward_f32": (_int, [_vp, _i64, _vp, _i64, _i64, _vp, _vp, _vp, _i64, _vp]),
    "incagg_adam_step_f32": (_int, [_vp, _i64, _f32, _f32, _f32, _f32, _f32, _vp]),
    "incagg_gather_rows_f32": (_int, [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _vp]),
    "incagg_scatter_rows_f32": (_int, [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _vp]),
    "incagg_copy_row_chunks_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _vp,
                                          _i64, _int, _int, _vp]),
    "incagg_stage_read_async": (_int, [_vp, _i64, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp,
                                       _i64, _vp]),
    "incagg_stage_synchronize": (_int, []),
    "incagg_stage_write_async": (_int, [_vp, _i64, _i64, _vp, _vp, _i64, _vp, _i64, _vp]),
    "incagg_stage_pending": (_i64, []),
    "incagg_relabel_bound": (_i64, [_vp, _i64, _vp, _i64]),
    "incagg_relabel_one_hop": (_int, [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _int, _vp, _vp, _vp,
                                      _vp, _vp]),
    "incagg_relabel_one_hop_within_batch": (_int, [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _int, _vp,
                                                   _vp, _vp, _vp, _vp]),
    "incagg_csr_transpose": (_int, [_vp, _vp, _vp, _i64, _i64, _int, _vp, _vp, _vp]),
    "incagg_partition_bfs": (_int, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "incagg_partition_refine": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "incagg_relabel_device_scratch_bytes": (_i64, [_i64, _i64]),
    "incagg_relabel_bound_device": (_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp]),
    "incagg_relabel_one_hop_device": (_int, [_vp, _i64, _vp, _vp, _vp, _i64, _int, _i64, _i64, _vp,
                                             _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "incagg_csr_transpose_device_scratch_bytes": (_i64, [_i64, _i64]),
    "incagg_csr_transpose_device": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _int, _vp, _i64, _vp,
                                           _vp, _vp, _vp]),
}

class SpmmFuseStruct(_c.Structure):
    """``incagg_spmm_fuse`` of include/incagg_gnn.h."""
    _fields_ = [("relu_x", _i32), ("gate", _vp), ("ld_gate", _i64),
                ("add2", _vp), ("ld_add2", _i64), ("add2_scale", _f32)]


REDUCE = {"sum": 0, "add": 0, "mean": 1}
ABI_VERSION = 8


class AdamSegment(_c.Structure):
    """``incagg_adam_segment`` of include/incagg_gnn.h."""
    _fields_ = [("p", _vp), ("g", _vp), ("m", _vp), ("v", _vp), ("step", _vp), ("n", _i64)]


class SpmmPlanStruct(_c.Structure):
    """``incagg_spmm_plan`` of include/incagg_gnn.h."""
    _fields_ = [("order", _vp), ("items", _vp), ("multi_rows", _vp), ("n_rows", _i64),
                ("n_hub", _i64), ("n_items", _i64), ("n_multi", _i64), ("hub_threshold", _i32),
                ("segment_edges", _i32), ("combine", _vp), ("combine_floats", _i64),
                ("edges", _vp)]


COMBINE_SLOT_FLOATS = 1056      # INCAGG_SPMM_COMBINE_SLOT_FLOATS

_handle = None


class IncAggLibraryError(RuntimeError):
    pass


class IncAggIndexError(RuntimeError, IndexError):
    """Out-of-range row / chunk: the reference's "Invalid index" assertion
    (csrc/cuda/async_cuda.cu:77-78,148-149), which surfaces as a RuntimeError."""


def lib():
    """The loaded library; raises if it has not been built."""
    global _handle
    if _handle is None:
        if not os.path.isfile(LIB_PATH):
            raise IncAggLibraryError(
                f"{LIB_PATH} not found: build it with `make -C incagg-gnn_amd/csrc` "
                "(there is no fallback implementation)")
        h = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)  # AttributeError here means header and library disagree
            fn.restype = res
            fn.argtypes = args
        if h.incagg_abi_version() != ABI_VERSION:
            raise IncAggLibraryError("libincagg_gnn.so ABI version mismatch")
        _handle = h
    return _handle


def last_error() -> str:
    return lib().incagg_last_error().decode("utf-8", "replace")


def check(rc: int) -> None:
    """Turns a negative status into the exception type the reference raises
    (its AT_ASSERTM failures surface as RuntimeError)."""
    if rc < 0:
        msg = last_error()
        if rc == -4:
            raise IncAggIndexError(msg)
        raise RuntimeError(msg)


def set_tuning(key: str, value: int) -> None:
    check(lib().incagg_set_tuning(key.encode(), int(value)))


def device_count() -> int:
    return int(lib().incagg_device_count())


def check_device_errors() -> None:
    check(lib().incagg_check_device_errors())
