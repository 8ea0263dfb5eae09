"""The C-ABI library loads without a GPU and exports every symbol include/*.h declares."""
import ctypes
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for header in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = open(header).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(incagg_\w+)\s*\(", text))
    return sorted(names)


def test_header_declares_the_expected_surface():
    names = declared_symbols()
    for must in ("incagg_spmm_csr_f32", "incagg_spmm_csr_incagg_f32", "incagg_gather_rows_f32",
                 "incagg_scatter_rows_f32", "incagg_copy_row_chunks_f32", "incagg_stage_read_async",
                 "incagg_stage_write_async", "incagg_stage_synchronize", "incagg_relabel_one_hop",
                 "incagg_relabel_one_hop_within_batch", "incagg_csr_transpose"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from torch_geometric_autoscale import _lib
    handle = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in declared_symbols() if not hasattr(handle, n)]
    assert not missing, f"declared in include/ but not exported: {missing}"
    # and the ctypes table binds exactly the declared set
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_library_reports_version_and_device_count():
    from torch_geometric_autoscale import _lib
    assert _lib.lib().incagg_abi_version() == 1
    assert _lib.device_count() >= 0
    assert _lib.last_error() is not None


def test_missing_library_fails_loudly(monkeypatch):
    from torch_geometric_autoscale import _lib
    monkeypatch.setattr(_lib, "_handle", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libincagg_gnn.so")
    with pytest.raises(_lib.IncAggLibraryError):
        _lib.lib()


def test_gpu_only_paths_refuse_cpu_tensors():
    import torch
    import torch_geometric_autoscale as tga
    adj = tga.SparseTensor(rowptr=torch.tensor([0, 1]), col=torch.tensor([0]), sparse_sizes=(1, 1))
    with pytest.raises(RuntimeError, match="GPU"):
        adj @ torch.ones(1, 4)
    with pytest.raises(RuntimeError, match="GPU"):
        tga.ops.gather_rows(torch.ones(2, 4), torch.tensor([0]))


def test_torch_ops_are_registered_under_the_reference_names():
    import torch
    import torch_geometric_autoscale  # noqa: F401
    ns = torch.ops.torch_geometric_autoscale
    for name in ("synchronize", "read_async", "write_async", "relabel_one_hop",
                 "relabel_one_hop_within_batch"):
        assert hasattr(ns, name)
