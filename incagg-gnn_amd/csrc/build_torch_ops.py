#!/usr/bin/env python3
"""Builds the two operator objects the reference package loads by name
(torch_geometric_autoscale/__init__.py:8-10): ``_relabel.so`` and ``_async.so``, next to this
package's ``__init__``.  Both are thin shells over libincagg_gnn.so (csrc/torch_ops.cpp), built
with the host compiler against the installed torch headers; no GPU is needed to build or load
them.  Usage: python build_torch_ops.py [--force]"""
import os
import subprocess
import sys
import sysconfig
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(os.path.dirname(HERE), "torch_geometric_autoscale")
LIB_DIR = os.path.join(os.path.dirname(HERE), "lib")
SRC = os.path.join(HERE, "torch_ops.cpp")
TARGETS = {"_relabel": "INCAGG_OPS_RELABEL", "_async": "INCAGG_OPS_ASYNC"}


def output(name: str) -> str:
    return os.path.join(PKG, name + ".so")


def _fresh(out: str) -> bool:
    deps = [SRC, os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "incagg_gnn.h")]
    return os.path.isfile(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps)


def build(force: bool = False):
    import torch
    from torch.utils import cpp_extension as ce
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    inc = [f"-I{p}" for p in ce.include_paths()]
    inc += [f"-I{sysconfig.get_paths()['include']}", f"-I{rocm}/include"]
    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)

    def one(item):
        name, macro = item
        out = output(name)
        if not force and _fresh(out):
            return out
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", f"-D{macro}",
               "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", f"-D_GLIBCXX_USE_CXX11_ABI={abi}",
               *inc, SRC, "-o", out, f"-L{torch_lib}", f"-L{LIB_DIR}",
               f"-Wl,-rpath,{torch_lib}", "-Wl,-rpath,$ORIGIN/../lib",
               "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-lincagg_gnn"]
        subprocess.check_call(cmd)
        return out

    with ThreadPoolExecutor(2) as pool:
        return list(pool.map(one, TARGETS.items()))


if __name__ == "__main__":
    print("\n".join(build(force="--force" in sys.argv)))
