"""Recipe: build the reference's OWN relabel operators into ``oracle/_ref/``.

TEST INFRASTRUCTURE ONLY.  Nothing shipped imports this module; only ``tests/``,
``__graft_entry__`` and the fixture generator use what it builds.

The two translation units are compiled from where they lie under
``/root/reference`` (``csrc/relabel.cpp`` and ``csrc/cpu/relabel_cpu.cpp``); no
reference source is copied into this repository and only the resulting shared
object lands in ``oracle/_ref/`` (git-ignored).  The object registers
``torch.ops.torch_geometric_autoscale.relabel_one_hop`` and
``...relabel_one_hop_within_batch`` exactly as ``csrc/relabel.cpp:40-42`` does.

The reference's async operators are NOT built: without a GPU toolchain pass
``csrc/async.cpp`` compiles to stubs that always raise (``csrc/async.cpp:13-41``)
and ``csrc/cuda/async_cuda.cu`` is CUDA host code.
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = os.environ.get("INCAGG_REFERENCE_ROOT", "/root/reference")
OUT_DIR = os.path.join(HERE, "_ref")
OUT_SO = os.path.join(OUT_DIR, "ref_relabel.so")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "csrc", "relabel.cpp"))


def build(force: bool = False) -> str:
    """Compile the reference relabel sources; returns the path of the .so.  Where the
    reference tree is absent (the GPU box) an object built earlier is used as it is."""
    if not reference_available():
        if os.path.isfile(OUT_SO):
            return OUT_SO
        raise FileNotFoundError(f"reference sources not found under {REF_ROOT}")
    srcs = [os.path.join(REF_ROOT, "csrc", "relabel.cpp"),
            os.path.join(REF_ROOT, "csrc", "cpu", "relabel_cpu.cpp")]
    if (not force and os.path.isfile(OUT_SO)
            and all(os.path.getmtime(OUT_SO) >= os.path.getmtime(s) for s in srcs)):
        return OUT_SO
    import torch
    from torch.utils import cpp_extension as ce
    os.makedirs(OUT_DIR, exist_ok=True)
    inc = [f"-I{p}" for p in ce.include_paths()]
    inc += [f"-I{sysconfig.get_paths()['include']}", f"-I{os.path.join(REF_ROOT, 'csrc')}"]
    libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    cmd = (["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w",
            f"-D_GLIBCXX_USE_CXX11_ABI={abi}"] + inc + srcs +
           ["-o", OUT_SO, f"-L{libdir}", f"-Wl,-rpath,{libdir}",
            "-ltorch", "-ltorch_cpu", "-lc10"])
    subprocess.check_call(cmd)
    return OUT_SO


_loaded = False


def load():
    """Load the built object (registering the reference ops) and return the two ops.

    NOTE: the shipped package registers operators under the same namespace, so a
    process may load EITHER this object OR the package, never both.  The fixture
    generator runs it in a process of its own.
    """
    global _loaded
    import torch
    if not _loaded:
        torch.ops.load_library(build())
        _loaded = True
    ns = torch.ops.torch_geometric_autoscale
    return ns.relabel_one_hop, ns.relabel_one_hop_within_batch


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
