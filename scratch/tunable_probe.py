"""The step's GEMM shapes with hipBLASLt's default choice against torch's TunableOp pick."""
import os
import sys
import time

import torch

dev = torch.device("cuda")
B, M, C, L = 16326, 128064, 256, 8


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


h = torch.randn(B, 288, device=dev)[:, :C]
w = torch.randn(C, C, device=dev)
term = torch.randn(B, 2080, device=dev)[:, :C]
wcat = torch.randn(C, L * C, device=dev)
g = torch.randn(B, 2080, device=dev)[:, :L * C]
x = torch.randn(M, 100, device=dev)
we = torch.randn(C, 100, device=dev)
be = torch.randn(C, device=dev)
gm = torch.randn(M, 288, device=dev)[:, :C]
out_h = torch.empty(B, 288, device=dev)[:, :C]
cases = {
    "update  term += h @ W        ": lambda: term.addmm_(h, w),
    "dH      dy @ W^T             ": lambda: torch.mm(term, w.t(), out=out_h),
    "dW      bmm 8 slabs          ": lambda: torch.bmm(h[:16320].unflatten(0, (8, 2040)).transpose(1, 2),
                                                        term[:16320].unflatten(0, (8, 2040))),
    "x0 terms x0 @ Wcat           ": lambda: torch.mm(h, wcat),
    "d x0    G @ Wcat^T           ": lambda: torch.mm(g, wcat.t()),
    "dWcat   bmm 8 slabs          ": lambda: torch.bmm(h[:16320].unflatten(0, (8, 2040)).transpose(1, 2),
                                                        g[:16320].unflatten(0, (8, 2040))),
    "embed   relu(x We^T + b)     ": lambda: torch._addmm_activation(be, x, we.t(), use_gelu=False),
    "dWe     bmm 32 slabs         ": lambda: torch.bmm(gm[:128064].unflatten(0, (32, 4002)).transpose(1, 2),
                                                        x[:128064].unflatten(0, (32, 4002))),
}
base = {k: timeit(f) for k, f in cases.items()}
torch.cuda.tunable.enable(True)
torch.cuda.tunable.set_max_tuning_duration(30)
torch.cuda.tunable.set_max_tuning_iterations(20)
for k, f in cases.items():
    t0 = time.perf_counter()
    f()
    torch.cuda.synchronize()
    t_tune = time.perf_counter() - t0
    print(f"{k} default {base[k]:7.1f} us   tuned {timeit(f):7.1f} us   (tuning took {t_tune:.1f} s)",
          flush=True)
