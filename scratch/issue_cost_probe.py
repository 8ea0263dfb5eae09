"""Host time to ISSUE the embedding of a batch (x W^T + b, relu) as one fused hipBLASLt call
against addmm + the relu row kernel; the GPU is kept far behind by a spin kernel so that
nothing blocks on it."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "incagg-gnn_amd"))
import torch  # noqa: E402
from torch_geometric_autoscale import ops  # noqa: E402

dev = torch.device("cuda")
for rows in (16326, 128064):
    x = torch.randn(rows, 100, device=dev)
    w = torch.randn(256, 100, device=dev)
    b = torch.randn(256, device=dev)
    cases = {
        "_addmm_activation": lambda: torch._addmm_activation(b, x, w.t(), use_gelu=False),
        "addmm + relu_rows_": lambda: ops.relu_rows_(torch.addmm(b, x, w.t())),
        "addmm": lambda: torch.addmm(b, x, w.t()),
        "mm": lambda: torch.mm(x, w.t()),
    }
    for name, fn in cases.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        n = 50
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        t_issue = (time.perf_counter() - t0) / n * 1e6
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        print(f"rows {rows:6d} {name:20s} issue {t_issue:7.1f} us   gpu {s.elapsed_time(e) / n * 1e3:7.1f} us", flush=True)
