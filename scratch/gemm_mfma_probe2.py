"""ops.rows_gemm against K: the slope is the cost of a 32-deep chunk, the intercept prologue +
epilogue; cold (caches flushed) and warm (same operands again)."""
import sys
import torch
sys.path.insert(0, "incagg-gnn_amd")
from torch_geometric_autoscale import ops

dev = torch.device("cuda", 0)
B, C = 16326, 256
flush = torch.empty(128 * 1024 * 1024, device=dev)
small = torch.empty(8 * 1024 * 1024, device=dev)


def timed(fn, cold, reps=20):
    ts = []
    for _ in range(reps):
        if cold:
            flush.zero_()
        else:       # (something small in front, so that the launch does not start from idle)
            torch.cuda._sleep(300000)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


empty = lambda: None
print("empty interval: cold %.1f warm %.1f us" % (timed(empty, True), timed(empty, False)))
for k in (32, 64, 128, 256):
    x = ops.rows_empty(B, k, dev).normal_()
    w = torch.randn(C, k, device=dev) * 0.05
    out = ops.rows_empty(B, C, dev)
    f = lambda: ops.rows_gemm(x, w, True, out=out)
    g = lambda: torch.mm(x, w.t(), out=out)
    print("k %3d  rows_gemm NT cold %.1f warm %.1f | torch cold %.1f warm %.1f us" % (
        k, timed(f, True), timed(f, False), timed(g, True), timed(g, False)))
