"""Times ops.rows_gemm (hand-written fp32 MFMA product, optional fused push) against torch.mm /
addmm_ + the chunk push it replaces, at the shapes of a GCNII layer of the headline step; caches
flushed between launches.  Usage: python3 scratch/gemm_mfma_probe.py"""
import sys
import torch
sys.path.insert(0, "incagg-gnn_amd")
from torch_geometric_autoscale import ops

dev = torch.device("cuda", 0)
B, C = 16326, 256
x = ops.rows_empty(B, C, dev).normal_()
w = torch.randn(C, C, device=dev) * 0.05
term = ops.rows_empty(B, C, dev).normal_()
table = ops.rows_empty(200000, C, dev).zero_()
flush = torch.empty(128 * 1024 * 1024, device=dev)
off, cnt = torch.tensor([5000]), torch.tensor([B])


def timed(fn, reps=20):
    ts = []
    for _ in range(reps):
        flush.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


out = ops.rows_empty(B, C, dev)
print("torch.mm NN            %.1f us" % timed(lambda: torch.mm(x, w, out=out)))
print("rows_gemm NN           %.1f us" % timed(lambda: ops.rows_gemm(x, w, False, out=out)))
print("torch.mm NT            %.1f us" % timed(lambda: torch.mm(x, w.t(), out=out)))
print("rows_gemm NT           %.1f us" % timed(lambda: ops.rows_gemm(x, w, True, out=out)))
print("addmm_ + push          %.1f us" % timed(lambda: (term.addmm_(x, w), ops.copy_row_chunks(
    table, term, off, cnt, packed_is_src=True, device=dev, relu=True))))
print("rows_gemm acc + push   %.1f us" % timed(lambda: ops.rows_gemm(
    x, w, False, out=term, accumulate=True, table=table, table_row0=5000, relu_push=True)))
