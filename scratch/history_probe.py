"""Achieved bandwidth of the History-row kernels and of the host staging path (MI355X)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "incagg-gnn_amd")); sys.path.insert(0, ROOT)
import torch
from torch_geometric_autoscale import ops
dev = torch.device("cuda")
N, F, K, B = 2_449_029, 256, 111_738, 16_326

def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3

table = torch.randn(N, F, device=dev)
idx = torch.randperm(N, device=dev)[:K]
rows = torch.randn(K, F, device=dev)
out = torch.empty(K, F, device=dev)
t = timeit(lambda: ops.gather_rows(table, idx, out=out))
print(f"gather_rows   {K} rows x {F}: {t*1e6:7.1f} us  {2*K*F*4/t/1e12:.2f} TB/s (read + write)")
t = timeit(lambda: ops.scatter_rows(table, idx, rows))
print(f"scatter_rows  {K} rows x {F}: {t*1e6:7.1f} us  {2*K*F*4/t/1e12:.2f} TB/s")
off = torch.tensor([1_000_000]); cnt = torch.tensor([B])
xb = torch.randn(B, F, device=dev)
t = timeit(lambda: ops.copy_row_chunks(table, xb, off, cnt, packed_is_src=True))
print(f"push chunk    {B} rows x {F}: {t*1e6:7.1f} us  {2*B*F*4/t/1e12:.2f} TB/s")
t = timeit(lambda: ops.copy_row_chunks(xb, table, off, cnt, packed_is_src=False))
print(f"pull chunk    {B} rows x {F}: {t*1e6:7.1f} us  {2*B*F*4/t/1e12:.2f} TB/s")
big = torch.randn(8 * B, F, device=dev); cnt8 = torch.tensor([8 * B])
t = timeit(lambda: ops.copy_row_chunks(table, big, off, cnt8, packed_is_src=True))
print(f"push chunk    {8*B} rows x {F}: {t*1e6:7.1f} us  {2*8*B*F*4/t/1e12:.2f} TB/s")
a = torch.randn(B, F, device=dev)
t = timeit(lambda: ops.rows_sub(a, xb))
print(f"rows_sub      {B} rows x {F}: {t*1e6:7.1f} us  {3*B*F*4/t/1e12:.2f} TB/s")
big2 = torch.randn(8 * B, F, device=dev)
t = timeit(lambda: ops.rows_sub(big, big2))
print(f"rows_sub      {8*B} rows x {F}: {t*1e6:7.1f} us  {3*8*B*F*4/t/1e12:.2f} TB/s")
t = timeit(lambda: out.copy_(rows))
print(f"torch copy_   {K} rows x {F}: {t*1e6:7.1f} us  {2*K*F*4/t/1e12:.2f} TB/s (device copy yardstick)")
# host-resident table: zero-copy gather / scatter over the host link
host = torch.randn(400_000, F).pin_memory()
hidx = torch.randperm(400_000, device=dev)[:K]
t = timeit(lambda: ops.gather_rows(host, hidx, out=out), n=10)
print(f"zero-copy gather from pinned host {K} rows: {t*1e3:6.2f} ms  {K*F*4/t/1e9:.1f} GB/s")
t = timeit(lambda: ops.scatter_rows(host, hidx, rows), n=10)
print(f"zero-copy scatter to pinned host  {K} rows: {t*1e3:6.2f} ms  {K*F*4/t/1e9:.1f} GB/s")
# staging: chunk read (pinned -> device) and write (device -> pinned) on a side stream
s = torch.cuda.Stream()
buf = torch.empty(K + B, F).pin_memory()
dst = torch.empty(K + B, F, device=dev)
offh, cnth = torch.tensor([100_000]), torch.tensor([B])
hidx_host = hidx.cpu()
import time
def staged_read():
    with torch.cuda.stream(s):
        ops.read_async(host, offh, cnth, hidx_host, dst, buf)
        ops.synchronize()
    s.synchronize()
staged_read()
t0 = time.perf_counter()
for _ in range(5): staged_read()
t = (time.perf_counter() - t0) / 5
print(f"read_async  chunk {B} + gathered {K} rows: {t*1e3:6.2f} ms  {(K+B)*F*4/t/1e9:.1f} GB/s (host gather + copy)")
def staged_write():
    with torch.cuda.stream(s):
        ops.write_async(xb, offh, cnth, host)
    s.synchronize()
staged_write()
t0 = time.perf_counter()
for _ in range(10): staged_write()
t = (time.perf_counter() - t0) / 10
print(f"write_async chunk {B} rows: {t*1e3:6.2f} ms  {B*F*4/t/1e9:.1f} GB/s")
