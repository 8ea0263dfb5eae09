"""Isolated timing of the forward aggregation on one ogbn-products-sized partition batch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "incagg-gnn_amd")); sys.path.insert(0, ROOT)
import torch
import torch_geometric_autoscale as tga
from torch_geometric_autoscale import ops
from torch_geometric_autoscale.loader import partition_batch
dev = torch.device("cuda")
scale = float(os.environ.get("SCALE", "1.0"))
data, f_in, n_cls = tga.get_data(None, "products", num_parts=150, scale=scale, features=False, device=dev, keep_on_device=True)
data.adj_t = tga.gcn_norm(data.adj_t.set_diag()).to("cpu")
sub = partition_batch(tga.Data(adj_t=data.adj_t, num_nodes=data.num_nodes), data.ptr, [int(os.environ.get("PART", "0"))])
adj = sub.data.adj_t.to(dev)
rp, c, v = adj.device_csr()
B, M, E, F = adj.size(0), adj.size(1), adj.nnz(), int(os.environ.get("F", "256"))
deg = (rp[1:] - rp[:-1]).cpu()
print(f"B={B} M={M} E={E} F={F} maxdeg={int(deg.max())} rows>128: {(deg>128).sum().item()} edges in rows>128: {deg[deg>128].sum().item()}")
PAD = int(os.environ.get("PAD", "0"))
x = torch.randn(M, F + PAD, device=dev)[:, :F]
def run(n=30, **kw):
    for _ in range(3): ops.spmm_csr(rp, c, v, x, M, "sum", max_row_nnz=adj.max_row_nnz(), **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): ops.spmm_csr(rp, c, v, x, M, "sum", max_row_nnz=adj.max_row_nnz(), **kw)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
t = run()
gather = E * (F * 4 + 8) + B * F * 4
algo = E * 8 + (B + 1) * 4 + M * F * 4 + B * F * 4
print(f"spmm pair: {t:.1f} us  gather {gather/t/1e6:.2f} TB/s  compulsory {algo/t/1e6:.2f} TB/s")
# split source variant
t2 = run(x_halo=x[B:].contiguous()) if False else None
# gather kernel alone
idx = sub.n_id[B:].to(dev)
table = torch.randn(data.num_nodes, F, device=dev)
for _ in range(3): ops.gather_rows(table, idx)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(30): ops.gather_rows(table, idx)
e.record(); torch.cuda.synchronize()
tg = s.elapsed_time(e) / 30 * 1e3
print(f"gather {idx.numel()} rows: {tg:.1f} us  {(idx.numel()*F*8)/tg/1e6:.2f} TB/s")
