"""Forward aggregation with out-of-batch rows (a) pulled into a compact buffer first, (b) read in
place from a History-sized table."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "incagg-gnn_amd")); sys.path.insert(0, ROOT)
import torch
import torch_geometric_autoscale as tga
from torch_geometric_autoscale import ops, _lib
from torch_geometric_autoscale.loader import partition_batch
dev = torch.device("cuda")
data, f_in, n_cls = tga.get_data(None, "products", num_parts=150, scale=1.0, features=False,
                                 device=dev, keep_on_device=True)
data.adj_t = tga.gcn_norm(data.adj_t.set_diag()).to("cpu")
sub = partition_batch(tga.Data(adj_t=data.adj_t, num_nodes=data.num_nodes), data.ptr, [3])
adj = sub.data.adj_t.to(dev)
B, M, E = adj.size(0), adj.size(1), adj.nnz()
rp, c, v = adj.device_csr()
N = data.num_nodes
table = torch.randn(N, 256, device=dev)
n_id = sub.n_id.to(dev)
x = torch.randn(B, 256, device=dev)
ro = adj.row_order()
mrn = adj.max_row_nnz()

def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

halo = ops.gather_rows(table, n_id[B:])
t_g = timeit(lambda: ops.gather_rows(table, n_id[B:]))
want = ops.spmm_csr(rp, c, v, x, M, "sum", max_row_nnz=mrn, x_halo=halo, row_order=ro)
t_a = timeit(lambda: ops.spmm_csr(rp, c, v, x, M, "sum", max_row_nnz=mrn, x_halo=halo, row_order=ro))
cip = adj.col_in_place(n_id[B:], B)
t_c = timeit(lambda: (adj.__dict__.pop("_in_place_cols", None), adj.col_in_place(n_id[B:], B)))
got = ops.spmm_csr(rp, cip, v, x, B + N, "sum", max_row_nnz=mrn, x_halo=table, row_order=ro)
t_b = timeit(lambda: ops.spmm_csr(rp, cip, v, x, B + N, "sum", max_row_nnz=mrn, x_halo=table, row_order=ro))
print(f"B={B} M={M} E={E}: gather {t_g:.1f} us + aggregate {t_a:.1f} us = {t_g + t_a:.1f} | in place {t_b:.1f} us "
      f"(equal: {torch.equal(want, got)}) | building the column ids {t_c:.1f} us")
for mode in (0, 1):
    _lib.set_tuning("spmm_sliced", mode)
    t_a = timeit(lambda: ops.spmm_csr(rp, c, v, x, M, "sum", max_row_nnz=mrn, x_halo=halo))
    t_b = timeit(lambda: ops.spmm_csr(rp, cip, v, x, B + N, "sum", max_row_nnz=mrn, x_halo=table))
    print(f"mode {mode}: pulled {t_a:.1f} us, in place {t_b:.1f} us")
