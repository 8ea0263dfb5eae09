"""Reddit-sized batch at width 41: launch time by the plan's chunk size (rows per chunk)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "incagg-gnn_amd"))
import torch  # noqa: E402
import torch_geometric_autoscale as tga  # noqa: E402
from torch_geometric_autoscale import ops, _lib  # noqa: E402
from torch_geometric_autoscale.loader import partition_batch  # noqa: E402

dev = torch.device("cuda")


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


data, f_in, n_cls = tga.get_data(None, "reddit", num_parts=200, scale=1.0, features=False,
                                 device=dev, keep_on_device=True)
data.adj_t = tga.gcn_norm(data.adj_t.set_diag()).to("cpu")
N = data.num_nodes
sub = partition_batch(tga.Data(adj_t=data.adj_t, num_nodes=N), data.ptr, list(range(100)))
adj = sub.data.adj_t.to(dev)
B, M, E = adj.size(0), adj.size(1), adj.nnz()
rp, c, v = adj.device_csr()
trp, tc, tv = adj.device_csr_t("sum")
cip = adj.col_in_place(sub.n_id.to(dev)[B:], B)
F = 41
xb, dy = (ops.rows_empty(r, F, dev).normal_() for r in (B, B))
table = ops.rows_empty(N, F, dev).normal_()
print(f"B={B} M={M} E={E}", flush=True)
for chunk in (256, 512, 1024, 2048, 4096):
    for thr, seg in ((128, 4096), (128, 1024), (256, 1024)):
        _lib.set_tuning("plan_chunk_rows", chunk)
        plan_ip = ops.SpmmPlan.build(rp, E, thr, seg).with_edges(ops.pack_edges(cip, v))
        plan_w = ops.SpmmPlan.build(trp[:B + 1], E, thr, seg).with_edges(ops.pack_edges(tc, tv))
        print(f"chunk {chunk} plan {thr}/{seg}: in place "
              f"{timeit(lambda: ops.spmm_csr(rp, cip, v, xb, B + N, x_halo=table, plan=plan_ip)):.1f} "
              f"T[B,B] {timeit(lambda: ops.spmm_csr(trp[:B + 1], tc, tv, dy, B, plan=plan_w)):.1f}",
              flush=True)
_lib.set_tuning("plan_chunk_rows", 1024)
_lib.check_device_errors()
