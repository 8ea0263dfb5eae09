"""Aggregation launches of the reddit-sized APPNP configuration (200 partitions, 100 per batch,
width 41): batch shape, plan statistics, forward / transposed launch times by plan threshold."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "incagg-gnn_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch_geometric_autoscale as tga  # noqa: E402
from torch_geometric_autoscale import ops, _lib  # noqa: E402
from torch_geometric_autoscale.loader import partition_batch  # noqa: E402

dev = torch.device("cuda")
F = int(os.environ.get("F", "41"))
data, f_in, n_cls = tga.get_data(None, "reddit", num_parts=200, scale=1.0, features=False,
                                 device=dev, keep_on_device=True)
data.adj_t = tga.gcn_norm(data.adj_t.set_diag()).to("cpu")
N = data.num_nodes
sub = partition_batch(tga.Data(adj_t=data.adj_t, num_nodes=N), data.ptr, list(range(100)))
adj = sub.data.adj_t.to(dev)
B, M, E = adj.size(0), adj.size(1), adj.nnz()
rp, c, v = adj.device_csr()
deg = (rp[1:] - rp[:-1]).to(torch.int64)
q = torch.quantile(deg.float(), torch.tensor([0.5, 0.9, 0.99], device=dev)).tolist()
print(f"N={N} B={B} M={M} E={E} F={F} deg median/p90/p99 {q} max {int(deg.max())} "
      f"rows>128 {int((deg > 128).sum())} with {int(deg[deg > 128].sum())} edges", flush=True)
cip = adj.col_in_place(sub.n_id.to(dev)[B:], B)
trp, tc, tv = adj.device_csr_t("sum")
wrp = trp[:B + 1]


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


def rnd(rows):
    return ops.rows_empty(rows, F, dev).normal_()


xb, table, dy, x0 = rnd(B), rnd(N), rnd(B), rnd(B)
xm = rnd(M)
for thr, seg in ((128, 4096), (256, 4096), (512, 4096), (1024, 4096), (128, 1024), (64, 4096),
                 (4096, 8192)):
    plan = ops.SpmmPlan.build(rp, E, thr, seg)
    plan_t = ops.SpmmPlan.build(wrp, E, thr, seg)
    res = [f"fwd in place + mix {timeit(lambda: ops.spmm_csr(rp, cip, v, xb, B + N, 'sum', 0.9, x0, 0.1, x_halo=table, plan=plan)):7.1f}",
           f"fwd one source {timeit(lambda: ops.spmm_csr(rp, c, v, xm, M, plan=plan)):7.1f}",
           f"T [B,B] {timeit(lambda: ops.spmm_csr(wrp, tc, tv, dy, B, plan=plan_t)):7.1f}"]
    print(f"plan {thr}/{seg} [hub {plan.n_hub} items {plan.n_items} multi {plan.n_multi}]: "
          + " | ".join(res), flush=True)
_lib.set_tuning("spmm_variant", 0)
plan = ops.SpmmPlan.build(rp, E)
print("whole-row kernel:", f"{timeit(lambda: ops.spmm_csr(rp, c, v, xm, M, plan=plan)):7.1f}", flush=True)
_lib.set_tuning("spmm_variant", -1)
_lib.check_device_errors()
