"""Reddit-sized APPNP batch at width 41: row pitch 48 floats (192 B, rows start on alternating
halves of a 128-byte line) against 64 floats (256 B, every row = two aligned lines) and 80."""
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


def rows(r, w, pitch):
    return torch.randn(r, pitch, device=dev)[:, :w]


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
print(f"reddit: B={B} M={M} E={E}", flush=True)
plan = ops.SpmmPlan.build(rp, E).with_edges(ops.pack_edges(c, v))
plan_ip = ops.SpmmPlan.build(rp, E).with_edges(ops.pack_edges(cip, v))
plan_t = ops.SpmmPlan.build(trp, E).with_edges(ops.pack_edges(tc, tv))
plan_w = ops.SpmmPlan.build(trp[:B + 1], E).with_edges(ops.pack_edges(tc, tv))
for F in (41, 47):
    for pitch in (48, 64, 80, 96):
        xm, xb, dy, table = rows(M, F, pitch), rows(B, F, pitch), rows(B, F, pitch), rows(N, F, pitch)
        out_b, out_m = rows(B, F, pitch), rows(M, F, pitch)
        print(f"  width {F} pitch {pitch}: "
              f"fwd {timeit(lambda: ops.spmm_csr(rp, c, v, xm, M, plan=plan, out=out_b)):.1f} "
              f"in place {timeit(lambda: ops.spmm_csr(rp, cip, v, xb, B + N, x_halo=table, plan=plan_ip, out=out_b)):.1f} "
              f"T[M,B] {timeit(lambda: ops.spmm_csr(trp, tc, tv, dy, B, plan=plan_t, out=out_m)):.1f} "
              f"T[B,B] {timeit(lambda: ops.spmm_csr(trp[:B + 1], tc, tv, dy, B, plan=plan_w, out=out_b)):.1f}",
              flush=True)
_lib.check_device_errors()
