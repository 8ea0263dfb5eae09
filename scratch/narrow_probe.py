"""Aggregation launches of the reddit-sized APPNP batch (width 41) and an arxiv-sized GCN batch
(width 256) with the library's own plans: plain descending-length order against the chunk-by-
chunk order (forced through the tuning keys)."""
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


for ds, parts, take, widths in (("reddit", 200, list(range(0, 200, 2)), (41, 64, 128, 256)),
                                ("arxiv", 80, list(range(0, 80, 2)), (40, 256)),
                                ("products", 150, [3], (47, 256))):
    data, f_in, n_cls = tga.get_data(None, ds, num_parts=parts, scale=1.0, features=False,
                                     device=dev, keep_on_device=True)
    data.adj_t = tga.gcn_norm(data.adj_t.set_diag()).to("cpu")
    N = data.num_nodes
    sub = partition_batch(tga.Data(adj_t=data.adj_t, num_nodes=N), data.ptr, take)
    adj = sub.data.adj_t.to(dev)
    B, M, E = adj.size(0), adj.size(1), adj.nnz()
    rp, c, v = adj.device_csr()
    trp, tc, tv = adj.device_csr_t("sum")
    cip = adj.col_in_place(sub.n_id.to(dev)[B:], B)
    print(f"{ds}: B={B} M={M} E={E}", flush=True)
    for F in widths:
        xm, xb, dy = (ops.rows_empty(r, F, dev).normal_() for r in (M, B, B))
        table = ops.rows_empty(N, F, dev).normal_()
        res = []
        for local_rows in (1 << 30, 32768):
            _lib.set_tuning("plan_local_rows", local_rows)
            pk = ops.pack_edges(c, v)
            plan = ops.SpmmPlan.build(rp, E).with_edges(pk)
            plan_ip = ops.SpmmPlan.build(rp, E).with_edges(ops.pack_edges(cip, v))
            plan_t = ops.SpmmPlan.build(trp, E).with_edges(ops.pack_edges(tc, tv))
            plan_w = ops.SpmmPlan.build(trp[:B + 1], E).with_edges(ops.pack_edges(tc, tv))
            res.append(f"[local>{local_rows}: fwd {timeit(lambda: ops.spmm_csr(rp, c, v, xm, M, plan=plan)):.1f} "
                       f"in place {timeit(lambda: ops.spmm_csr(rp, cip, v, xb, B + N, x_halo=table, plan=plan_ip)):.1f} "
                       f"T[M,B] {timeit(lambda: ops.spmm_csr(trp, tc, tv, dy, B, plan=plan_t)):.1f} "
                       f"T[B,B] {timeit(lambda: ops.spmm_csr(trp[:B + 1], tc, tv, dy, B, plan=plan_w)):.1f}]")
        print(f"  width {F}: " + " ".join(res), flush=True)
    del data, adj, table
    torch.cuda.empty_cache()
_lib.set_tuning("plan_local_rows", 32768)
_lib.check_device_errors()
