"""How much of the aggregation launch is row structure: the real products batch against
matrices with the SAME column stream cut into rows of equal length (no long rows, no partly
filled 8-edge chunks)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "incagg-gnn_amd"))
import torch  # noqa: E402
import torch_geometric_autoscale as tga  # noqa: E402
from torch_geometric_autoscale import ops, _lib  # noqa: E402
from torch_geometric_autoscale.loader import partition_batch  # noqa: E402

dev = torch.device("cuda")
F = 256
data, f_in, n_cls = tga.get_data(None, "products", num_parts=150, scale=1.0, features=False,
                                 device=dev, keep_on_device=True)
data.adj_t = tga.gcn_norm(data.adj_t.set_diag()).to("cpu")
sub = partition_batch(tga.Data(adj_t=data.adj_t, num_nodes=data.num_nodes), data.ptr, [0])
adj = sub.data.adj_t.to(dev)
B, M, E = adj.size(0), adj.size(1), adj.nnz()
rp, c, v = adj.device_csr()
x = ops.rows_empty(M, F, dev).normal_()


def timeit(fn, n=40):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


plan = ops.SpmmPlan.build(rp, E)
print(f"real batch B={B} E={E}: {timeit(lambda: ops.spmm_csr(rp, c, v, x, M, plan=plan)):.1f} us "
      f"[hub {plan.n_hub} items {plan.n_items}]", flush=True)
for d in (8, 16, 32, 56, 64, 128, 512, 4096):
    rows = E // d
    rp2 = (torch.arange(rows + 1, device=dev, dtype=torch.int32) * d).contiguous()
    plan2 = ops.SpmmPlan.build(rp2, E)
    t = timeit(lambda: ops.spmm_csr(rp2, c, v, x, M, plan=plan2))
    t1 = timeit(lambda: ops.spmm_csr(rp2, c, None, x, M, plan=plan2))
    ident = ops.SpmmPlan(torch.arange(rows, device=dev, dtype=torch.int32), plan2.items,
                         plan2.multi_rows, plan2.n_hub, plan2.n_items, plan2.n_multi,
                         plan2.hub_threshold, plan2.segment_edges) if plan2.n_hub == 0 else plan2
    t2 = timeit(lambda: ops.spmm_csr(rp2, c, v, x, M, plan=ident))
    print(f"rows of {d:4d} edges ({rows} rows, hub {plan2.n_hub} items {plan2.n_items}): {t:.1f} us  "
          f"{rows * d * 1024 / t / 1e6:.2f} TB/s gathered | no weights {t1:.1f} | identity order {t2:.1f}",
          flush=True)
_lib.check_device_errors()
