"""Transposed [M, B] aggregation of a products batch (layer-0 backward: B heavy in-batch rows,
M - B out-of-batch rows with 1-2 edges each, gate rows read, all M rows written): the plan's
order against one that interleaves blocks of heavy and light rows."""
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
N = data.num_nodes
sub = partition_batch(tga.Data(adj_t=data.adj_t, num_nodes=N), data.ptr, [3])
adj = sub.data.adj_t.to(dev)
B, M, E = adj.size(0), adj.size(1), adj.nnz()
trp, tc, tv = adj.device_csr_t("sum")
print(f"B={B} M={M} E={E}", flush=True)


def timeit(fn, n=20):
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


dy = ops.rows_empty(B, F, dev).normal_()
gates = [torch.randn(M, F, device=dev) for _ in range(6)]       # rotated: HBM-cold like in a step
outs = [ops.rows_empty(M, F, dev) for _ in range(6)]
plan = ops.SpmmPlan.build(trp, E).with_edges(ops.pack_edges(tc, tv))
k = [0]


def run(p):
    k[0] += 1
    return ops.spmm_csr(trp, tc, tv, dy, B, plan=p, gate=gates[k[0] % 6], out=outs[k[0] % 6])


def interleaved(plan, rows_per_block=32):
    short = plan.order[plan.n_hub:]
    nb = short.numel() // rows_per_block
    head, tail = short[:nb * rows_per_block].view(nb, rows_per_block), short[nb * rows_per_block:]
    heavy = head[:, 0] < B
    h, l = head[heavy], head[~heavy]
    # position of every block on a common time axis: block i of a class of n sits at (i + .5) / n
    pos = torch.cat([(torch.arange(h.size(0), device=dev) + 0.5) / max(h.size(0), 1),
                     (torch.arange(l.size(0), device=dev) + 0.5) / max(l.size(0), 1)])
    mixed = torch.cat([h, l])[torch.argsort(pos, stable=True)]
    order = torch.cat([plan.order[:plan.n_hub], mixed.reshape(-1), tail]).contiguous()
    twin = ops.SpmmPlan(order, plan.items, plan.multi_rows, plan.n_hub, plan.n_items, plan.n_multi,
                        plan.hub_threshold, plan.segment_edges)
    return twin.with_edges(plan.edges)


ref = run(plan).clone()
p2 = interleaved(plan)
got = run(p2)
print("max diff", float((got - ref).abs().max()), flush=True)
print(f"plan order {timeit(lambda: run(plan)):.1f} us   interleaved {timeit(lambda: run(p2)):.1f} us",
      flush=True)
_lib.set_tuning("plan_local_rows", 1 << 30)
p3 = ops.SpmmPlan.build(trp, E).with_edges(ops.pack_edges(tc, tv))
print(f"by length only {timeit(lambda: run(p3)):.1f} us", flush=True)
_lib.set_tuning("plan_local_rows", 32768)
_lib.check_device_errors()
