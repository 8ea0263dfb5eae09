"""Narrow aggregation (width 41: two 128-byte slices) of a reddit-sized batch: the plan's row
order (descending length) against an order that keeps the rows an XCD pair processes within one
quarter of the batch (rows of a quarter only in workgroups whose index is congruent mod 4), so
that the sources a private L2 sees are mostly that quarter's rows."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "incagg-gnn_amd"))
import torch  # noqa: E402
import torch_geometric_autoscale as tga  # noqa: E402
from torch_geometric_autoscale import ops, _lib  # noqa: E402
from torch_geometric_autoscale.loader import partition_batch  # noqa: E402

dev = torch.device("cuda")
F = int(os.environ.get("F", "41"))
DS = os.environ.get("DS", "reddit")
PARTS = 200 if DS == "reddit" else 150
data, f_in, n_cls = tga.get_data(None, DS, num_parts=PARTS, scale=1.0, features=False,
                                 device=dev, keep_on_device=True)
data.adj_t = tga.gcn_norm(data.adj_t.set_diag()).to("cpu")
N = data.num_nodes
sub = partition_batch(tga.Data(adj_t=data.adj_t, num_nodes=N), data.ptr, list(range(0, 200, 2)) if DS == 'reddit' else [3])
adj = sub.data.adj_t.to(dev)
B, M, E = adj.size(0), adj.size(1), adj.nnz()
rp, c, v = adj.device_csr()
print(f"B={B} M={M} E={E} F={F}", flush=True)


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


def bucketed(plan, n_buckets, do_items=False, sub=1, rows_per_block=32):
    short = plan.order[plan.n_hub:].long()
    bucket = short * n_buckets // plan.n_rows
    lists = [short[bucket == k] for k in range(n_buckets)]
    if sub > 1:     # inside a bucket: chunk of rows first, then descending length
        lists = [l[torch.argsort(l * (n_buckets * sub) // plan.n_rows, stable=True)] for l in lists]
    full = min(int(l.numel()) // rows_per_block for l in lists)
    # block j of the short rows is workgroup n_items + j: its bucket is (n_items + j) % n_buckets
    first = [(k - plan.n_items) % n_buckets for k in range(n_buckets)]  # block slot -> bucket
    head = torch.stack([lists[(plan.n_items + s) % n_buckets][:full * rows_per_block]
                        .view(full, rows_per_block) for s in range(n_buckets)], 1)
    rest = torch.cat([l[full * rows_per_block:] for l in lists])
    deg = (rp[1:] - rp[:-1]).long()
    rest = rest[torch.argsort(deg[rest], descending=True, stable=True)]
    order = torch.cat([plan.order[:plan.n_hub].long(), head.reshape(-1), rest]).to(torch.int32)
    assert order.numel() == plan.n_rows and int(torch.unique(order).numel()) == plan.n_rows
    items = plan.items
    if do_items and plan.n_items >= n_buckets:
        it = plan.items[:plan.n_items]
        ib = it[:, 0].long() * n_buckets // plan.n_rows
        ilists = [it[ib == k] for k in range(n_buckets)]
        if sub > 1:
            ilists = [l[torch.argsort(l[:, 0].long() * (n_buckets * sub) // plan.n_rows, stable=True)]
                      for l in ilists]
        ifull = min(int(l.size(0)) for l in ilists)
        ihead = torch.stack([l[:ifull] for l in ilists], 1).reshape(-1, 2)
        irest = torch.cat([l[ifull:] for l in ilists])
        items = torch.cat([ihead, irest]).contiguous()
        assert items.size(0) == plan.n_items
    twin = ops.SpmmPlan(order, items, plan.multi_rows, plan.n_hub, plan.n_items,
                        plan.n_multi, plan.hub_threshold, plan.segment_edges)
    return twin.with_edges(plan.edges)


xm = ops.rows_empty(M, F, dev).normal_()
plan = ops.SpmmPlan.build(rp, E).with_edges(ops.pack_edges(c, v))
print(f"plan: hub {plan.n_hub} items {plan.n_items} multi {plan.n_multi}", flush=True)
ref = ops.spmm_csr(rp, c, v, xm, M, plan=plan).clone()
print(f"plan order: {timeit(lambda: ops.spmm_csr(rp, c, v, xm, M, plan=plan)):.1f} us", flush=True)
for nb, di, sub in [tuple(int(z) for z in t.split(',')) for t in os.environ.get('CASES', '4,1,25 1,1,100').split()]:
    p2 = bucketed(plan, nb, di, sub)
    got = ops.spmm_csr(rp, c, v, xm, M, plan=p2)
    print(f"{nb} buckets items {di} sub {sub}: max diff {float((got - ref).abs().max()):.2e}  "
          f"{timeit(lambda: ops.spmm_csr(rp, c, v, xm, M, plan=p2)):.1f} us", flush=True)
_lib.check_device_errors()
