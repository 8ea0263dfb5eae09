"""Aggregation kernels on one ogbn-products-sized partition batch: forward [B, M] from one
source matrix, from fresh + pulled rows, from fresh rows + the History table in place; the
transposed [M, B] and [B, B] launches of the backward pass.  Sweeps the plan's threshold and
segment length.  Env: SCALE, PART, F, SWEEP=0/1, REPS."""
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
scale = float(os.environ.get("SCALE", "1.0"))
F = int(os.environ.get("F", "256"))
REPS = int(os.environ.get("REPS", "30"))
data, f_in, n_cls = tga.get_data(None, "products", num_parts=150, scale=scale, features=False,
                                 device=dev, keep_on_device=True)
data.adj_t = tga.gcn_norm(data.adj_t.set_diag()).to("cpu")
N = data.num_nodes
sub = partition_batch(tga.Data(adj_t=data.adj_t, num_nodes=N), data.ptr,
                      [int(os.environ.get("PART", "0"))])
adj = sub.data.adj_t.to(dev)
B, M, E = adj.size(0), adj.size(1), adj.nnz()
rp, c, v = adj.device_csr()
deg = (rp[1:] - rp[:-1])
print(f"B={B} M={M} E={E} F={F} N={N} max_deg={int(deg.max())} "
      f"rows>128: {int((deg > 128).sum())} edges in them: {int(deg[deg > 128].sum())} "
      f"rows>1024: {int((deg > 1024).sum())}", flush=True)


def timeit(fn, n=REPS):
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


def rnd(rows):   # row pitch as the package allocates gathered matrices (ops.row_pitch)
    return ops.rows_empty(rows, F, dev).normal_()


x = rnd(M)
xb = rnd(B).copy_(x[:B])
halo = rnd(M - B).copy_(x[B:])
table = rnd(N)
n_id = sub.n_id.to(dev)
cip = adj.col_in_place(n_id[B:], B)
trp, tc, tv = adj.device_csr_t("sum")
dy = rnd(B)

shapes = {
    "fwd [B,M] one source": lambda plan: ops.spmm_csr(rp, c, v, x, M, plan=plan),
    "fwd fresh + pulled": lambda plan: ops.spmm_csr(rp, c, v, xb, M, x_halo=halo, plan=plan),
    "fwd fresh + table in place": lambda plan: ops.spmm_csr(rp, cip, v, xb, B + N, x_halo=table,
                                                            plan=plan),
}
t_shapes = {
    "T [M,B]": (trp, M),
    "T [B,B] window": (trp[:B + 1], B),
}

# reference results from the whole-row kernel without a plan
_lib.set_tuning("spmm_variant", 0)
want = {k: fn(None).clone() for k, fn in shapes.items()}
want_t = {k: ops.spmm_csr(r, tc, tv, dy, B).clone() for k, (r, _) in t_shapes.items()}
_lib.set_tuning("spmm_variant", -1)


def run(label, variant, thr, seg):
    _lib.set_tuning("spmm_variant", variant)
    plan = None if thr is None else ops.SpmmPlan.build(rp, E, thr, seg)
    res = []
    for k, fn in shapes.items():
        err = float((fn(plan) - want[k]).abs().max())
        res.append(f"{k} {timeit(lambda: fn(plan)):.1f} us (err {err:.1e})")
    for k, (r, rows) in t_shapes.items():
        p = None if thr is None else ops.SpmmPlan.build(r, E, thr, seg)
        err = float((ops.spmm_csr(r, tc, tv, dy, B, plan=p) - want_t[k]).abs().max())
        res.append(f"{k} {timeit(lambda: ops.spmm_csr(r, tc, tv, dy, B, plan=p)):.1f} us (err {err:.1e})")
    info = "" if plan is None else f" [hub {plan.n_hub} items {plan.n_items} multi {plan.n_multi}]"
    print(f"{label}{info}: " + " | ".join(res), flush=True)
    _lib.set_tuning("spmm_variant", -1)


run("rows, no plan", 0, None, None)
run("rows, plan 128/1024", 0, 128, 1024)
run("slices, no plan", 1, None, None)
run("slices, plan 128/1024", 1, 128, 1024)
if os.environ.get("SWEEP", "1") == "1":
    for thr in (32, 64, 128, 256):
        for seg in (256, 512, 1024, 2048, 4096):
            run(f"slices, plan {thr}/{seg}", 1, thr, seg)
_lib.check_device_errors()
print("done", flush=True)

# ---- out-of-batch rows HBM-cold (as inside a training step: every layer reads another table),
# with the edges of every row in storage order vs in-batch columns first
if os.environ.get("COLD", "1") == "1":
    tables = [rnd(N) for _ in range(8)]
    row = torch.repeat_interleave(torch.arange(B, device=dev), deg.to(torch.int64))
    key = row * 2 + (c >= B).to(torch.int64)
    perm = torch.argsort(key, stable=True)
    c2, v2, cip2 = c[perm].contiguous(), v[perm].contiguous(), cip[perm].contiguous()
    plan = ops.SpmmPlan.build(rp, E)
    for label, cc, vv in (("storage order", cip, v), ("in-batch columns first", cip2, v2)):
        k = [0]

        def call():
            k[0] += 1
            return ops.spmm_csr(rp, cc, vv, xb, B + N, x_halo=tables[k[0] % 8], plan=plan)
        ref = ops.spmm_csr(rp, cip, v, xb, B + N, x_halo=tables[1], plan=plan)
        k[0] = 0
        err = float((call() - ref).abs().max())
        print(f"cold tables, {label}: {timeit(call):.1f} us (err {err:.1e})", flush=True)
    for label, cc, vv in (("storage order", c, v), ("in-batch columns first", c2, v2)):
        print(f"warm one source, {label}: {timeit(lambda: ops.spmm_csr(rp, cc, vv, x, M, plan=plan)):.1f} us",
              flush=True)
