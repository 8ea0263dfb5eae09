"""Forward aggregation of a products batch as ONE launch over [fresh rows ; table in place]
against TWO launches: in-batch columns from the fresh rows, then out-of-batch columns from the
table added on top.  Tables rotated (HBM-cold rows), relu on the fresh rows as in the step."""
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
sub = partition_batch(tga.Data(adj_t=data.adj_t, num_nodes=N), data.ptr, [0])
adj = sub.data.adj_t.to(dev)
B, M, E = adj.size(0), adj.size(1), adj.nnz()
rp, c, v = adj.device_csr()
cip = adj.col_in_place(sub.n_id.to(dev)[B:], B)
tables = [ops.rows_empty(N, F, dev).normal_() for _ in range(6)]
xb = torch.randn(B, 2080, device=dev)[:, :F]        # a block of the x_0-term matrix


def split(rp, col, val, keep):
    row = torch.repeat_interleave(torch.arange(B, device=dev), (rp[1:] - rp[:-1]).long())
    row, col2, val2 = row[keep], col[keep], val[keep]
    rp2 = torch.zeros(B + 1, dtype=torch.int32, device=dev)
    rp2[1:] = torch.cumsum(torch.bincount(row, minlength=B), 0).to(torch.int32)
    return rp2, col2.contiguous(), val2.contiguous()


ib = cip < B
rp_i, c_i, v_i = split(rp, cip, v, ib)
rp_o, c_o, v_o = split(rp, cip, v, ~ib)
c_o_tab = (c_o - B).contiguous()                     # table row ids: single-source launch
print(f"B={B} E={E} in-batch edges {int(ib.sum())} out-of-batch {int((~ib).sum())}", flush=True)
plan = ops.SpmmPlan.build(rp, E).with_edges(ops.pack_edges(cip, v))
plan_i = ops.SpmmPlan.build(rp_i, int(ib.sum())).with_edges(ops.pack_edges(c_i, v_i))
plan_o = ops.SpmmPlan.build(rp_o, int((~ib).sum())).with_edges(ops.pack_edges(c_o_tab, v_o))


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


k = [0]


def one():
    k[0] += 1
    return ops.spmm_csr(rp, cip, v, xb, B + N, x_halo=tables[k[0] % 6], plan=plan, relu_x=True)


def first():
    return ops.spmm_csr(rp_i, c_i, v_i, xb, B, plan=plan_i, relu_x=True)


y1 = first()
out = ops.rows_empty(B, F, dev)


def second():
    k[0] += 1
    return ops.spmm_csr(rp_o, c_o_tab, v_o, tables[k[0] % 6], N, add=y1, add_scale=1.0, out=out,
                        plan=plan_o)


def two():
    first()
    return second()


k[0] = 0
ref = one().clone()
k[0] = 0
got = two().clone()
print("max diff", float((ref - got).abs().max()), flush=True)
print(f"one launch {timeit(one):.1f} us | in-batch part {timeit(first):.1f} us | out-of-batch part "
      f"{timeit(second):.1f} us | both {timeit(two):.1f} us", flush=True)
_lib.check_device_errors()

# ---- does it help to have touched the table rows just before (they would then come from the
# Infinity Cache instead of HBM)?  The touch is a gather of the batch's out-of-batch rows.
oob_ids = sub.n_id.to(dev)[B:].contiguous()
scratch_rows = ops.rows_empty(oob_ids.numel(), F, dev)
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(24)]
for label, touch in (("cold", False), ("touched just before", True)):
    for i in range(24):
        k[0] += 1
        tab = tables[k[0] % 6]
        if touch:
            ops.gather_rows(tab, oob_ids, out=scratch_rows)
        ev[i][0].record()
        ops.spmm_csr(rp, cip, v, xb, B + N, x_halo=tab, plan=plan, relu_x=True)
        ev[i][1].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev[4:])
    print(f"one launch, table rows {label}: {ts[len(ts) // 2] * 1e3:.1f} us (event interval)", flush=True)
_lib.check_device_errors()

# ---- the two parts as concurrent launches on two streams (no dependency between them): do
# L2-served gathers and HBM-cold gathers overlap when they come from different waves?
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
outs = [ops.rows_empty(B, F, dev) for _ in range(2)]


def concurrent():
    main = torch.cuda.current_stream()
    sa.wait_stream(main)
    sb.wait_stream(main)
    with torch.cuda.stream(sa):
        ops.spmm_csr(rp_i, c_i, v_i, xb, B, plan=plan_i, relu_x=True, out=outs[0])
    with torch.cuda.stream(sb):
        k[0] += 1
        ops.spmm_csr(rp_o, c_o_tab, v_o, tables[k[0] % 6], N, out=outs[1], plan=plan_o)
    main.wait_stream(sa)
    main.wait_stream(sb)


def serial_pair():
    ops.spmm_csr(rp_i, c_i, v_i, xb, B, plan=plan_i, relu_x=True, out=outs[0])
    k[0] += 1
    ops.spmm_csr(rp_o, c_o_tab, v_o, tables[k[0] % 6], N, out=outs[1], plan=plan_o)


gc, gs = None, None
for name, fn in (("two streams", concurrent), ("one stream", serial_pair)):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(6):
            fn()
    print(f"independent parts, {name}: {timeit(g.replay, 10) / 6:.1f} us per pair", flush=True)
_lib.check_device_errors()
# ---- ONE launch whose rows are the in-batch parts followed by the out-of-batch parts (2B rows):
# every wave then gathers either L2-resident fresh rows or HBM-cold table rows, never both
rp2 = torch.cat([rp_i, rp_i[-1] + rp_o[1:]]).contiguous()
col2 = torch.cat([c_i, c_o]).contiguous()
val2 = torch.cat([v_i, v_o]).contiguous()
plan2 = ops.SpmmPlan.build(rp2, E).with_edges(ops.pack_edges(col2, val2))
out2 = ops.rows_empty(2 * B, F, dev)


def separated():
    k[0] += 1
    return ops.spmm_csr(rp2, col2, val2, xb, B + N, x_halo=tables[k[0] % 6], plan=plan2, relu_x=True,
                        out=out2)


k[0] = 0
got2 = separated()
print("separated rows: max diff", float((ref - (got2[:B] + got2[B:])).abs().max()), flush=True)
print(f"one launch over 2B part-rows: {timeit(separated):.1f} us", flush=True)
_lib.check_device_errors()
if os.environ.get("SPLIT_ONLY"):
    sys.exit(0)

# ---- a chain of 8 layers: aggregation then a [B,256]x[256,256] product, with the out-of-batch
# partial sums either inside each aggregation launch or computed on a side stream up front
W = torch.randn(F, F, device=dev) * 0.05
P = [ops.rows_empty(B, F, dev) for _ in range(8)]
ys = [ops.rows_empty(B, F, dev) for _ in range(8)]
hbuf = torch.empty(B, F, device=dev)
side = torch.cuda.Stream()


def chain_one():
    for l in range(8):
        y = ops.spmm_csr(rp, cip, v, xb, B + N, x_halo=tables[l % 6], plan=plan, relu_x=True,
                         out=ys[l])
        torch.mm(y, W, out=hbuf)
        torch.mm(hbuf, W, out=hbuf2)


hbuf2 = torch.empty(B, F, device=dev)
plans_o = [ops.SpmmPlan.build(rp_o, int((~ib).sum())).with_edges(ops.pack_edges(c_o_tab, v_o))
           for _ in range(8)]


def chain_split():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    evs = []
    with torch.cuda.stream(side):
        for l in range(8):
            ops.spmm_csr(rp_o, c_o_tab, v_o, tables[l % 6], N, out=P[l], plan=plans_o[l])
            e = torch.cuda.Event()
            e.record(side)
            evs.append(e)
    for l in range(8):
        main.wait_event(evs[l])
        y = ops.spmm_csr(rp_i, c_i, v_i, xb, B, plan=plan_i, relu_x=True, add=P[l], add_scale=1.0,
                         out=ys[l])
        torch.mm(y, W, out=hbuf)
        torch.mm(hbuf, W, out=hbuf2)
    main.wait_stream(side)


def graphed(fn):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g.replay


for name, fn in (("one launch per layer", chain_one), ("split, side stream", chain_split)):
    r = graphed(fn)
    print(f"{name}: {timeit(r, 20):.1f} us per 8-layer chain", flush=True)
_lib.check_device_errors()
