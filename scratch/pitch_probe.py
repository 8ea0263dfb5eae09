"""Row pitch of the gathered matrices: 128-byte feature slices pinned to one XCD each only touch
addresses congruent to slice*128 modulo the pitch; with a power-of-two pitch that is a fixed
pattern in the low address bits.  Forward (fresh rows + table in place, tables rotated = HBM
cold) and transposed [B, B] window at several pitches.  Env: F (width)."""
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
F = int(os.environ.get("F", "256"))
data, f_in, n_cls = tga.get_data(None, "products", num_parts=150, scale=1.0, features=False,
                                 device=dev, keep_on_device=True)
data.adj_t = tga.gcn_norm(data.adj_t.set_diag()).to("cpu")
N = data.num_nodes
sub = partition_batch(tga.Data(adj_t=data.adj_t, num_nodes=N), data.ptr, [0])
adj = sub.data.adj_t.to(dev)
B, M, E = adj.size(0), adj.size(1), adj.nnz()
rp, c, v = adj.device_csr()
cip = adj.col_in_place(sub.n_id.to(dev)[B:], B)
trp, tc, tv = adj.device_csr_t("sum")
wrp = trp[:B + 1]
plan = ops.SpmmPlan.build(rp, E)
plan_t = ops.SpmmPlan.build(wrp, E)
plan_tm = ops.SpmmPlan.build(trp, E)


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


def pitched(rows, pitch):
    return torch.randn(rows, pitch, device=dev)[:, :F]


print(f"B={B} M={M} E={E} F={F}", flush=True)
PITCHES = [int(v) for v in os.environ.get("PITCHES", "").split(",") if v] or \
    [F + pad for pad in (0, 32, 64, 96, 160, 8)]
for pitch in PITCHES:
    tables = [pitched(N, pitch) for _ in range(6)]
    xb = pitched(B, pitch)
    xm = pitched(M, pitch)
    dy = pitched(B, pitch)
    gate = pitched(B, pitch)
    k = [0]

    def fwd():
        k[0] += 1
        return ops.spmm_csr(rp, cip, v, xb, B + N, x_halo=tables[k[0] % 6], plan=plan)

    def fwd_pitched_out():
        k[0] += 1
        out = torch.empty(B, pitch, device=dev)[:, :F]
        return ops.spmm_csr(rp, cip, v, xb, B + N, x_halo=tables[k[0] % 6], plan=plan, out=out)
    res = [f"fwd fresh + table {timeit(fwd):6.1f}", f"same, pitched out {timeit(fwd_pitched_out):6.1f}",
           f"fwd one source {timeit(lambda: ops.spmm_csr(rp, c, v, xm, M, plan=plan)):6.1f}",
           f"T [B,B] gated {timeit(lambda: ops.spmm_csr(wrp, tc, tv, dy, B, plan=plan_t, gate=gate)):6.1f}",
           f"T [M,B] {timeit(lambda: ops.spmm_csr(trp, tc, tv, dy, B, plan=plan_tm)):6.1f}"]
    print(f"pitch {pitch:4d} floats: " + " | ".join(res), flush=True)
    del tables, xb, xm, dy, gate
_lib.check_device_errors()
