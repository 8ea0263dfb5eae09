"""Plan segment length beyond the longest row (no row cut into several segments, no zeroing
launch) against the default, on products partition batches: forward from fresh rows + table in
place (tables rotated so that their rows are HBM-cold) and the transposed [B, B] window."""
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
F = 256
data, f_in, n_cls = tga.get_data(None, "products", num_parts=150, scale=1.0, features=False,
                                 device=dev, keep_on_device=True)
data.adj_t = tga.gcn_norm(data.adj_t.set_diag()).to("cpu")
N = data.num_nodes
tables = [torch.randn(N, F, device=dev) for _ in range(8)]


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


for part in (0, 7, 33):
    sub = partition_batch(tga.Data(adj_t=data.adj_t, num_nodes=N), data.ptr, [part])
    adj = sub.data.adj_t.to(dev)
    B, M, E = adj.size(0), adj.size(1), adj.nnz()
    rp, c, v = adj.device_csr()
    deg = rp[1:] - rp[:-1]
    xb = torch.randn(B, F, device=dev)
    cip = adj.col_in_place(sub.n_id.to(dev)[B:], B)
    trp, tc, tv = adj.device_csr_t("sum")
    wrp = trp[:B + 1]
    tdeg = wrp[1:] - wrp[:-1]
    dy = torch.randn(B, F, device=dev)
    print(f"part {part}: B={B} M={M} E={E} max_deg={int(deg.max())} max_deg_T_window={int(tdeg.max())}",
          flush=True)
    for seg in (1024, 4096, 8192, 16384, 32768, 65536):
        plan = ops.SpmmPlan.build(rp, E, 128, seg)
        plan_t = ops.SpmmPlan.build(wrp, E, 128, seg)
        k = [0]

        def fwd():
            k[0] += 1
            return ops.spmm_csr(rp, cip, v, xb, B + N, x_halo=tables[k[0] % 8], plan=plan)

        def bwd():
            return ops.spmm_csr(wrp, tc, tv, dy, B, plan=plan_t, gate=xb)
        print(f"  seg {seg:6d}: multi fwd {plan.n_multi} T {plan_t.n_multi} | fwd {timeit(fwd):6.1f} us"
              f" | T window gated {timeit(bwd):6.1f} us", flush=True)
_lib.check_device_errors()
