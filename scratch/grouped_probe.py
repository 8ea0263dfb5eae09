"""A/B of the aggregation kernel variants (rows / sliced / grouped) on one
ogbn-products-sized partition batch: forward [B, M], transposed [M, B], split source."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "incagg-gnn_amd")); sys.path.insert(0, ROOT)
import torch
import torch_geometric_autoscale as tga
from torch_geometric_autoscale import ops, _lib
from torch_geometric_autoscale.loader import partition_batch

dev = torch.device("cuda")
scale = float(os.environ.get("SCALE", "1.0"))
data, f_in, n_cls = tga.get_data(None, "products", num_parts=150, scale=scale, features=False,
                                 device=dev, keep_on_device=True)
data.adj_t = tga.gcn_norm(data.adj_t.set_diag()).to("cpu")
sub = partition_batch(tga.Data(adj_t=data.adj_t, num_nodes=data.num_nodes), data.ptr,
                      [int(os.environ.get("PART", "0"))])
adj = sub.data.adj_t.to(dev)
F = int(os.environ.get("F", "256"))
B, M, E = adj.size(0), adj.size(1), adj.nnz()
print(f"B={B} M={M} E={E} F={F}", flush=True)


def order_of(rp, block):
    deg = (rp[1:] - rp[:-1]).to(torch.int64)
    n = deg.numel()
    if block <= 0:
        return None
    key = (torch.arange(n, device=rp.device) // block) * (1 << 32) + (int(deg.max()) - deg)
    return torch.argsort(key, stable=True).to(torch.int32)


def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def ab(name, rp, c, v, x, n_cols, mrn, x_halo=None):
    _lib.set_tuning("spmm_sliced", 0)
    want = ops.spmm_csr(rp, c, v, x, n_cols, "sum", max_row_nnz=mrn, x_halo=x_halo)
    nnz = c.numel()
    res = []
    for mode in (0, 1):
        _lib.set_tuning("spmm_sliced", mode)
        t = timeit(lambda: ops.spmm_csr(rp, c, v, x, n_cols, "sum", max_row_nnz=mrn, x_halo=x_halo))
        res.append(f"mode{mode} {t:.1f}")
    _lib.set_tuning("spmm_sliced", 2)
    for block in (0, 256, 1024, 4096, 1 << 30):
        ro = order_of(rp, block)
        got = ops.spmm_csr(rp, c, v, x, n_cols, "sum", max_row_nnz=mrn, x_halo=x_halo, row_order=ro)
        err = float((got - want).abs().max())
        t = timeit(lambda: ops.spmm_csr(rp, c, v, x, n_cols, "sum", max_row_nnz=mrn, x_halo=x_halo,
                                        row_order=ro))
        res.append(f"grouped/blk{block if block < (1 << 30) else 'all'} {t:.1f} (err {err:.1e})")
    print(f"{name}: nnz={nnz} rows={rp.numel() - 1} | " + " | ".join(res), flush=True)
    _lib.set_tuning("spmm_sliced", -1)


rp, c, v = adj.device_csr()
x = torch.randn(M, F, device=dev)
ab("forward [B,M]", rp, c, v, x, M, adj.max_row_nnz())
ab("forward split", rp, c, v, x[:B].contiguous(), M, adj.max_row_nnz(), x_halo=x[B:].contiguous())
trp, tc, tv, _tmax = adj.device_csr_t("sum")
dy = torch.randn(B, F, device=dev)
ab("transposed [M,B]", trp, tc, tv, dy, B, -1)
ab("transposed [B,B] window", trp[:B + 1], tc, tv, dy, B, -1)
def grouped_only(name, rp, c, v, x, n_cols, mrn):
    ro = order_of(rp, 1 << 30)
    _lib.set_tuning("spmm_sliced", 2)
    t = timeit(lambda: ops.spmm_csr(rp, c, v, x, n_cols, "sum", max_row_nnz=mrn, row_order=ro))
    _lib.set_tuning("spmm_sliced", -1)
    return f"{name} {t:.1f}"


for thr in (24, 32, 48, 64, 96, 128):
    for sl in (64, 128):
        _lib.set_tuning("sliced_hub_threshold", thr)
        _lib.set_tuning("sliced_hub_slice", sl)
        print(f"thr {thr} slice {sl}: ", grouped_only("fwd", rp, c, v, x, M, adj.max_row_nnz()),
              grouped_only("T[M,B]", trp, tc, tv, dy, B, -1),
              grouped_only("T[B,B]", trp[:B + 1], tc, tv, dy, B, -1), flush=True)
_lib.check_device_errors()
