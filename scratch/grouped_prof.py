"""Forward aggregation of one ogbn-products-sized batch in one kernel variant (MODE env), for rocprofv3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "incagg-gnn_amd")); sys.path.insert(0, ROOT)
import torch
import torch_geometric_autoscale as tga
from torch_geometric_autoscale import ops, _lib
from torch_geometric_autoscale.loader import partition_batch
dev = torch.device("cuda")
data, f_in, n_cls = tga.get_data(None, "products", num_parts=150, scale=1.0, features=False,
                                 device=dev, keep_on_device=True)
data.adj_t = tga.gcn_norm(data.adj_t.set_diag()).to("cpu")
sub = partition_batch(tga.Data(adj_t=data.adj_t, num_nodes=data.num_nodes), data.ptr, [0])
adj = sub.data.adj_t.to(dev)
B, M, E = adj.size(0), adj.size(1), adj.nnz()
rp, c, v = adj.device_csr()
x = torch.randn(M, 256, device=dev)
mode = int(os.environ.get("MODE", "2"))
_lib.set_tuning("spmm_sliced", mode)
for k in ("sliced_hub_threshold", "sliced_hub_slice"):
    if os.environ.get(k.upper()):
        _lib.set_tuning(k, int(os.environ[k.upper()]))
deg = (rp[1:] - rp[:-1]).to(torch.int64)
ro = torch.argsort(int(deg.max()) - deg, stable=True).to(torch.int32) if mode == 2 else None
if os.environ.get("TRANSPOSED"):
    rp, c, v, _ = adj.device_csr_t("sum")
    x = torch.randn(B, 256, device=dev)
    M = B
    deg = (rp[1:] - rp[:-1]).to(torch.int64)
    ro = torch.argsort(int(deg.max()) - deg, stable=True).to(torch.int32) if mode == 2 else None
for _ in range(20):
    ops.spmm_csr(rp, c, v, x, M, "sum", max_row_nnz=-1, row_order=ro)
torch.cuda.synchronize()
