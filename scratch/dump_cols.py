"""Writes the column array of one products partition batch (CSR order, and in the plan's
processing order) as raw int32 for scratch/gather_ceiling."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "incagg-gnn_amd"))
import torch  # noqa: E402
import torch_geometric_autoscale as tga  # noqa: E402
from torch_geometric_autoscale.loader import partition_batch  # noqa: E402

dev = torch.device("cuda")
data, f_in, n_cls = tga.get_data(None, "products", num_parts=150, scale=1.0, features=False,
                                 device=dev, keep_on_device=True)
data.adj_t = tga.gcn_norm(data.adj_t.set_diag()).to("cpu")
sub = partition_batch(tga.Data(adj_t=data.adj_t, num_nodes=data.num_nodes), data.ptr, [0])
rowptr, col, _ = sub.data.adj_t.csr()
print("B", sub.batch_size, "M", sub.n_id.numel(), "E", col.numel(), flush=True)
col.to(torch.int32).numpy().tofile(sys.argv[1])
