"""Device batch builder under a kernel trace: wall time per new batch (relabel, transposed CSR,
plans of the forward and both backward windows) next to the GPU time of its kernels.
Usage: rocprofv3 --kernel-trace --stats ... -- python3 scratch/build_trace.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "incagg-gnn_amd")); sys.path.insert(0, ROOT)
import torch
import torch_geometric_autoscale as tga
from torch_geometric_autoscale.loader import device_partition_batch, _device_graph
dev = torch.device("cuda")
data, f_in, n_cls = tga.get_data(None, "products", num_parts=150, scale=1.0, features=False,
                                 device=dev, keep_on_device=True)
data.adj_t = tga.gcn_norm(data.adj_t.set_diag()).to("cpu")
d = tga.Data(adj_t=data.adj_t, num_nodes=data.num_nodes)
_device_graph(d.adj_t, dev)


def build(p):
    sub = device_partition_batch(d, data.ptr, [p], dev)
    adj = sub.data.adj_t
    b = sub.batch_size
    adj.plan()
    adj.device_csr_t("sum")
    adj.plan_t("sum", 0, b)
    adj.plan_t("sum", 0, adj.size(1))
    return sub


for p in range(3):
    build(p)
torch.cuda.synchronize()
n = 30
for rep in range(4):    # (the first pass still grows torch's block cache: batches differ in size)
    t0 = time.perf_counter()
    for p in range(10 + rep * n, 10 + (rep + 1) * n):
        build(p)
    torch.cuda.synchronize()
    print(f"device builder, pass {rep}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms wall per new batch",
          flush=True)
