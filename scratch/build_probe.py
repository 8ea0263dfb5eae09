"""Batch construction time, host builder vs device builder, ogbn-products-sized graph."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "incagg-gnn_amd")); sys.path.insert(0, ROOT)
import torch
import torch_geometric_autoscale as tga
from torch_geometric_autoscale import ops
from torch_geometric_autoscale.loader import partition_batch, device_partition_batch, _device_graph
dev = torch.device("cuda")
data, f_in, n_cls = tga.get_data(None, "products", num_parts=150, scale=1.0, features=False,
                                 device=dev, keep_on_device=True)
data.adj_t = tga.gcn_norm(data.adj_t.set_diag()).to("cpu")
d = tga.Data(adj_t=data.adj_t, num_nodes=data.num_nodes)
_device_graph(d.adj_t, dev)
for in_batch_only in (False, True):
    # host: relabel + upload + transposed CSR + orders
    t_h = []
    for p in range(3, 9):
        t0 = time.perf_counter()
        sub = partition_batch(d, data.ptr, [p], in_batch_only=in_batch_only)
        t1 = time.perf_counter()
        adj = sub.data.adj_t.to(dev); adj.device_csr_t("sum"); torch.cuda.synchronize()
        t2 = time.perf_counter()
        t_h.append((t1 - t0, t2 - t1))
    t_d = []
    for p in range(3, 9):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sub = device_partition_batch(d, data.ptr, [p], dev, in_batch_only=in_batch_only)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        sub.data.adj_t.device_csr_t("sum"); sub.data.adj_t.plan(); torch.cuda.synchronize()
        t2 = time.perf_counter()
        t_d.append((t1 - t0, t2 - t1))
    med = lambda xs: sorted(xs)[len(xs) // 2] * 1e3
    print(f"in_batch_only={in_batch_only} E={sub.data.adj_t.nnz()}: host relabel {med([a for a, _ in t_h]):.2f} ms"
          f" + upload/transpose/orders {med([b for _, b in t_h]):.2f} ms | device relabel "
          f"{med([a for a, _ in t_d]):.2f} ms + transpose/orders {med([b for _, b in t_d]):.2f} ms", flush=True)
