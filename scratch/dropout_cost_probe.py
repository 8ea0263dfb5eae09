"""Per-step cost of random edge dropout (utils.py:62-73, applied at every training step,
main.py:68) and of per-row neighbour sampling (loader.py:217-246) on one ogbn-products-sized
partition batch: GPU time of the preparation alone, and of a whole eagerly issued push/pull
step with and without it.  Usage: python3 scratch/dropout_cost_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "incagg-gnn_amd"))
import torch  # noqa: E402
import torch_geometric_autoscale as tga  # noqa: E402
from torch_geometric_autoscale.loader import device_partition_batch, sample_neighbors  # noqa: E402
from torch_geometric_autoscale.models import GCN2  # noqa: E402
from torch_geometric_autoscale.utils import dropout  # noqa: E402

dev = torch.device("cuda", 0)
data, f_in, n_cls = tga.get_data(None, "products", num_parts=150, device=dev, keep_on_device=True)
ptr = data.ptr
data.adj_t = tga.gcn_norm(data.adj_t.set_diag()).to("cpu")
ddata = tga.Data(**{k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in data if k != "ptr"})
sub = device_partition_batch(ddata, ptr, [7], dev)
adj, b = sub.data.adj_t, sub.batch_size
print("batch rows", b, "with halo", sub.n_id.numel(), "edges", adj.nnz(), flush=True)
model = GCN2(data.num_nodes, f_in, 256, n_cls, 8, alpha=0.1, theta=0.5, shared_weights=False,
             dropout=0.0, device=dev).to(dev)
for h in model.histories:
    h.emb.normal_(0, 0.1)
opt = torch.optim.SGD(model.parameters(), lr=1e-3)
x, y = sub.data.x, sub.data.y[:b]


def step(a, n_id):
    out = model(x, a, b, n_id, sub.offset, sub.count)["out"]
    opt.zero_grad(set_to_none=True)
    torch.nn.functional.cross_entropy(out, y).backward()
    opt.step()


def gpu_ms(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps, (time.perf_counter() - t0) / reps * 1e3


def prepared(a):
    """Everything a step derives from the adjacency (forward + backward edge lists, plans)."""
    a.plan_edges_in_place(sub.n_id[b:], b)
    a.plan_t("sum", 0, b)
    a.plan_t("sum", 0, a.size(1))
    return a


print("step, plain                 gpu %.3f ms  wall %.3f ms" % gpu_ms(lambda: step(adj, sub.n_id)))
print("edge dropout, preparation   gpu %.3f ms  wall %.3f ms" % gpu_ms(lambda: prepared(dropout(adj, 0.2))))
print("step with edge dropout      gpu %.3f ms  wall %.3f ms" % gpu_ms(lambda: step(dropout(adj, 0.2), sub.n_id)))
print("sampling 10/row, preparation gpu %.3f ms  wall %.3f ms" % gpu_ms(
    lambda: prepared(sample_neighbors(adj, 10, sub.n_id)[0])))
print("step with sampling 10/row   gpu %.3f ms  wall %.3f ms" % gpu_ms(
    lambda: step(sample_neighbors(adj, 10, sub.n_id)[0], sub.n_id)))
tga._lib.check_device_errors()
