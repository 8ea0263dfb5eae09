"""The BASELINE.json configurations other than the headline one, through the training driver at full
(synthetic) size on one MI355X: GCN / ogbn-arxiv-sized, APPNP / reddit-sized, GraphSAGE /
ogbn-products-sized with host-pinned History tables behind the staging pools."""
import logging, os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "incagg-gnn_amd")); sys.path.insert(0, ROOT)
logging.basicConfig(level=logging.INFO, format="%(message)s")
from torch_geometric_autoscale import train, _lib
which = sys.argv[1]
common = {"max_steps": -1, "num_workers": 0, "reg_weight_decay": 0, "nonreg_weight_decay": 0}
CONFIGS = {
    "gcn_arxiv": dict(dataset="arxiv", model="GCN", batch_size=40, VR_update=True, params=dict(
        architecture=dict(num_layers=3, hidden_channels=256, dropout=0.5, drop_input=False,
                          batch_norm=True, residual=False),
        num_parts=40, pool_size=2, lr=0.01, grad_norm=None, epochs=4, **common)),
    "appnp_reddit": dict(dataset="reddit", model="APPNP", batch_size=100, VR_update=True, loop=False,
                         params=dict(architecture=dict(num_layers=5, hidden_channels=256, alpha=0.1,
                                                       dropout=0.3),
                                     num_parts=200, pool_size=2, lr=0.01, grad_norm=1.0, epochs=4,
                                     **common)),
    # conf/model/gcn2.yaml: arxiv (shared weights, batch norm) and reddit (drop_input, batch norm)
    "gcn2_arxiv": dict(dataset="arxiv", model="GCN2", batch_size=20, VR_update=True, params=dict(
        architecture=dict(num_layers=4, hidden_channels=256, dropout=0.3, drop_input=False,
                          batch_norm=True, residual=False, shared_weights=True, alpha=0.2, theta=0.5),
        num_parts=80, pool_size=2, lr=0.01, grad_norm=None, epochs=4, **common)),
    "gcn2_reddit": dict(dataset="reddit", model="GCN2", batch_size=100, VR_update=True, params=dict(
        architecture=dict(num_layers=4, hidden_channels=256, dropout=0.5, drop_input=True,
                          batch_norm=True, residual=False, shared_weights=False, alpha=0.1, theta=0.5),
        num_parts=200, pool_size=2, lr=0.01, grad_norm=None, epochs=4, **common)),
    # conf/model/graphsage.yaml: arxiv (batch norm)
    "sage_arxiv": dict(dataset="arxiv", model="GraphSAGE", batch_size=40, VR_update=True, params=dict(
        architecture=dict(num_layers=3, hidden_channels=256, dropout=0.5, drop_input=False,
                          batch_norm=True, residual=False),
        num_parts=80, pool_size=2, lr=0.01, grad_norm=None, epochs=4, **common)),
    "sage_products_host": dict(dataset="products", model="GraphSAGE", batch_size=1, VR_update=False,
                               history_device="host", params=dict(
        architecture=dict(num_layers=3, hidden_channels=256, dropout=0.3, drop_input=False,
                          batch_norm=False, residual=False),
        num_parts=150, pool_size=2, lr=0.005, grad_norm=None, epochs=2, **common)),
}
c = dict(CONFIGS[which])
if os.environ.get("GAS") == "1":     # the push/pull step instead of the incremental one
    c["VR_update"] = False
if os.environ.get("EPOCHS"):
    c["params"] = dict(c["params"], epochs=int(os.environ["EPOCHS"]))
conf = {"seed": 0, "dataset": {"name": c["dataset"]}, "batch_size": c["batch_size"],
        "VR_update": c["VR_update"], "history_device": c.get("history_device", "model"),
        "synthetic_scale": float(os.environ.get("SCALE", "1.0")),
        "model": {"name": c["model"], "norm": True, "loop": c.get("loop", True),
                  "params": {c["dataset"]: c["params"]}}}
t0 = time.perf_counter()
res = train.run(conf)
_lib.check_device_errors()
print(which, f"total {time.perf_counter() - t0:.1f} s")
for h in res["history"]:
    print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in h.items()}), flush=True)
