"""End-to-end run of the training driver at the headline scale: GCNII (8 x 256) on the
ogbn-products-sized synthetic graph, 150 partitions, tables in HBM, batches built on the GPU."""
import logging, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "incagg-gnn_amd")); sys.path.insert(0, ROOT)
logging.basicConfig(level=logging.INFO, format="%(message)s")
from torch_geometric_autoscale import train, _lib
vr = os.environ.get("VR", "0") == "1"
scale = float(os.environ.get("SCALE", "1.0"))
parts = max(2, int(round(150 * scale)))
conf = {"seed": 0, "dataset": {"name": "products"}, "batch_size": 1, "VR_update": vr,
        "synthetic_scale": scale,
        "model": {"name": "GCN2", "norm": True, "loop": True, "params": {"products": {
            "architecture": {"num_layers": 8, "hidden_channels": 256, "dropout": 0.0, "drop_input": False,
                             "batch_norm": False, "residual": False, "shared_weights": False,
                             "alpha": 0.1, "theta": 0.5},
            "num_parts": parts, "max_steps": -1, "pool_size": 1, "num_workers": 0, "lr": 0.001,
            "reg_weight_decay": 0, "nonreg_weight_decay": 0, "grad_norm": None, "epochs": int(os.environ.get("EPOCHS", "4"))}}}}
if os.environ.get("GRAPHS"):
    conf["graphs"] = os.environ["GRAPHS"] == "1"
res = train.run(conf)
_lib.check_device_errors()
print(json.dumps(res["graphs"]), flush=True)
for h in res["history"]:
    print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in h.items()}), flush=True)
