"""Worker of tests/test_gpu_train_graphs.py, run as a process of its own (a GPU fault then ends
this process and not the test session): the training driver with recorded steps and a recorded
refresh sweep against the same run issued eagerly.

argv: model name (GCN2 | GCN | GraphSAGE | APPNP), step kind (gas | vr), epochs.
Prints one JSON line with the largest deviations."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT, os.path.join(ROOT, "incagg-gnn_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402


def conf_for(model, vr, epochs, graphs):
    arch = {"num_layers": 3, "hidden_channels": 64, "dropout": 0.0, "drop_input": False,
            "batch_norm": False, "residual": False}
    if model == "GCN2":
        arch.update(alpha=0.1, theta=0.5, shared_weights=False)
    if model == "APPNP":
        arch = {"num_layers": 3, "hidden_channels": 64, "alpha": 0.1, "dropout": 0.0}
    params = {"num_parts": 12, "max_steps": -1, "pool_size": 2, "num_workers": 0, "lr": 0.01,
              "reg_weight_decay": 0.0, "nonreg_weight_decay": 0.0, "grad_norm": 1.0,
              "epochs": epochs, "architecture": arch}
    return {"seed": 5, "dataset": {"name": "arxiv"}, "synthetic_scale": 0.05, "batch_size": 1,
            "VR_update": vr, "graphs": graphs,
            "model": {"name": model, "norm": True, "loop": True, "params": {"arxiv": params}}}


def main():
    model, kind, epochs = sys.argv[1], sys.argv[2], int(sys.argv[3])
    from torch_geometric_autoscale import train, _lib
    vr = kind == "vr"
    runs = {}
    for graphs in (False, True):
        res = train.run(conf_for(model, vr, epochs, graphs))
        torch.cuda.synchronize()
        _lib.check_device_errors()
        net = res["model"]
        runs[graphs] = {
            "params": {k: v.detach().cpu() for k, v in net.state_dict().items()},
            "tables": [h.emb.cpu() for h in list(net.histories) + list(net.histories_ag)],
            "out": res["out"].cpu(), "loss": [h["loss"] for h in res["history"]],
            "graphs": res["graphs"],
        }
        print(f"graphs={graphs}: {res['graphs']} losses {runs[graphs]['loss']}", file=sys.stderr,
              flush=True)

    def dev(a, b):
        return float((a - b).abs().max() / (1.0 + b.abs().max()))
    eager, rec = runs[False], runs[True]
    report = {
        "recorded": rec["graphs"], "eager": eager["graphs"],
        "param_dev": max(dev(rec["params"][k], eager["params"][k]) for k in eager["params"]),
        "table_dev": max(dev(a, b) for a, b in zip(rec["tables"], eager["tables"])),
        "out_dev": dev(rec["out"], eager["out"]),
        "loss_dev": max(abs(a - b) for a, b in zip(rec["loss"], eager["loss"])),
        "loss": rec["loss"],
    }
    print(json.dumps(report), flush=True)


if __name__ == "__main__":
    main()
