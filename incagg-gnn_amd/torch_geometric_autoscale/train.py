"""Training driver with the semantics of the reference's main.py:46-264: partition the graph,
fill the History tables with one refresh sweep, then per epoch a pass over the shuffled
partition batches (at most ``max_steps`` of them) followed by a refresh sweep whose output is
scored on the train / val / test masks.

The configuration is a plain nested dict with the reference's hydra keys
(conf/config.yaml, conf/model/*.yaml): ``seed, device, root, dataset.name, model.name,
model.norm, model.loop, model.params.<dataset>.{architecture.*, num_parts, batch_size,
max_steps, pool_size, num_workers, lr, reg_weight_decay, nonreg_weight_decay, grad_norm,
epochs, edge_dropout}, batch_size, VR_update, dropout, num_neighbors, drift_norm, log_every,
aggregate_combined, use_aggregation``.  :func:`load_conf` reads the reference's YAML files from
a directory the user points at; ``key=value`` overrides use hydra's dotted syntax:

    python -m torch_geometric_autoscale.train model=gcn2 dataset=products VR_update=true \\
        conf_dir=/path/to/conf model.params.epochs=3
"""
import copy
import logging
import random
import sys
import time
from typing import Any, Dict, Optional

import numpy as np
import torch

from . import models
from .data import gcn_norm, get_data
from .loader import DeviceSubgraphLoader, EvalSubgraphLoader, SubgraphLoader
from .metis import load_partition, metis, permute
from .nn import CrossEntropyLoss
from .optim import Adam
from .utils import compute_micro_f1, dropout

log = logging.getLogger("incagg.train")

DEFAULTS: Dict[str, Any] = {
    "seed": 0, "device": 0, "root": None, "log_every": 1, "batch_size": 1, "VR_update": False,
    "dropout": None, "num_neighbors": -1, "drift_norm": 2, "period_updates_in_one_epoch": 0,
    "aggregate_combined": True, "use_aggregation": True, "partition_file": None,
    "history_device": "model", "synthetic_scale": 1.0,
    # "device": graph, features and labels live in HBM and batches are built there by the HIP
    # batch builder (needs a GPU); "host": the reference's DataLoader path (loader.py:95-284)
    "batch_builder": "auto",
    # HIP graphs: training steps over static batches (built once and kept by the loader) and the
    # refresh sweep are recorded at their first execution and replayed from then on.  "auto": on
    # a GPU with the History tables in HBM.
    "graphs": "auto",
    "dataset": {"name": "arxiv"},
    "model": {"name": "GCN", "norm": True, "loop": True, "params": {}},
}


def mini_train(model, loader, criterion, optimizer, max_steps, grad_norm=None, edge_dropout=0.0,
               epoch=0, VR_update=False, period_updates_in_one_epoch=0, ptr=None, debug_flag=False,
               period_update_trigger=None, drift_norm=2, aggregate_combined=True,
               use_aggregation=True, graphs=None):
    """One training epoch (main.py:47-96).  Batches without training nodes are skipped; the
    loss is averaged over training nodes; the epoch stops after ``max_steps`` batches.

    ``graphs`` (a ``graphs.StepGraphs``): with a loader that keeps its batch objects
    (``static_batches``) and no edge dropout, the step of a batch -- forward, loss, backward,
    gradient clipping, optimizer step, loss bookkeeping -- is recorded into a HIP graph right
    after its first (eager) execution and replayed whenever the batch comes up again."""
    model.train()
    total_loss, total_examples = 0.0, 0     # the loss sum stays on the device until the end
    replayable = (graphs is not None and getattr(loader, "static_batches", False)
                  and not edge_dropout and torch.device(model.device).type == "cuda")
    if replayable:
        total_loss = graphs.__dict__.get("_loss_sum")
        if total_loss is None:
            total_loss = graphs.__dict__["_loss_sum"] = torch.zeros((), device=model.device)
        total_loss.zero_()
    for i, (batch, batch_size, n_id, offset, count) in enumerate(loader):
        x = batch.x.to(model.device)
        adj_t = batch.adj_t.to(model.device)
        y = batch.y[:batch_size].to(model.device)
        train_mask = batch.train_mask[:batch_size].to(model.device)
        n_train = batch.__dict__.get("_n_train")       # counted once per batch object
        if n_train is None:
            n_train = int(train_mask.sum())
            batch.__dict__["_n_train"] = n_train
            # row numbers instead of a boolean mask: indexing with them has a static shape
            batch.__dict__["_train_idx"] = train_mask.nonzero(as_tuple=False).view(-1)
        if n_train == 0:
            continue
        train_idx = batch.__dict__["_train_idx"]
        adj_t = dropout(adj_t, p=edge_dropout)

        def step(x=x, adj_t=adj_t, y=y, train_idx=train_idx, batch_size=batch_size, n_id=n_id,
                 offset=offset, count=count, i=i):
            if VR_update:
                ret = model.VR_call(x, adj_t, batch_size, n_id, offset, count,
                                    debug_flag=debug_flag, drift_norm=drift_norm, epoch=epoch,
                                    batch_idx=i)
            else:
                ret = model(x, adj_t, batch_size, n_id, offset, count, drift_norm=drift_norm,
                            aggregate_combined=aggregate_combined,
                            use_aggregation=use_aggregation)
            out = ret['out']
            optimizer.zero_grad()
            loss = criterion(out.index_select(0, train_idx), y.index_select(0, train_idx))
            loss.backward()
            if grad_norm is not None:
                torch.nn.utils.clip_grad_norm_(model.parameters(), grad_norm)
            optimizer.step()
            return loss.detach()

        if replayable:
            key = ("train-step", id(batch), bool(VR_update))    # (the loader keeps `batch` alive)
            if key in graphs:
                graphs.replay(key)
            else:
                def counted(step=step, n_train=n_train):
                    total_loss.add_(step() * n_train)
                graphs.warm(counted, times=1)       # the step itself, issued eagerly ...
                graphs.capture(key, counted)        # ... and its recording (runs nothing)
        else:
            total_loss = total_loss + step() * n_train
        total_examples += n_train
        # the epoch may stop early so that the tables get refreshed (main.py:90-92)
        if (i + 1) >= max_steps and (i + 1) < len(loader):
            break
    return {"loss": float(total_loss) / max(total_examples, 1)}


@torch.no_grad()
def full_test(model, data):
    model.eval()
    return model(data.x.to(model.device), data.adj_t.to(model.device))['out'].cpu()


@torch.no_grad()
def mini_test(model, loader, use_aggregation=True):
    model.eval()
    return model(loader=loader, use_aggregation=use_aggregation)


# ------------------------------------------------------------------ configuration
def _merge(base: Dict[str, Any], extra: Dict[str, Any]) -> Dict[str, Any]:
    out = copy.deepcopy(base)
    for k, v in extra.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict):
            out[k] = _merge(out[k], v)
        else:
            out[k] = copy.deepcopy(v)
    return out


def _parse_scalar(text: str):
    import yaml
    return yaml.safe_load(text)


def load_conf(conf_dir: Optional[str], model: str, dataset: str) -> Dict[str, Any]:
    """``DEFAULTS`` + ``<conf_dir>/config.yaml`` + ``<conf_dir>/model/<model>.yaml`` (the layout
    of the reference's conf/), with ``dataset.name`` set.  Without a directory only the
    defaults apply and ``model.params`` has to come from overrides or the caller."""
    import os
    import yaml
    conf = copy.deepcopy(DEFAULTS)
    if conf_dir:
        top = os.path.join(conf_dir, "config.yaml")
        if os.path.isfile(top):
            with open(top) as f:
                loaded = yaml.safe_load(f) or {}
            loaded.pop("defaults", None)
            loaded.pop("hydra", None)
            conf = _merge(conf, loaded)
        path = os.path.join(conf_dir, "model", f"{model.lower()}.yaml")
        with open(path) as f:
            conf["model"] = _merge(conf["model"], yaml.safe_load(f) or {})
    else:
        conf["model"]["name"] = model
    conf["dataset"] = {"name": dataset}
    return conf


def apply_overrides(conf: Dict[str, Any], overrides) -> Dict[str, Any]:
    """``a.b.c=value`` items (hydra's syntax); values are parsed as YAML scalars."""
    conf = copy.deepcopy(conf)
    for item in overrides:
        key, _, text = item.partition("=")
        node = conf
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = _parse_scalar(text)
    return conf


# ------------------------------------------------------------------ the run
def run(conf: Dict[str, Any]) -> Dict[str, Any]:
    """main.py:112-264.  Returns the per-epoch history and the final scores."""
    conf = _merge(DEFAULTS, conf)
    seed = int(conf["seed"])
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)
    name = conf["dataset"]["name"]
    params = conf["model"]["params"]
    if name in params and isinstance(params[name], dict):
        params = params[name]                       # main.py:122
    params = copy.deepcopy(params)
    arch = dict(params.get("architecture", {}))
    if conf.get("dropout") is not None:             # main.py:127
        arch["dropout"] = conf["dropout"]
    edge_dropout = params.get("edge_dropout", 0.0) or 0.0
    grad_norm = params.get("grad_norm")
    grad_norm = None if grad_norm is None or isinstance(grad_norm, str) else grad_norm
    device = torch.device(f"cuda:{conf['device']}" if torch.cuda.is_available() else "cpu")
    batch_size = int(conf["batch_size"])
    num_parts = int(params["num_parts"])

    t = time.perf_counter()
    # synthetic graphs are sampled, given self loops and normalised on the GPU when there is one
    # (seconds instead of minutes at ogbn-products size), then handed to the host-side loaders
    on_gpu = device.type == "cuda"
    data, in_channels, out_channels = get_data(conf["root"], name, num_parts=num_parts,
                                               scale=float(conf["synthetic_scale"]), seed=seed,
                                               planted_labels=True, device=device,
                                               keep_on_device=on_gpu)
    log.info(f"data ready [{time.perf_counter() - t:.2f}s]")
    if conf.get("partition_file"):
        perm, ptr = load_partition(conf["partition_file"], data.num_nodes)
        data = permute(data, perm, log=False)
    elif "ptr" in data:                              # synthetic graphs come pre-partitioned
        ptr = data.ptr
    else:
        perm, ptr = metis(data.adj_t, num_parts=num_parts, log=False)
        data = permute(data, perm, log=False)
    if conf["model"].get("loop", True):
        data.adj_t = data.adj_t.set_diag()
    if conf["model"].get("norm", True):
        data.adj_t = gcn_norm(data.adj_t, add_self_loops=False)
    if data.adj_t.device().type != "cpu":
        data.adj_t = data.adj_t.to("cpu")
    criterion = CrossEntropyLoss() if data.y.dim() == 1 else torch.nn.BCEWithLogitsLoss()

    vr = bool(conf["VR_update"])
    loader_data = data.__class__(**{k: v for k, v in data if k != "ptr"})
    builder = conf["batch_builder"]
    if builder == "auto":
        builder = "device" if device.type == "cuda" else "host"
    if builder == "device":
        dev_data = loader_data.__class__(**{
            k: (v.to(device) if isinstance(v, torch.Tensor) and k != "adj_t" else v)
            for k, v in loader_data})
        # (like the reference's collate choice, loader.py:187-188: sampling applies to the
        # one-hop batches of the push/pull step only)
        train_loader = DeviceSubgraphLoader(dev_data, ptr, device, batch_size=batch_size,
                                            shuffle=True, in_batch_only=vr,
                                            num_neighbors=-1 if vr else conf["num_neighbors"])
        eval_loader = DeviceSubgraphLoader(dev_data, ptr, device, batch_size=batch_size)
    else:
        train_loader = SubgraphLoader(loader_data, ptr, batch_size=batch_size, shuffle=True,
                                      num_workers=int(params.get("num_workers", 0)),
                                      num_neighbors=conf["num_neighbors"], type='train', IB=vr,
                                      log=False)
        eval_loader = EvalSubgraphLoader(loader_data, ptr, batch_size=batch_size, log=False)
    buffer_size = max(n_id.numel() for _, _, n_id, _, _ in eval_loader) * 2     # main.py:177

    GNN = getattr(models, conf["model"]["name"])
    hist_dev = device if conf["history_device"] == "model" else None
    model = GNN(num_nodes=data.num_nodes, in_channels=in_channels, out_channels=out_channels,
                pool_size=params.get("pool_size"), buffer_size=buffer_size, device=hist_dev,
                **arch).to(device)
    groups = [                                                                    # main.py:196-201
        dict(params=list(model.reg_modules.parameters()),
             weight_decay=params.get("reg_weight_decay", 0.0)),
        dict(params=list(model.nonreg_modules.parameters()),
             weight_decay=params.get("nonreg_weight_decay", 0.0)),
    ]
    if device.type == "cuda":       # the same update, one launch per group
        optimizer = Adam([g for g in groups if g["params"]], lr=params.get("lr", 0.01))
    else:
        optimizer = torch.optim.Adam(groups, lr=params.get("lr", 0.01))
    use_agg = bool(conf["use_aggregation"])

    use_graphs = conf["graphs"]
    if use_graphs == "auto":
        use_graphs = device.type == "cuda" and hist_dev is not None and model.pool is None
    step_graphs = sweep_graph = None
    if use_graphs:
        from .graphs import StepGraphs, SweepGraph
        step_graphs = StepGraphs(device)
        sweep_graph = SweepGraph(model, eval_loader, with_aggregates=vr, use_aggregation=use_agg,
                                 graphs=step_graphs)
    # the refresh output travels to the host through ONE pinned buffer (no pageable copies
    # between replays of recorded work)
    out_host = None

    def refresh():
        nonlocal step_graphs
        model.eval()
        if sweep_graph is not None:
            out = sweep_graph()
            if not sweep_graph.recorded:    # sweeps stay eager: then so do the steps
                step_graphs = None
            return out
        if vr:
            return model.mini_inference_vr(loader=eval_loader, use_aggregation=use_agg)
        return model.mini_inference(loader=eval_loader, use_aggregation=use_agg)

    def to_host(t):
        nonlocal out_host
        if not t.is_cuda:
            return t
        if out_host is None or out_host.shape != t.shape or out_host.dtype != t.dtype:
            out_host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        out_host.copy_(t, non_blocking=True)
        torch.cuda.current_stream(t.device).synchronize()
        return out_host

    t = time.perf_counter()
    refresh()                                                                      # main.py:211-215
    log.info(f"histories filled [{time.perf_counter() - t:.2f}s]")
    max_steps = params.get("max_steps", -1)
    max_steps = max_steps if max_steps != -1 else int(num_parts / batch_size)     # main.py:223
    best_val = final_test = 0.0
    history = []
    out = None
    y_host = data.y
    for epoch in range(int(params.get("epochs", 1))):
        t0 = time.perf_counter()
        res = mini_train(model, train_loader, criterion, optimizer, max_steps, grad_norm,
                         edge_dropout, epoch, VR_update=vr, ptr=ptr, drift_norm=conf["drift_norm"],
                         aggregate_combined=bool(conf["aggregate_combined"]), use_aggregation=use_agg,
                         graphs=step_graphs)
        t1 = time.perf_counter()
        out = to_host(refresh())
        t2 = time.perf_counter()
        train_acc = compute_micro_f1(out, y_host, data.train_mask)
        val_acc = compute_micro_f1(out, y_host, data.val_mask)
        test_acc = compute_micro_f1(out, y_host, data.test_mask)
        if val_acc > best_val:
            best_val, final_test = val_acc, test_acc
        history.append({"epoch": epoch, "loss": res["loss"], "train": train_acc, "val": val_acc,
                        "test": test_acc, "train_seconds": t1 - t0, "refresh_seconds": t2 - t1})
        if epoch % int(conf["log_every"]) == 0:
            log.info(f"Epoch: {epoch:04d}, Loss: {res['loss']:.4f}, Train: {train_acc:.4f}, "
                     f"Val: {val_acc:.4f}, Test: {test_acc:.4f}, Final: {final_test:.4f}")
    log.info(f"Val: {best_val:.4f}, Test: {final_test:.4f}")
    # (beyond the reference's scores: the trained model, the graph it was trained on and the
    # output of the last refresh, for callers that go on with them)
    return {"history": history, "val": best_val, "test": final_test, "max_steps": max_steps,
            "model": model, "data": data, "out": None if out is None else out.clone(),
            "graphs": {"train_steps_recorded": 0 if step_graphs is None else sum(
                           1 for k in step_graphs.graphs if k[0] == "train-step"),
                       "sweep_recorded": bool(sweep_graph is not None and sweep_graph.recorded)}}


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    logging.basicConfig(level=logging.INFO, format="%(message)s")
    plain = dict(a.partition("=")[::2] for a in argv if "=" in a)
    conf = load_conf(plain.pop("conf_dir", None), plain.pop("model", "gcn"),
                     plain.pop("dataset", "arxiv"))
    conf = apply_overrides(conf, [f"{k}={v}" for k, v in plain.items()])
    run(conf)
    return 0


if __name__ == "__main__":
    sys.exit(main())
