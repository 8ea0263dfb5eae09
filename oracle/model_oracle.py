"""ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by the shipped package).

Plain numpy restatement, on the WHOLE graph, of what the reference's models compute in
eval mode, layer by layer.  Because layer-wise inference with up-to-date histories is
exact, these full-graph values are what a refresh sweep must leave in the tables:

    histories[i]    == input of layer i                      (models/gcn.py:367-369,390)
    histories_ag[i] == A @ histories[i]  (mean for SAGE)      (models/gcn.py:361,386-387)
    _out            == model output                          (models/gcn.py:402-405)

and the incremental-aggregation forward on a batch is
    h = A_ib @ (x - M_in[batch]) + M_ag[batch]               (models/gcn2.py:255)
evaluated with explicit table contents.

PARITY UNPINNED for the layer arithmetic itself (GCNConv / GCN2Conv / SAGEConv live in
torch-geometric, absent here and un-pinned in the reference); formulas as published and
as spelled out in models/graphsage.py:623-644, models/gcn.py:175-178, models/appnp.py:85.
"""
import numpy as np

from . import cpu_path as cp

F32 = np.float32


def _lin(x, w, b=None):
    y = x.astype(F32) @ w.T.astype(F32)
    return y if b is None else y + b


def _bn_eval(h, sd, prefix, eps=1e-5):
    rm, rv = sd[prefix + "running_mean"], sd[prefix + "running_var"]
    return ((h - rm) / np.sqrt(rv + F32(eps)) * sd[prefix + "weight"] + sd[prefix + "bias"]).astype(F32)


def _pad(x, width):
    if x.shape[1] >= width:
        return x
    out = np.zeros((x.shape[0], width), dtype=F32)
    out[:, :x.shape[1]] = x
    return out


def full_graph(kind, sd, cfg, rowptr, col, val, x):
    """Returns (layer_inputs, layer_aggregates, output); sd = numpy state dict."""
    L = cfg["num_layers"]
    reduce = "mean" if kind == "sage" else "sum"
    v = None if kind == "sage" else val
    inputs, aggs = [], []
    if kind in ("gcn", "sage"):
        h = x.astype(F32)
        if cfg.get("linear"):
            h = np.maximum(_lin(h, sd["lins.0.weight"], sd["lins.0.bias"]), 0)
        for i in range(L):
            inputs.append(h)
            agg = cp.spmm(rowptr, col, v, h, reduce)
            aggs.append(agg)
            if kind == "gcn":
                o = _lin(agg, sd[f"convs.{i}.lin.weight"]) + sd[f"convs.{i}.bias"]
            else:
                o = _lin(agg, sd[f"convs.{i}.lin_l.weight"], sd[f"convs.{i}.lin_l.bias"]) \
                    + _lin(h, sd[f"convs.{i}.lin_r.weight"])
            if i < L - 1 or cfg.get("linear"):
                if cfg.get("batch_norm"):
                    o = _bn_eval(o, sd, f"bns.{i}.")
                if cfg.get("residual") and o.shape[1] == h.shape[1]:
                    o = o + h
                o = np.maximum(o, 0)
            h = o
        if cfg.get("linear"):
            h = _lin(h, sd["lins.1.weight"], sd["lins.1.bias"])
        return inputs, aggs, h
    if kind == "gcn2":
        alpha = cfg["alpha"]
        h = x0 = np.maximum(_lin(x, sd["lins.0.weight"], sd["lins.0.bias"]), 0)
        for i in range(L):
            inputs.append(h)
            agg = cp.spmm(rowptr, col, val, h)
            aggs.append(agg)
            beta = np.log(cfg["theta"] / (i + 1) + 1)
            w2 = sd.get(f"convs.{i}.weight2")
            ax = agg * F32(1 - alpha)
            x0a = x0 * F32(alpha)
            if w2 is None:
                p = ax + x0a
                o = F32(1 - beta) * p + F32(beta) * (p @ sd[f"convs.{i}.weight1"])
            else:
                o = F32(1 - beta) * ax + F32(beta) * (ax @ sd[f"convs.{i}.weight1"]) \
                    + F32(1 - beta) * x0a + F32(beta) * (x0a @ w2)
            if cfg.get("batch_norm"):
                o = _bn_eval(o, sd, f"bns.{i}.")
            if cfg.get("residual"):
                o = o + h
            h = np.maximum(o, 0).astype(F32)
        return inputs, aggs, _lin(h, sd["lins.1.weight"], sd["lins.1.bias"])
    if kind == "appnp":
        alpha = cfg["alpha"]
        h = np.maximum(_lin(x, sd["lins.0.weight"], sd["lins.0.bias"]), 0)
        h = x0 = _lin(h, sd["lins.1.weight"], sd["lins.1.bias"])
        for i in range(L):
            inputs.append(h)
            agg = cp.spmm(rowptr, col, val, h)
            aggs.append(agg)
            h = (F32(1 - alpha) * agg + F32(alpha) * x0).astype(F32)
        return inputs, aggs, h
    raise ValueError(kind)


def vr_forward(kind, sd, cfg, ib_rowptr, ib_col, ib_val, x_batch, tables_in, tables_ag, rows):
    """Incremental-aggregation forward on one batch: `rows` = global ids of the batch,
    tables_* = lists of [N, width] arrays.  Eval mode (no dropout)."""
    L = cfg["num_layers"]
    reduce = "mean" if kind == "sage" else "sum"
    v = None if kind == "sage" else ib_val

    def vr_agg(i, h):
        w = h.shape[1]
        return cp.incagg(ib_rowptr, ib_col, v, h, tables_in[i][rows][:, :w],
                         tables_ag[i][rows][:, :w], reduce)

    if kind in ("gcn", "sage"):
        h = x_batch.astype(F32)
        if cfg.get("linear"):
            h = np.maximum(_lin(h, sd["lins.0.weight"], sd["lins.0.bias"]), 0)
        for i in range(L):
            agg = vr_agg(i, h)
            if kind == "gcn":
                o = _lin(agg, sd[f"convs.{i}.lin.weight"]) + sd[f"convs.{i}.bias"]
            else:
                o = _lin(agg, sd[f"convs.{i}.lin_l.weight"], sd[f"convs.{i}.lin_l.bias"]) \
                    + _lin(h, sd[f"convs.{i}.lin_r.weight"])
            if i < L - 1 or cfg.get("linear"):
                if cfg.get("batch_norm"):
                    o = _bn_eval(o, sd, f"bns.{i}.")
                if cfg.get("residual") and o.shape[1] == h.shape[1]:
                    o = o + h
                o = np.maximum(o, 0)
            h = o
        if cfg.get("linear"):
            h = _lin(h, sd["lins.1.weight"], sd["lins.1.bias"])
        return h
    if kind == "gcn2":
        alpha = cfg["alpha"]
        h = x0 = np.maximum(_lin(x_batch, sd["lins.0.weight"], sd["lins.0.bias"]), 0)
        for i in range(L):
            agg = vr_agg(i, h)
            beta = np.log(cfg["theta"] / (i + 1) + 1)
            w2 = sd.get(f"convs.{i}.weight2")
            ax, x0a = agg * F32(1 - alpha), x0 * F32(alpha)
            if w2 is None:
                p = ax + x0a
                o = F32(1 - beta) * p + F32(beta) * (p @ sd[f"convs.{i}.weight1"])
            else:
                o = F32(1 - beta) * ax + F32(beta) * (ax @ sd[f"convs.{i}.weight1"]) \
                    + F32(1 - beta) * x0a + F32(beta) * (x0a @ w2)
            if cfg.get("batch_norm"):
                o = _bn_eval(o, sd, f"bns.{i}.")
            if cfg.get("residual"):
                o = o + h
            h = np.maximum(o, 0).astype(F32)
        return _lin(h, sd["lins.1.weight"], sd["lins.1.bias"])
    if kind == "appnp":
        alpha = cfg["alpha"]
        h = np.maximum(_lin(x_batch, sd["lins.0.weight"], sd["lins.0.bias"]), 0)
        h = x0 = _lin(h, sd["lins.1.weight"], sd["lins.1.bias"])
        for i in range(L):
            h = (F32(1 - alpha) * vr_agg(i, h) + F32(alpha) * x0).astype(F32)
        return h
    raise ValueError(kind)


def padded(x, width):
    return _pad(np.asarray(x, dtype=F32), width)
