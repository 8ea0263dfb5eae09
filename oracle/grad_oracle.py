"""ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by the shipped package).

Whole-model loss and weight gradients of ONE training step on a batch, in float64, by torch
autograd over dense tensors: the same layer formulas as ``model_oracle`` (which states them for
inference on the whole graph), here in training mode on a partition batch,

* push/pull step (models/gcn.py:97-205, models/gcn2.py:78-185, models/appnp.py:44-106,
  models/graphsage.py:110-366): layer 0 aggregates the (embedded) features of all rows of
  the batch, every later layer the fresh in-batch rows plus the out-of-batch rows of that
  layer's History table, which are constants;
* incremental-aggregation step (models/gcn.py:209-279, models/gcn2.py:187-323,
  models/appnp.py:108-137, models/graphsage.py:539-707): every layer computes
  ``A_ib (x - M_in) + M_ag`` over in-batch edges, ``M_in`` / ``M_ag`` constants;

followed by the mean cross entropy over the batch rows (main.py:80).  Dropout is off.

PARITY UNPINNED like ``model_oracle``: the layer arithmetic is third-party code the reference
neither vendors nor pins; the formulas are the published ones.
"""
import numpy as np
import torch

F64 = torch.float64


def _params(sd):
    out = {}
    for k, v in sd.items():
        t = torch.from_numpy(np.asarray(v)).to(F64) if np.issubdtype(np.asarray(v).dtype, np.floating) \
            else torch.from_numpy(np.asarray(v))
        if t.dtype == F64 and "running_" not in k:
            t.requires_grad_(True)
        out[k] = t
    return out


def _aggregate(rowptr, col, val, x, n_rows, mean):
    rowptr = torch.from_numpy(np.asarray(rowptr, dtype=np.int64))
    col = torch.from_numpy(np.asarray(col, dtype=np.int64))
    deg = rowptr[1:] - rowptr[:-1]
    row = torch.repeat_interleave(torch.arange(n_rows), deg)
    w = torch.ones(col.numel(), dtype=F64) if val is None or mean \
        else torch.from_numpy(np.asarray(val)).to(F64)
    if mean:
        w = w / deg.clamp(min=1).to(F64)[row]
    out = torch.zeros((n_rows, x.size(1)), dtype=F64)
    return out.index_add(0, row, w[:, None] * x[col])


def _bn_train(h, p, prefix, eps=1e-5):
    return torch.nn.functional.batch_norm(h, None, None, p[prefix + "weight"], p[prefix + "bias"],
                                          training=True, eps=eps)


def step_gradients(kind, sd, cfg, rowptr, col, val, x, batch_size, y, halo=None, m_in=None,
                   m_ag=None):
    """``(loss, out, {name: gradient})`` as float64 numpy values.

    ``rowptr/col/val``: the batch adjacency (``[B, M]``; ``[B, B]`` for the incremental step);
    ``x``: features of the batch's rows (all ``M`` of them for the push/pull step);
    ``halo[i]`` (push/pull): rows ``n_id[B:]`` of ``histories[i]`` for ``i >= 1``;
    ``m_in[i]`` / ``m_ag[i]`` (incremental step): rows ``n_id[:B]`` of both tables of layer ``i``.
    """
    p = _params(sd)
    b, L = batch_size, cfg["num_layers"]
    vr = m_in is not None
    mean = kind == "sage"
    x = torch.from_numpy(np.asarray(x)).to(F64)
    if vr:
        x = x[:b]

    def const(a, width):
        return torch.from_numpy(np.asarray(a)).to(F64)[:, :width]

    def agg_of(i, h_all, h):
        """Aggregate consumed by layer ``i``: ``h_all`` are the rows layer 0 sees (all rows of a
        push/pull batch), ``h`` the fresh in-batch rows of later layers."""
        if vr:
            src = h_all if i == 0 else h
            w = src.size(1)
            return _aggregate(rowptr, col, val, src[:b] - const(m_in[i], w), b, mean) \
                + const(m_ag[i], w)
        src = h_all if i == 0 else torch.cat([h, const(halo[i], h.size(1))])
        return _aggregate(rowptr, col, val, src, b, mean)

    def lin(t, w, bias=None):
        o = t @ p[w].T
        return o if bias is None or bias not in p else o + p[bias]

    if kind in ("gcn", "sage"):
        h_all = x
        if cfg.get("linear"):
            h_all = torch.relu(lin(h_all, "lins.0.weight", "lins.0.bias"))
        h = h_all[:b]
        for i in range(L):
            agg = agg_of(i, h_all, h)
            root = h_all[:b] if i == 0 else h
            if kind == "gcn":
                o = lin(agg, f"convs.{i}.lin.weight") + p[f"convs.{i}.bias"]
            else:
                o = lin(agg, f"convs.{i}.lin_l.weight", f"convs.{i}.lin_l.bias") \
                    + lin(root, f"convs.{i}.lin_r.weight")
            if i < L - 1 or cfg.get("linear"):
                if cfg.get("batch_norm"):
                    o = _bn_train(o, p, f"bns.{i}.")
                if cfg.get("residual") and o.size(1) == root.size(1):
                    o = o + root
                o = torch.relu(o)
            h = o
        out = lin(h, "lins.1.weight", "lins.1.bias") if cfg.get("linear") else h
    elif kind == "gcn2":
        alpha = cfg["alpha"]
        h_all = torch.relu(lin(x, "lins.0.weight", "lins.0.bias"))
        x0 = h_all[:b]
        h = x0
        for i in range(L):
            agg = agg_of(i, h_all, h)
            beta = float(np.log(cfg["theta"] / (i + 1) + 1))
            ax, x0a = (1 - alpha) * agg, alpha * x0
            if f"convs.{i}.weight2" not in p:
                q = ax + x0a
                o = (1 - beta) * q + beta * (q @ p[f"convs.{i}.weight1"])
            else:
                o = (1 - beta) * ax + beta * (ax @ p[f"convs.{i}.weight1"]) \
                    + (1 - beta) * x0a + beta * (x0a @ p[f"convs.{i}.weight2"])
            if cfg.get("batch_norm"):
                o = _bn_train(o, p, f"bns.{i}.")
            if cfg.get("residual"):
                o = o + h
            h = torch.relu(o)
        out = lin(h, "lins.1.weight", "lins.1.bias")
    elif kind == "appnp":
        alpha = cfg["alpha"]
        h_all = lin(torch.relu(lin(x, "lins.0.weight", "lins.0.bias")), "lins.1.weight",
                    "lins.1.bias")
        x0 = h_all[:b]
        h = x0
        for i in range(L):
            h = (1 - alpha) * agg_of(i, h_all, h) + alpha * x0
        out = h
    else:
        raise ValueError(kind)

    loss = torch.nn.functional.cross_entropy(out, torch.from_numpy(np.asarray(y, dtype=np.int64)))
    loss.backward()
    grads = {k: t.grad.numpy() for k, t in p.items() if t.requires_grad and t.grad is not None}
    return float(loss.detach()), out.detach().numpy(), grads
