"""The training driver end to end on the GPU (HIP kernels, History tables in HBM): a few
epochs on a small synthetic graph with planted labels must learn them, in both training modes
and with host-resident tables behind the staging pools."""
import pytest
import torch

pytestmark = pytest.mark.gpu

ARCH = {"num_layers": 3, "hidden_channels": 64, "dropout": 0.1, "drop_input": False,
        "batch_norm": False, "residual": False}
PARAMS = {"num_parts": 16, "max_steps": -1, "pool_size": 2, "num_workers": 0, "lr": 0.01,
          "reg_weight_decay": 0.0, "nonreg_weight_decay": 0.0, "grad_norm": 1.0, "epochs": 5}


def _conf(model, vr, arch_extra=None, **top):
    params = dict(PARAMS, architecture=dict(ARCH, **(arch_extra or {})))
    conf = {"seed": 3, "dataset": {"name": "arxiv"}, "synthetic_scale": 0.05, "batch_size": 4,
            "VR_update": vr, "model": {"name": model, "norm": True, "loop": True,
                                       "params": {"arxiv": params}}}
    conf.update(top)
    return conf


@pytest.mark.parametrize("model,vr,extra", [
    ("GCN", False, None), ("GCN", True, None),
    ("GCN2", True, {"alpha": 0.1, "theta": 0.5, "shared_weights": False}),
    ("GraphSAGE", False, None),
    ("APPNP", True, {"alpha": 0.1}),
])
def test_driver_learns_planted_labels(model, vr, extra):
    from torch_geometric_autoscale import train, _lib
    assert torch.cuda.is_available()
    conf = _conf(model, vr, dict(extra or {}))
    if model == "APPNP":    # its constructor takes only these (models/appnp.py:20-23)
        conf["model"]["params"]["arxiv"]["architecture"] = {
            "num_layers": 3, "hidden_channels": 64, "alpha": 0.1, "dropout": 0.1}
    res = train.run(conf)
    _lib.check_device_errors()
    hist = res["history"]
    assert hist[-1]["loss"] < hist[0]["loss"]
    assert res["val"] > 0.3, res            # 16 planted classes: chance is 1/16


def test_driver_with_host_tables_and_staging_pools():
    from torch_geometric_autoscale import train, _lib
    res = train.run(_conf("GCN2", False, {"alpha": 0.1, "theta": 0.5, "shared_weights": True},
                          history_device="host"))
    _lib.check_device_errors()
    assert res["history"][-1]["loss"] < res["history"][0]["loss"]


def test_driver_with_host_and_device_batch_builders():
    """Both batch builders drive the same training loop (batch order differs: the loaders
    shuffle with different generators), and both runs learn."""
    from torch_geometric_autoscale import train
    for builder in ("host", "device"):
        conf = _conf("GCN2", False, {"alpha": 0.1, "theta": 0.5, "shared_weights": False,
                                     "dropout": 0.0}, batch_builder=builder)
        res = train.run(conf)
        assert res["history"][-1]["loss"] < res["history"][0]["loss"], builder
        assert res["val"] > 0.3, (builder, res["val"])


@pytest.mark.parametrize("model,extra", [
    ("GCN2", {"alpha": 0.1, "theta": 0.5, "shared_weights": False}),
    ("GraphSAGE", None),
])
def test_driver_with_neighbour_sampling_and_edge_dropout(model, extra):
    """``num_neighbors`` and ``edge_dropout`` (loader.py:217-246, main.py:68) on the device
    batch builder: batches are sampled on the GPU at every visit, their weights dropped at every
    step, the steps are issued eagerly (nothing about such a batch is static), and the run
    still learns the planted labels."""
    from torch_geometric_autoscale import train, _lib
    conf = _conf(model, False, dict(extra or {}), num_neighbors=8, batch_builder="device")
    conf["model"]["params"]["arxiv"]["edge_dropout"] = 0.2
    res = train.run(conf)
    _lib.check_device_errors()
    assert res["graphs"]["train_steps_recorded"] == 0
    assert res["history"][-1]["loss"] < res["history"][0]["loss"]
    assert res["val"] > 0.3, res["val"]


@pytest.mark.parametrize("model,vr,extra,kind", [
    ("GCN2", True, {"alpha": 0.1, "theta": 0.5, "shared_weights": False, "dropout": 0.0}, "gcn2"),
    ("GCN2", False, {"alpha": 0.1, "theta": 0.5, "shared_weights": False, "dropout": 0.0}, "gcn2"),
    ("GCN", True, {"dropout": 0.0}, "gcn"),
    ("GraphSAGE", False, {"dropout": 0.0}, "sage"),
    # the headline architecture: 8 layers x 256, a weight pair per layer (fused layer kernel,
    # wide x_0 GEMM, 8-slab weight gradients), both step kinds
    ("GCN2", False, {"alpha": 0.1, "theta": 0.5, "shared_weights": False, "dropout": 0.0,
                     "num_layers": 8, "hidden_channels": 256}, "gcn2"),
    ("GCN2", True, {"alpha": 0.1, "theta": 0.5, "shared_weights": False, "dropout": 0.0,
                    "num_layers": 8, "hidden_channels": 256}, "gcn2"),
])
def test_driver_ends_with_exact_tables_and_outputs(model, vr, extra, kind):
    """After the driver's last refresh the tables hold the full-graph layer inputs (and, in the
    incremental mode, their aggregates) of the TRAINED weights and the refresh output is the
    full-batch output: checked against the numpy model oracle at 1e-4, i.e. training steps
    (with activations folded into their consumers: no dropout here), table pushes and the
    layer-wise sweep leave the state the reference's loop defines (main.py:211-215,233-234)."""
    import numpy as np
    import model_cases as mc
    from oracle import model_oracle as mo
    from torch_geometric_autoscale import train, _lib
    conf = _conf(model, vr, dict(extra))
    conf["model"]["params"]["arxiv"]["epochs"] = 2
    res = train.run(conf)
    _lib.check_device_errors()
    net, data = res["model"], res["data"]
    cfg = dict(ARCH, **extra)
    rowptr, col, val = (t.numpy() for t in data.adj_t.csr())
    inputs, aggs, want = mo.full_graph(kind, mc.numpy_state(net), cfg, rowptr, col, val,
                                       data.x.cpu().numpy())
    np.testing.assert_allclose(res["out"].numpy(), want, **mc.TOL, err_msg="refresh output")
    first = 0 if vr else 1          # (the plain sweep does not keep the layer-0 inputs)
    for i in range(first, net.num_layers):
        table = net.histories[i]
        np.testing.assert_allclose(table.emb.cpu().numpy(),
                                   mo.padded(inputs[i], table.embedding_dim), **mc.TOL,
                                   err_msg=f"M_in of layer {i}")
        if vr:
            table = net.histories_ag[i]
            np.testing.assert_allclose(table.emb.cpu().numpy(),
                                       mo.padded(aggs[i], table.embedding_dim), **mc.TOL,
                                       err_msg=f"M_ag of layer {i}")
