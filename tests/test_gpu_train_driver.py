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
