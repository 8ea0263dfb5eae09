"""The training driver (semantics of the reference's main.py:46-264) on a small synthetic graph,
host orchestration only (kernels swapped for the CPU oracle by tests/cpu_shim.py)."""
import pytest
import torch

from cpu_shim import cpu_kernels

PARAMS = {
    "architecture": {"num_layers": 2, "hidden_channels": 32, "dropout": 0.0, "drop_input": False,
                     "batch_norm": False, "residual": False},
    "num_parts": 8, "max_steps": -1, "pool_size": 1, "num_workers": 0, "lr": 0.01,
    "reg_weight_decay": 0.0, "nonreg_weight_decay": 0.0, "grad_norm": None, "epochs": 4,
}


def _conf(vr, model="GCN", extra=None):
    conf = {"seed": 1, "dataset": {"name": "arxiv"}, "synthetic_scale": 0.015, "batch_size": 2,
            "VR_update": vr, "model": {"name": model, "norm": True, "loop": True,
                                       "params": {"arxiv": dict(PARAMS, **(extra or {}))}}}
    return conf


@pytest.mark.parametrize("vr", [False, True])
def test_driver_trains_and_scores(vr):
    from torch_geometric_autoscale import train
    with cpu_kernels():
        res = train.run(_conf(vr))
    hist = res["history"]
    assert len(hist) == 4 and res["max_steps"] == 4          # num_parts / batch_size
    assert hist[-1]["loss"] < hist[0]["loss"]
    assert hist[-1]["train"] > 0.5                             # 8 planted classes: chance is 1/8
    assert 0.0 < res["val"] <= 1.0 and 0.0 <= res["test"] <= 1.0


def test_driver_gcn2_with_weight_decay_groups_and_max_steps():
    from torch_geometric_autoscale import train
    extra = {"architecture": dict(PARAMS["architecture"], alpha=0.1, theta=0.5, shared_weights=False),
             "max_steps": 1, "epochs": 2, "reg_weight_decay": 5e-4, "grad_norm": 1.0}
    with cpu_kernels():
        res = train.run(_conf(False, "GCN2", extra))
    assert res["max_steps"] == 1 and len(res["history"]) == 2


def test_conf_files_and_overrides(tmp_path):
    from torch_geometric_autoscale import train
    (tmp_path / "model").mkdir()
    (tmp_path / "config.yaml").write_text(
        "defaults:\n  - model: gcn\nseed: 7\ndevice: 0\nVR_update: false\nbatch_size: 3\nlog_every: 5\n")
    (tmp_path / "model" / "gcn.yaml").write_text(
        "name: GCN\nnorm: true\nloop: true\nparams:\n  arxiv:\n    architecture:\n      num_layers: 3\n"
        "      hidden_channels: 64\n    num_parts: 40\n    lr: 0.005\n    grad_norm: none\n")
    conf = train.load_conf(str(tmp_path), "gcn", "arxiv")
    assert conf["seed"] == 7 and conf["batch_size"] == 3 and "defaults" not in conf
    assert conf["model"]["params"]["arxiv"]["architecture"]["hidden_channels"] == 64
    assert conf["dataset"]["name"] == "arxiv" and conf["drift_norm"] == 2      # default kept
    conf = train.apply_overrides(conf, ["VR_update=true", "model.params.arxiv.lr=0.1",
                                        "model.params.arxiv.architecture.dropout=0.5", "dropout=0.25"])
    assert conf["VR_update"] is True and conf["model"]["params"]["arxiv"]["lr"] == 0.1
    assert conf["model"]["params"]["arxiv"]["architecture"]["dropout"] == 0.5 and conf["dropout"] == 0.25


def test_npz_dataset_round_trip(tmp_path):
    import numpy as np
    import torch_geometric_autoscale as tga
    adj, _ = tga.synthetic_graph(200, 1600, 4, seed=0)
    rowptr, col, _ = adj.csr()
    rng = np.random.default_rng(0)
    x = rng.standard_normal((200, 6)).astype(np.float32)
    y = rng.integers(0, 3, 200)
    np.savez(tmp_path / "tiny.npz", rowptr=rowptr.numpy(), col=col.numpy(), x=x, y=y,
             train_mask=rng.random(200) < 0.5)
    data, f_in, n_cls = tga.get_data(str(tmp_path), "tiny")
    assert f_in == 6 and n_cls == 3 and data.num_nodes == 200
    assert torch.equal(data.adj_t.csr()[1], col) and data.train_mask.dtype == torch.bool
    with pytest.raises(NotImplementedError):
        tga.get_data(str(tmp_path), "absent")
