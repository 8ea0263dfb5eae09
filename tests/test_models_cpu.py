"""Host orchestration of the four models (sweeps, push/pull forward, incremental forward,
one training step) with the HIP entry points swapped for the CPU oracle -- what runs here
is the Python control flow, the History bookkeeping and the batch construction; the
kernels themselves are covered by the `-m gpu` tests."""
import pytest
import torch

import model_cases as mc
from cpu_shim import cpu_kernels


@pytest.mark.parametrize("name", list(mc.MODEL_CASES))
def test_model_paths_on_cpu(name):
    data, ptr, n_cls = mc.case_graph(name)
    f_in = data.x.size(1)
    cpu = torch.device("cpu")
    with cpu_kernels():
        model = mc.make_model(name, data, f_in, n_cls, emb_device="cpu")
        loader, want = mc.check_refresh_sweep(name, model, data, ptr)
        mc.check_push_pull_forward(name, model, data, ptr, want, cpu)
        if name in mc.SMALL_CASES:
            mc.check_plain_sweep(name, model, data, ptr)
            mc.check_ablation_modes(name, model, data, ptr, cpu)
        mc.check_step_gradients(name, model, data, ptr, cpu, vr=False)
        mc.check_incagg_forward(name, model, data, ptr, cpu)
        mc.check_step_gradients(name, model, data, ptr, cpu, vr=True)
        mc.check_training_step(name, model, data, ptr, cpu, vr=False)
        mc.check_training_step(name, model, data, ptr, cpu, vr=True)


@pytest.mark.parametrize("name", ["gcn2", "sage", "appnp", "gcn"])
def test_sampled_and_edge_dropped_batches_on_cpu(name):
    """A training step on a neighbour-sampled batch (loader.py:217-246) and on an edge-dropped
    one (utils.py:62-73, main.py:68): host orchestration against the float64 step oracle."""
    data, ptr, n_cls = mc.case_graph(name)
    cpu = torch.device("cpu")
    with cpu_kernels():
        model = mc.make_model(name, data, data.x.size(1), n_cls, emb_device="cpu")
        mc.check_refresh_sweep(name, model, data, ptr)
        torch.manual_seed(11)
        mc.check_step_gradients(name, model, data, ptr, cpu, vr=False, num_neighbors=4)
        mc.check_step_gradients(name, model, data, ptr, cpu, vr=False, edge_dropout=0.3)
        mc.check_step_gradients(name, model, data, ptr, cpu, vr=True, edge_dropout=0.3)
        mc.check_step_gradients(name, model, data, ptr, cpu, vr=False, num_neighbors=6,
                                edge_dropout=0.2)
