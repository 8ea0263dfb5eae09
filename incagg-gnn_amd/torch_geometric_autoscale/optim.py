"""``Adam`` with the update of ``torch.optim.Adam`` (the optimizer of main.py:160-165) as ONE
launch over all parameters of a group (incagg_adam_step_f32).

torch's fused multi-tensor kernel walks the tensors in chunks of 65 536 elements: the ~1 M
parameters of the models here give it 20 workgroups (43 us per step on MI355X, plus a launch that
advances the step counters); this one gives every 1 024 elements a workgroup.  The step counts
(one per parameter, as in torch) live on the device and are advanced there, so a step recorded
into a HIP graph keeps counting when it is replayed.
"""
import ctypes
from typing import Iterable, Union

import torch
from torch import Tensor

from . import _lib, ops


class Adam:
    """Subset of ``torch.optim.Adam``: ``params`` (tensors or group dicts with their own ``lr`` /
    ``weight_decay``, as main.py:160-165 builds them), ``lr``, ``betas``, ``eps``,
    ``weight_decay`` (added to the gradient, not decoupled).  fp32 parameters on one GPU.

    ``step()`` reads ``p.grad`` of every parameter; parameters without a gradient are skipped.
    Inside a recorded step the gradients must live at the same addresses in every replay (they
    do when the backward pass is part of the recording)."""

    def __init__(self, params: Iterable[Union[Tensor, dict]], lr: float = 1e-3,
                 betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        params = list(params)
        if not params:
            raise ValueError("optimizer got an empty parameter list")
        if not isinstance(params[0], dict):
            params = [{"params": params}]
        self.defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
        self.param_groups = []
        dev = None
        for g in params:
            group = dict(self.defaults)
            group.update(g)
            group["params"] = list(group["params"])
            for p in group["params"]:
                if p.dtype != torch.float32 or not p.is_cuda:
                    raise TypeError("incagg Adam updates float32 parameters on the GPU")
                if not p.is_contiguous():
                    raise ValueError("parameters must be contiguous")
                dev = p.device if dev is None else dev
                if p.device != dev:
                    raise ValueError("all parameters must be on one device")
            self.param_groups.append(group)
        self.device = dev
        # moments of a group in one allocation, every tensor's part 256-byte aligned
        self.state = {}
        for group in self.param_groups:
            sizes = [-(-p.numel() // 64) * 64 for p in group["params"]]
            flat = torch.zeros(2 * sum(sizes), dtype=torch.float32, device=dev)
            steps = torch.zeros(len(sizes), dtype=torch.float32, device=dev)
            off = 0
            for i, (p, n) in enumerate(zip(group["params"], sizes)):
                self.state[p] = {"step": steps[i],
                                 "exp_avg": flat[off:off + p.numel()].view_as(p),
                                 "exp_avg_sq": flat[sum(sizes) + off:sum(sizes) + off + p.numel()]
                                 .view_as(p)}
                off += n

    def zero_grad(self, set_to_none: bool = True) -> None:
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.detach_().zero_()

    @torch.no_grad()
    def step(self) -> None:
        L = _lib.lib()
        for group in self.param_groups:
            with_grad = [p for p in group["params"] if p.grad is not None]
            if not with_grad:
                continue
            segs = (_lib.AdamSegment * len(with_grad))()
            keep = []
            for s, p in zip(segs, with_grad):
                g = p.grad
                if g.dtype != torch.float32 or g.device != p.device:
                    raise TypeError("gradients must be float32 on the parameter's device")
                if not g.is_contiguous():
                    g = g.contiguous()
                    keep.append(g)
                st = self.state[p]
                s.p, s.g, s.m, s.v, s.step, s.n = (
                    p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(),
                    st["exp_avg_sq"].data_ptr(), st["step"].data_ptr(), p.numel())
            b1, b2 = group["betas"]
            with ops._on_device(self.device):
                _lib.check(L.incagg_adam_step_f32(
                    ctypes.addressof(segs), len(with_grad), group["lr"], b1, b2, group["eps"],
                    group["weight_decay"], ops._stream_of(self.device)))

    def state_dict(self) -> dict:
        return {"param_groups": [{k: (v.clone() if isinstance(v, Tensor) else v)
                                  for k, v in g.items() if k != "params"}
                                 for g in self.param_groups],
                "state": [[{k: v.clone() for k, v in self.state[p].items()} for p in g["params"]]
                          for g in self.param_groups]}

    def load_state_dict(self, sd: dict) -> None:
        for group, saved, moments in zip(self.param_groups, sd["param_groups"], sd["state"]):
            group.update(saved)
            for p, st in zip(group["params"], moments):
                for k, v in st.items():
                    self.state[p][k].copy_(v)
