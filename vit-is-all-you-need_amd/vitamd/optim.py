"""Fused AdamW on the HIP path: same constructor arguments and update
rule as the reference's `torch.optim.AdamW` (train_vit.py:82), one kernel per parameter tensor,
state kept in fp32.  Works with `utils.get_lr_scheduler` (it is a torch.optim.Optimizer)."""
from __future__ import annotations

import torch

from . import lib as _lib


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.load()
        stream = torch.cuda.current_stream().cuda_stream
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise _lib.VitamdError("AdamW: parameters must be contiguous fp32 ROCm device tensors")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                code = L.vitamd_adamw_step(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                           p.numel(), float(group["lr"]), b1, b2, group["eps"], group["weight_decay"], st["step"],
                                           stream)
                _lib.check(code, "adamw_step")
        from .functions import WEIGHTS
        WEIGHTS.clear()   # the kernel updated the weights behind torch's version counters: drop the bf16 copies
        return loss
