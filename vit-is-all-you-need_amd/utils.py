"""Drop-in for the reference's `utils` module (reference utils.py:1-9)."""
import math

import torch


def get_params_str(m):
    return f"{sum(p.numel() for p in m.parameters()) / 1e6:.1f}M"


def lr_factor(step, warmup_steps, train_steps, min_ratio):
    """Multiplier on the base LR at optimiser step `step`, reproducing the reference schedule
    (utils.py:5-9: SequentialLR[linear warmup, CosineAnnealingLR(T_max=train_steps), constant] with
    milestones [warmup_steps, train_steps]) including its two quirks: the cosine phase starts
    counting at the warmup milestone, and from `train_steps` on the rate returns to the base LR."""
    if step < warmup_steps:
        return min(1.0, step / warmup_steps)
    if step < train_steps:
        t = step - warmup_steps
        return min_ratio + (1.0 - min_ratio) * (1.0 + math.cos(math.pi * t / train_steps)) / 2.0
    return 1.0


def get_lr_scheduler(optim, warmup_steps, train_steps, min_lr):
    """Same call signature as the reference; one closed-form LambdaLR instead of three chained
    schedulers (per-group base LR is respected)."""
    base = [g["lr"] for g in optim.param_groups]
    lambdas = [(lambda s, b=b: lr_factor(s, warmup_steps, train_steps, (min_lr / b) if b else 0.0)) for b in base]
    return torch.optim.lr_scheduler.LambdaLR(optim, lambdas)
