"""hipGraph capture of the forward + loss + backward step (torch.cuda.CUDAGraph is hipGraph on ROCm).

New functionality - the reference has no graph capture.  Every libvitamd entry point only enqueues work on the
stream it is given (include/vitamd.h), so a whole training step - the per-step weight cast, ~25 kernels per
layer, the side-stream weight-gradient GEMMs with their event fork/join, the loss - records into one graph.
What it buys: the launch-bound configurations.  BASELINE configs[0] (ViT-S, 32x32, batch 64) spends 2.8 ms per
step in Python + launch overhead eagerly and 1.4 ms replayed; the headline ViT-B/16 batch-256 step is GPU-bound
(36 ms of kernels) and gains nothing.

    step = GraphedStep(model, torch.nn.functional.cross_entropy, x_example, y_example)
    for x, y in loader:
        loss = step(x, y)          # static tensor: read it before the next call
        optim.step()               # p.grad are the graph's static gradient tensors (re-attached every call)

Shapes and dtypes are frozen at capture.  Not for dropout > 0 (the mask seed is a host value baked into the
captured launches) and not under vitamd.ddp.DataParallel: its per-parameter hooks and the per-layer `layer_ready` calls are Python
that runs DURING backward and decides, from bucket state, what to enqueue (and torch.distributed work handles are waited for on the
host in finish()); none of that replays from a graph.  A DataParallel model is refused at construction.
"""
from __future__ import annotations

import torch

from . import functions as F


class GraphedStep:
    def __init__(self, model: torch.nn.Module, loss_fn, example_x: torch.Tensor, example_y: torch.Tensor, warmup: int = 3):
        if not example_x.is_cuda:
            raise F.ops._lib.VitamdError("GraphedStep needs device tensors (there is no CPU path)")
        for m in model.modules():
            if float(getattr(m, "dropout", 0.0) or 0.0) > 0.0:      # attention dropout is applied in eval() too (reference transformer.py:28)
                raise NotImplementedError("GraphedStep with dropout > 0: the mask seed would be frozen into the graph")
        from .ddp import DataParallel
        if isinstance(model, DataParallel) or any(isinstance(m, DataParallel) for m in model.modules()):
            raise NotImplementedError("GraphedStep around vitamd.ddp.DataParallel: the bucket hooks run Python during backward (see the module docstring)")
        self.model, self.loss_fn = model, loss_fn
        self.x = example_x.detach().clone()
        self.y = example_y.detach().clone()
        self.params = [p for p in model.parameters() if p.requires_grad]
        # warm up (allocator pools, hipFuncSetAttribute one-time calls, weight-cache groups) ON THE STREAM THE CAPTURE WILL USE: autograd's
        # AccumulateGrad nodes remember the stream they were created on, and nodes created on another stream than the capturing one
        # made every replayed backward warn about (and potentially synchronise on) a stream mismatch
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(max(1, warmup)):
                self._eager()
            model.zero_grad(set_to_none=True)      # so the captured backward ASSIGNS fresh gradient tensors
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=s):
            # detached: the static loss tensor must not keep the captured step's autograd graph (and with it the parameters'
            # AccumulateGrad nodes, bound to the capture stream) alive - an eager backward on another stream afterwards would warn
            self.loss = self._eager(zero=False).detach()
        self.grads = [p.grad for p in self.params]

    def _eager(self, zero=True):
        if zero:
            self.model.zero_grad(set_to_none=True)
        F.WEIGHTS.clear()                            # the weight casts are part of every step (the optimiser changes the weights)
        loss = self.loss_fn(self.model(self.x), self.y)
        loss.backward()
        return loss

    def __call__(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        if x.shape != self.x.shape or y.shape != self.y.shape or x.dtype != self.x.dtype or y.dtype != self.y.dtype:
            raise F.ops._lib.VitamdError(f"GraphedStep was captured for {tuple(self.x.shape)} / {tuple(self.y.shape)}")
        self.x.copy_(x, non_blocking=True)
        self.y.copy_(y, non_blocking=True)
        self.graph.replay()
        F.WEIGHTS.clear()                            # host-side cache state: whatever eager call comes next must re-cast
        for p, g in zip(self.params, self.grads):
            p.grad = g
        return self.loss
