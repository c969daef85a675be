"""Soak: 300 optimiser steps of ViT-B/16 (batch 128) on one fixed synthetic batch through the HIP path + fused AdamW + the
reference LR schedule: the loss must fall (memorisation), stay finite, and device memory must not grow."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
import train_vit as TV, utils as U
from vitamd.optim import AdamW
dev = torch.device("cuda")
torch.manual_seed(0)
m = TV.ViTClassifier(TV.ViTConfig(224, 3, 16, "B", 1, 0.0), num_classes=1000).to(dev)
x = torch.randn(128, 3, 224, 224, device=dev); y = torch.randint(0, 1000, (128,), device=dev)
opt = AdamW(m.parameters(), lr=3e-4, weight_decay=0.05)
sched = U.get_lr_scheduler(opt, 20, 300, 1e-5)
losses, mem = [], []
t0 = time.perf_counter()
for i in range(300):
    losses.append(float(TV.train_step(m, x, y, opt, sched)))
    if i % 50 == 0 or i == 299:
        mem.append(torch.cuda.max_memory_allocated() / 2**30)
        print(f"step {i:3d} loss {losses[-1]:.4f} max-alloc {mem[-1]:.2f} GiB", flush=True)
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 300 * 1e3:.1f} ms/step incl. optimiser; finite={all(l == l and abs(l) < 1e4 for l in losses)}; "
      f"loss {losses[0]:.3f} -> {losses[-1]:.3f}; memory growth after step 50: {mem[-1] - mem[1]:.3f} GiB")
