"""The once-per-step batched weight cast (+ transpose) of ViT-B: one launch, 85 M parameters (340 MB read, 2 x 170 MB written)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd.functions import WeightCache
dev = torch.device("cuda")
D = 768
ws = [torch.randn(s, device=dev) for _ in range(12) for s in ((3 * D, D), (4 * D, D), (D, 4 * D))]
cache = WeightCache()
def run():
    cache.clear(); cache.prepare(ws, True)
run(); torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): run()
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) / 20 * 1e3
nbytes = sum(w.numel() for w in ws) * 8
print(f"batched cast + transpose of {len(ws)} weights: {us:.1f} us per launch = {nbytes / us / 1e6:.2f} TB/s")
