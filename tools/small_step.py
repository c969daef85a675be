"""BASELINE configs[0] (ViT-S, 32x32, batch 64, 10 classes) on the GPU: launch-bound?  eager step time vs kernel time."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
import train_vit as TV
from vitamd import functions as F
dev = torch.device("cuda")
torch.manual_seed(0)
model = TV.ViTClassifier(TV.ViTConfig(32, 3, 16, "S", 1, 0.0), num_classes=10).to(dev)
x = torch.randn(64, 3, 32, 32, device=dev); y = torch.randint(0, 10, (64,), device=dev)
def step():
    model.zero_grad(set_to_none=True); F.WEIGHTS.clear()
    torch.nn.functional.cross_entropy(model(x), y).backward()
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize(); print("eager: %.3f ms/step" % ((time.perf_counter() - t0) / 50 * 1e3))
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
