"""Whole-step A/B of functions.TN_FORM_POLICY (which weight-gradient kernel: 8-wave SHARED / 12-wave EXCLUSIVE), interleaved, medians, production library."""
import os, sys, time, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
import train_vit as TV
from vitamd import functions as F
dev = torch.device("cuda")
torch.manual_seed(0)
model = TV.ViTClassifier(TV.ViTConfig(224, 3, 16, "B", 1, 0.0)).to(dev)
x = torch.randn(256, 3, 224, 224, device=dev); y = torch.randint(0, 1000, (256,), device=dev)
def step():
    model.zero_grad(set_to_none=True); F.WEIGHTS.clear(); torch.nn.functional.cross_entropy(model(x), y).backward()
def timed(n=6):
    step(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for _ in range(3): step()
pol = ("auto", "shared", "exclusive")
res = {p: [] for p in pol}
for r in range(5):
    for p in pol:
        F.TN_FORM_POLICY = p; res[p].append(timed())
F.TN_FORM_POLICY = "auto"
for p in pol: print(f"TN_FORM_POLICY = {p:10s} median {statistics.median(res[p]):.2f} ms/step  {['%.2f' % q for q in res[p]]}", flush=True)
