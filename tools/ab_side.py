"""Whole-step A/B of the side-stream policy knobs of vitamd.functions (interleaved, medians)."""
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
    model.zero_grad(set_to_none=True); F.WEIGHTS.clear()
    torch.nn.functional.cross_entropy(model(x), y).backward()
def timed(n=5):
    step(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
cfgs = {"p0_wgs180": (0, 180, True), "p0_wgs144": (0, 144, True), "p0_wgs128": (0, 128, True), "p0_wgs108": (0, 108, True), "p0_wgs96": (0, 96, True),
        "p0_wgs72": (0, 72, True), "p1_wgs96": (1, 96, True), "p1_wgs72": (1, 72, True)}
for _ in range(3): step()
res = {k: [] for k in cfgs}
for r in range(5):
    for k, (pol, wgs, side) in cfgs.items():
        F.SIDE_POLICY, F.TN_TARGET_WGS, F.SIDE.enabled = pol, wgs, side
        res[k].append(timed())
for k in cfgs: print("%-18s median %.2f ms/step  %s" % (k, statistics.median(res[k]), ["%.2f" % v for v in res[k]]), flush=True)
