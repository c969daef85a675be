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
cfgs = {"policy0_wgs252": (0, 252, True), "policy1_wgs252": (1, 252, True), "policy2_wgs252": (2, 252, True), "policy0_wgs192": (0, 192, True),
        "no side stream": (0, 252, False)}
if len(sys.argv) > 1:          # explicit list: policy:wgs ...
    cfgs = {"policy0_wgs252": (0, 252, True)}
    for a in sys.argv[1:]:
        pol, wgs = a.split(":"); cfgs[f"policy{pol}_wgs{wgs}"] = (int(pol), int(wgs), True)
for _ in range(3): step()
res = {k: [] for k in cfgs}
for r in range(5):
    for k, (pol, wgs, side) in cfgs.items():
        F.SIDE_POLICY, F.TN_TARGET_WGS, F.SIDE.enabled = pol, wgs, side
        res[k].append(timed())
for k in cfgs: print("%-18s median %.2f ms/step  %s" % (k, statistics.median(res[k]), ["%.2f" % v for v in res[k]]), flush=True)
