"""A/B of the split-K factor (as a target workgroup count; 0 = the kernel's own rule) of the weight-gradient GEMMs on the whole
step: they overlap the dgrad chain on the side stream."""
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
for _ in range(3): step()
cfgs = [int(a) for a in sys.argv[1:]] or [72, 108, 128, 144, 180, 216, 252]
res = {k: [] for k in cfgs}
for r in range(4):
    for k in cfgs:
        F.TN_TARGET_WGS = k; res[k].append(timed())
F.TN_TARGET_WGS = None
for k in cfgs: print("target workgroups=%s median %.2f ms/step  %s" % (k, statistics.median(res[k]), ["%.2f" % v for v in res[k]]))
