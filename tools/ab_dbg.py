"""A/B of vitamd_set_debug knobs on the whole training step (interleaved, medians).  usage: ab_dbg.py name=bits ...
(AB_NO_SIDE=1 in the environment: weight-gradient GEMMs on the main stream)"""
import os, sys, time, statistics, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
import train_vit as TV
from vitamd import functions as F, lib
lib.use_experimental(); L = lib.load(); L.vitamd_set_debug.argtypes = [ctypes.c_int]
if os.environ.get("AB_NO_SIDE") == "1": F.SIDE.enabled = False
cfgs = {"production": 0}
for a in sys.argv[1:]:
    k, v = a.split("="); cfgs[k] = int(v, 0)
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
res = {k: [] for k in cfgs}
for r in range(5):
    for k, bits in cfgs.items():
        L.vitamd_set_debug(bits); res[k].append(timed())
L.vitamd_set_debug(0)
for k in cfgs: print("%-24s median %.2f ms/step  %s" % (k, statistics.median(res[k]), ["%.2f" % v for v in res[k]]), flush=True)
