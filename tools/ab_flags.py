"""Whole-step A/B of boolean knobs of vitamd.functions (interleaved, medians).  usage: ab_flags.py NAME[=0|1] ..."""
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
flags = [a.split("=")[0] for a in sys.argv[1:]]
defaults = {f: getattr(F, f) for f in flags}
cfgs = {"defaults": {}}
for f in flags: cfgs[f"{f} = {not defaults[f]}"] = {f: not defaults[f]}
for _ in range(3): step()
res = {k: [] for k in cfgs}
for r in range(4):
    for k, over in cfgs.items():
        for f, v in defaults.items(): setattr(F, f, over.get(f, v))
        res[k].append(timed())
for f, v in defaults.items(): setattr(F, f, v)
for k, v in res.items(): print(f"{k:36s} median {statistics.median(v):.2f} ms/step  {['%.2f' % q for q in v]}", flush=True)
