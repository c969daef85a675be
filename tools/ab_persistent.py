"""Whole-step A/B: every large NT GEMM through the persistent (static strided tile list) ping-pong kernel of the experimental
library (tile code 20) against the production launch, side stream on / off.  Measured: -0.45 ms on one box, +-0.0 on another; a
dynamic form (per-XCD tile counters, work stealing) was built and measured +0.7 ms - neither kept."""
import os, sys, time, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import lib
lib.use_experimental()
import train_vit as TV
from vitamd import functions as F, ops
dev = torch.device("cuda")
torch.manual_seed(0)
model = TV.ViTClassifier(TV.ViTConfig(224, 3, 16, "B", 1, 0.0)).to(dev)
x = torch.randn(256, 3, 224, 224, device=dev); y = torch.randint(0, 1000, (256,), device=dev)
orig = ops.gemm_nt
MODE = {"tile": None}       # (code for the shapes the production rule gives 320-row tiles, code for the 256-row ones)
def patched(a, b, epi, **kw):
    if MODE["tile"] and kw.get("tile", 0) == 0 and epi in (ops.EPI_BIAS_BF16, ops.EPI_RESID_F32, ops.EPI_GELU_DG, ops.EPI_DMUL) and a.shape[0] >= 20000 and a.shape[1] % 64 == 0:
        kw["tile"] = MODE["tile"][1] if b.shape[0] == 2304 else MODE["tile"][0]
    return orig(a, b, epi, **kw)
ops.gemm_nt = patched
def step():
    model.zero_grad(set_to_none=True); F.WEIGHTS.clear(); torch.nn.functional.cross_entropy(model(x), y).backward()
def timed(n=6):
    step(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for _ in range(3): step()
for side in (True, False):
    F.SIDE.enabled = side
    modes = {"production launch": None, "persistent, static lists, 320 rows everywhere": (20, 20)}
    res = {k: [] for k in modes}
    for r in range(4):
        for k, t in modes.items():
            MODE["tile"] = t; res[k].append(timed())
    for k, v in res.items():
        print(f"side stream {'on ' if side else 'off'}  {k:46s} median {statistics.median(v):.2f} ms/step  {['%.2f' % q for q in v]}", flush=True)
