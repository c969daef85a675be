"""A/B on the whole training step of the two dataflow choices made in vitamd/functions.py: gelu' stored by the fc1 epilogue vs
evaluated in the fc2-dgrad epilogue; LayerNorm backward reading xhat from the saved bf16 LN output vs recomputing it from x."""
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
cfgs = {"production": (True, True, False), "gelu' evaluated in backward": (False, True, False), "LN backward recomputes xhat from fp32 x": (True, False, False),
        "residual add fused into the attention forward": (True, True, True)}
res = {k: [] for k in cfgs}
for r in range(5):
    for k, (gs, lx, ar) in cfgs.items():
        F.GELU_STORED_GRAD, F.LN_BWD_XHAT, F.ATTN_FUSED_RESID = gs, lx, ar; res[k].append(timed())
F.GELU_STORED_GRAD, F.LN_BWD_XHAT, F.ATTN_FUSED_RESID = True, True, False
for k in cfgs: print("%-46s median %.2f ms/step  %s" % (k, statistics.median(res[k]), ["%.2f" % v for v in res[k]]))
