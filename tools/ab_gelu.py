"""A/B on the whole training step: gelu' evaluated in the fc1 epilogue and stored (production) vs evaluated in the
fc2-dgrad epilogue from the stored pre-activation.  Interleaved, medians."""
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
res = {True: [], False: []}
for r in range(6):
    for flag in (True, False):
        F.GELU_STORED_GRAD = flag; res[flag].append(timed())
print("stored gelu' (production): median %.2f ms/step  %s" % (statistics.median(res[True]), ["%.2f" % v for v in res[True]]))
print("gelu' in backward        : median %.2f ms/step  %s" % (statistics.median(res[False]), ["%.2f" % v for v in res[False]]))
