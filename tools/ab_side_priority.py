"""Whole-step A/B of the HIP stream priority of the weight-gradient side stream (and of the main stream): a higher-priority stream's workgroups are dispatched
first whenever CUs free up.  Interleaved, medians; production library."""
import os, sys, time, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
import train_vit as TV
from vitamd import functions as F
dev = torch.device("cuda", torch.cuda.current_device())
print("priority range (low, high):", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a", flush=True)
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
F.claim_streams(dev)
for _ in range(3): step()
streams = {"side default": (None, None), "side high": (-1, None), "main high": (None, -1), "side low (main default)": (0, None)}
made = {}
for k, (ps, pm) in streams.items():
    made[k] = (torch.cuda.Stream(device=dev, priority=ps) if ps is not None else F.SIDE._streams[dev], torch.cuda.Stream(device=dev, priority=pm) if pm is not None else None)
res = {k: [] for k in streams}
for r in range(5):
    for k, (s_side, s_main) in made.items():
        F.SIDE._streams[dev] = s_side
        if s_main is not None:
            with torch.cuda.stream(s_main): res[k].append(timed())
        else:
            res[k].append(timed())
for k in streams: print("%-26s median %.2f ms/step  %s" % (k, statistics.median(res[k]), ["%.2f" % v for v in res[k]]), flush=True)
