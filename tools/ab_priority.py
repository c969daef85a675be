"""Does stream priority help?  The step on a HIGH-priority stream (input-gradient chain = critical path) with the weight-gradient
side stream at default (low) priority, against everything at default priority."""
import os, sys, time, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
import train_vit as TV
from vitamd import functions as F
dev = torch.device("cuda")
torch.manual_seed(0)
print("priority range:", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")
model = TV.ViTClassifier(TV.ViTConfig(224, 3, 16, "B", 1, 0.0)).to(dev)
x = torch.randn(256, 3, 224, 224, device=dev); y = torch.randint(0, 1000, (256,), device=dev)
def step():
    model.zero_grad(set_to_none=True); F.WEIGHTS.clear()
    torch.nn.functional.cross_entropy(model(x), y).backward()
def timed(stream, n=5):
    with torch.cuda.stream(stream):
        step(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): step()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
hi = torch.cuda.Stream(priority=-1)
lo_side = torch.cuda.Stream(priority=0)
default = torch.cuda.current_stream()
for _ in range(3): step()
cfgs = {"all default": (default, None), "main high / side low": (hi, lo_side), "main default / side HIGH": (default, torch.cuda.Stream(priority=-1))}
res = {k: [] for k in cfgs}
orig = dict(F.SIDE._streams)
for r in range(4):
    for k, (main, side) in cfgs.items():
        F.SIDE._streams = {dev: side} if side is not None else dict(orig)
        if side is not None: F.SIDE._streams = {torch.device("cuda", torch.cuda.current_device()): side}
        res[k].append(timed(main))
for k in cfgs: print("%-28s median %.2f ms/step  %s" % (k, statistics.median(res[k]), ["%.2f" % v for v in res[k]]))
