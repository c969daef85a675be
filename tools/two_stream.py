"""Timing experiment: does running the two halves of the batch on two HIP streams (so that one half's HBM-bound GEMM
epilogues overlap the other half's MFMA main loops) beat one full-batch pass?  Timing only - the halves share the
split-K workspace and gradient buffers, so the gradients of this script are not meaningful."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
import train_vit as TV
from vitamd import functions as F
side = int(os.environ.get("SIDE", "1"))
F.SIDE.enabled = bool(side)
dev = torch.device("cuda")
torch.manual_seed(0)
model = TV.ViTClassifier(TV.ViTConfig(224, 3, 16, "B", 1, 0.0)).to(dev)
B = 256
x = torch.randn(B, 3, 224, 224, device=dev); y = torch.randint(0, 1000, (B,), device=dev)
ce = torch.nn.functional.cross_entropy

def step_full():
    model.zero_grad(set_to_none=True); F.WEIGHTS.clear()
    ce(model(x), y).backward()

def make_split(n):
    streams = [torch.cuda.Stream() for _ in range(n)]
    xs, ys = x.chunk(n), y.chunk(n)
    def step():
        model.zero_grad(set_to_none=True); F.WEIGHTS.clear()
        cur = torch.cuda.current_stream()
        losses = []
        for s, xi, yi in zip(streams, xs, ys):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                losses.append(ce(model(xi), yi))
        for s, l in zip(streams, losses):
            with torch.cuda.stream(s):
                l.backward()
        for s in streams:
            cur.wait_stream(s)
    return step

def timeit(fn, name, n=6):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); print(f"{name}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms/step", flush=True)

timeit(step_full, f"full batch 256, one stream (side={side})")
for n in (2, 4):
    try:
        timeit(make_split(n), f"{n} x {B // n} on {n} streams (side={side})")
    except Exception as e:
        print("split", n, "failed:", repr(e)[:300])
timeit(step_full, "full again")
