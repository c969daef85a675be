"""Feasibility probe: two half-batch training steps on two CU-masked streams (bits 0-127 / 128-255 of the HIP CU mask = 16 CUs of
every XCD each) against one full-batch step on the whole chip.  Two model replicas, so no gradient sharing is involved."""
import ctypes, os, sys, time, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
import train_vit as TV
from vitamd import functions as F
hip = ctypes.CDLL("libamdhip64.so")
def masked_stream(words):
    s = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)
dev = torch.device("cuda")
torch.manual_seed(0)
cfg = TV.ViTConfig(224, 3, 16, "B", 1, 0.0)
full = TV.ViTClassifier(cfg).to(dev)
lanes = [TV.ViTClassifier(cfg).to(dev) for _ in range(2)]
x = torch.randn(256, 3, 224, 224, device=dev); y = torch.randint(0, 1000, (256,), device=dev)
xs, ys = x.chunk(2), y.chunk(2)
F1 = 0xffffffff
streams = {"halves": [masked_stream([F1] * 4 + [0] * 4), masked_stream([0] * 4 + [F1] * 4)],
           "unmasked2": [torch.cuda.Stream(), torch.cuda.Stream()]}
def step_full():
    full.zero_grad(set_to_none=True); F.WEIGHTS.clear()
    torch.nn.functional.cross_entropy(full(x), y).backward()
def step_lanes(ss):
    F.WEIGHTS.clear()
    for m, s, xi, yi in zip(lanes, ss, xs, ys):
        with torch.cuda.stream(s):
            m.zero_grad(set_to_none=True)
            torch.nn.functional.cross_entropy(m(xi), yi).backward()
def timed(fn, n=6):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for side in (False, True):
    F.SIDE.enabled = side
    res = {"full": [], "halves": [], "unmasked2": []}
    for _ in range(2): step_full(); step_lanes(streams["halves"]); step_lanes(streams["unmasked2"])
    torch.cuda.synchronize()
    for r in range(4):
        res["full"].append(timed(step_full))
        for k in ("halves", "unmasked2"):
            res[k].append(timed(lambda: step_lanes(streams[k])))
    for k, v in res.items():
        print(f"side={side} {k:10s} median {statistics.median(v):6.2f} ms per 256 images  {['%.2f' % t for t in v]}", flush=True)
