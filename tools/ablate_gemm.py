"""Timing-only ablations of the 256x256 pipe GEMM: which resource bounds the main loop?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import ops
from vitamd import lib as _explib; _explib.use_experimental()
dev = torch.device("cuda")
M, D = 256 * 197, 768
g = torch.Generator(device="cpu").manual_seed(0)
rb = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dev, torch.bfloat16)
x768, x3072 = rb(M, D), rb(M, 4 * D)
wqkv, w2 = rb(3 * D, D, scale=0.03), rb(D, 4 * D, scale=0.03)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
w1 = rb(4 * D, D, scale=0.03)
for name, a, b in (("K=768 N=2304", x768, wqkv), ("K=768 N=3072", x768, w1), ("K=3072 N=768", x3072, w2)):
    for tile, what in ((2, "pipe"), (6, "persistent"), (2, "pipe"), (6, "persistent")):
        print(f"{name:14s} {what:18s} {t(lambda: ops.gemm_nt(a, b, ops.EPI_BIAS_BF16, tile=tile)):8.1f} us", flush=True)
