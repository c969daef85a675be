"""LayerNorm kernels at the headline shape (M = 50 432, D = 768), as the step calls them."""
import os, sys, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import ops, lib
if len(sys.argv) > 1 and sys.argv[1] == "sweep": lib.use_experimental()
dev = torch.device("cuda")
M, D = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (256 * 197, 768)     # usage: bench_ln.py [sweep|-] [M D]
g = torch.Generator(device="cpu").manual_seed(2)
xf = torch.randn(M, D, generator=g).to(dev); add = torch.randn(M, D, generator=g).to(dev, torch.bfloat16)
dy = torch.randn(M, D, generator=g).to(dev, torch.bfloat16); res = torch.randn(M, D, generator=g).to(dev)
_, y, mean, rstd = ops.layernorm_fwd(xf)
def t(fn, n=20):
    fn(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
cases = {"fwd (6 B/elem)": (lambda: ops.layernorm_fwd(xf), 6), "fwd + add (12 B/elem)": (lambda: ops.layernorm_fwd(xf, addend=add), 12),
         "bwd on bf16 xhat + residual + bf16 copy (14 B/elem)": (lambda: ops.layernorm_bwd(dy, xf, mean, rstd, g_res=res, want_bf16=True, xhat=y), 14),
         "same + column sums of the bf16 copy": (lambda: ops.layernorm_bwd(dy, xf, mean, rstd, g_res=res, want_bf16=True, xhat=y, colsum=cs), 14)}
cs = torch.zeros(D, device=dev)
for _ in range(1):
    for name, (fn, bpe) in cases.items():
        us = statistics.median(t(fn) for _ in range(5))
        print(f"{name:52s} {us:6.1f} us  {M * D * bpe / us / 1e6:.2f} TB/s")

if len(sys.argv) > 1 and sys.argv[1] == "sweep":      # experimental library: block cap of the column-sum form (dbg bits 24-31, units of 256 blocks)
    import ctypes
    from vitamd import lib
    L = lib.load(); L.vitamd_set_debug.argtypes = [ctypes.c_int]
    fn = cases["same + column sums of the bf16 copy"][0]
    for cap in (1, 2, 3, 4, 6, 8, 12, 16, 32, 64):
        L.vitamd_set_debug(cap << 24); us = statistics.median(t(fn) for _ in range(5)); L.vitamd_set_debug(0)
        print(f"column-sum form, at most {256 * cap:5d} blocks: {us:6.1f} us")
