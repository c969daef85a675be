"""dgelu GEMM: how much of it is the gelu' VALU math?  interleaved A/B, medians."""
import os, sys, torch, ctypes, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import ops, lib
lib.use_experimental(); L = lib.load(); L.vitamd_set_debug.argtypes = [ctypes.c_int]
dev = torch.device("cuda")
M, D = 256 * 197, 768
g = torch.Generator(device="cpu").manual_seed(0)
rb = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dev, torch.bfloat16)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
x, w = rb(M, D), rb(4 * D, D, scale=0.03)
aux = rb(M, 4 * D); cs = torch.zeros(4 * D, device=dev); bias = torch.randn(4 * D, device=dev)
dg = lambda: ops.gemm_nt(x, w, ops.EPI_DGELU, aux=aux, colsum=cs, tile=2)
ge = lambda: ops.gemm_nt(x, w, ops.EPI_GELU, bias=bias, tile=2)
pl = lambda: ops.gemm_nt(x, w, ops.EPI_BIAS_BF16, bias=bias, tile=2)
cfg = {"dgelu": (dg, 0), "dgelu no-math": (dg, 0x40000), "dgelu no-stores": (dg, 0x10000), "dgelu no-math no-stores": (dg, 0x50000),
       "gelu": (ge, 0), "gelu no-math": (ge, 1), "gelu no-stores": (ge, 0x10000), "gelu no-math no-stores": (ge, 0x10001), "plain bias (1 store)": (pl, 0)}
for _ in range(20): dg()
res = {k: [] for k in cfg}
for r in range(7):
    for k, (fn, bits) in cfg.items():
        L.vitamd_set_debug(bits | (255 << 8)); res[k].append(t(fn))
L.vitamd_set_debug(0)
for k in cfg: print(f"{k:28s} {statistics.median(res[k]):6.1f} us", flush=True)
