"""fc1-shaped GEMM (K=768, N=3072) under different epilogues: how much is VALU, how much is stores?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import ops, lib
import ctypes
lib.use_experimental(); L = lib.load(); L.vitamd_set_debug.argtypes=[ctypes.c_int]
dev = torch.device("cuda")
M, D = 256 * 197, 768
g = torch.Generator(device="cpu").manual_seed(0)
rb = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dev, torch.bfloat16)
x, w = rb(M, D), rb(4 * D, D, scale=0.03)
bias = torch.randn(4 * D, device=dev)
res = torch.randn(M, 4 * D, device=dev)
pre = rb(M, 4 * D)
cs = torch.zeros(4 * D, device=dev)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for tile in (256, 2):
    print(f"tile {tile}")
    print(f"  bias->bf16  (1 x 310MB store)            {t(lambda: ops.gemm_nt(x, w, ops.EPI_BIAS_BF16, bias=bias, tile=tile)):7.1f} us")
    print(f"  f32 out     (1 x 620MB store)            {t(lambda: ops.gemm_nt(x, w, ops.EPI_F32, tile=tile)):7.1f} us")
    print(f"  resid f32   (620MB load + 620MB store)   {t(lambda: ops.gemm_nt(x, w, ops.EPI_RESID_F32, bias=bias, aux=res, tile=tile)):7.1f} us")
    print(f"  gelu        (2 x 310MB store + erf)      {t(lambda: ops.gemm_nt(x, w, ops.EPI_GELU, bias=bias, tile=tile)):7.1f} us")
    print(f"  dgelu       (310 load + 310 store + erf+exp) {t(lambda: ops.gemm_nt(x, w, ops.EPI_DGELU, aux=pre, colsum=cs, tile=tile)):7.1f} us")

for tile in (2,):
    for bits, what in ((1, "gelu: no erf math"), (2, "gelu: no 2nd store"), (3, "gelu: neither")):
        L.vitamd_set_debug(bits)
        print(f"  {what:28s} {t(lambda: ops.gemm_nt(x, w, ops.EPI_GELU, bias=bias, tile=tile)):7.1f} us")
    L.vitamd_set_debug(4)
    print(f"  resid: no load               {t(lambda: ops.gemm_nt(x, w, ops.EPI_RESID_F32, bias=bias, aux=res, tile=tile)):7.1f} us")
    L.vitamd_set_debug(0)

print("non-temporal output stores (dbg bit 3):")
for bits in (0, 8):
    L.vitamd_set_debug(bits)
    print(f"  dbg={bits}: bias {t(lambda: ops.gemm_nt(x, w, ops.EPI_BIAS_BF16, bias=bias, tile=2)):7.1f}  gelu {t(lambda: ops.gemm_nt(x, w, ops.EPI_GELU, bias=bias, tile=2)):7.1f}  dgelu {t(lambda: ops.gemm_nt(x, w, ops.EPI_DGELU, aux=pre, colsum=cs, tile=2)):7.1f} us")
xk = rb(M, 4 * D); wk = rb(D, 4 * D, scale=0.03); b1 = torch.randn(D, device=dev); r1 = torch.randn(M, D, device=dev)
for bits in (0, 8):
    L.vitamd_set_debug(bits)
    print(f"  dbg={bits}: fc2-shape plain {t(lambda: ops.gemm_nt(xk, wk, ops.EPI_BIAS_BF16, tile=2)):7.1f}  resid {t(lambda: ops.gemm_nt(xk, wk, ops.EPI_RESID_F32, bias=b1, aux=r1, tile=2)):7.1f} us")
L.vitamd_set_debug(0)

print("phase stagger of odd first-round workgroups (dbg bits 8..15, ~1 us units):")
for st in (0, 4, 8, 12, 16, 24):
    L.vitamd_set_debug(st << 8)
    print(f"  stagger {st:2d}: qkv-like bias {t(lambda: ops.gemm_nt(x, w, ops.EPI_BIAS_BF16, bias=bias, tile=2)):7.1f}  gelu {t(lambda: ops.gemm_nt(x, w, ops.EPI_GELU, bias=bias, tile=2)):7.1f}  dgelu {t(lambda: ops.gemm_nt(x, w, ops.EPI_DGELU, aux=pre, colsum=cs, tile=2)):7.1f}  fc2 resid {t(lambda: ops.gemm_nt(xk, wk, ops.EPI_RESID_F32, bias=b1, aux=r1, tile=2)):7.1f} us")
L.vitamd_set_debug(0)
