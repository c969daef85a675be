"""dgrad-fc2 on the seam kernel (M = 50432, K = 768, N = 3072, stored-derivative multiply + fc1 bias-gradient column sums): what the factor loads and the
column sums cost (timing-only ablation bits 2 and 3 of experimental builds), beside the plain-bias epilogue of the same GEMM."""
import os, sys, statistics, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import ops, lib
lib.use_experimental(); L = lib.load(); L.vitamd_set_debug.argtypes = [ctypes.c_int]
dev = torch.device("cuda")
M, D = 256 * 197, 768
g = torch.Generator(device="cpu").manual_seed(0)
dy = torch.randn(M, D, generator=g).to(dev, torch.bfloat16); w = (torch.randn(4 * D, D, generator=g) * 0.03).to(dev, torch.bfloat16)
fac = torch.randn(M, 4 * D, generator=g).to(dev, torch.bfloat16); cs = torch.zeros(4 * D, device=dev); bias = torch.randn(4 * D, device=dev)
o1 = torch.empty(M, 4 * D, device=dev, dtype=torch.bfloat16)
def t(fn, n=10):
    for _ in range(2): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
rows = {}
for name, bits, fn in (("bias only (256-row seam)", 0, lambda: ops.gemm_nt(dy, w, ops.EPI_BIAS_BF16, bias=bias, out=o1, tile=24)),
                       ("dmul + colsum", 0, lambda: ops.gemm_nt(dy, w, ops.EPI_DMUL, aux=fac, colsum=cs, out=o1, tile=24)),
                       ("no factor loads", 4, lambda: ops.gemm_nt(dy, w, ops.EPI_DMUL, aux=fac, colsum=cs, out=o1, tile=24)),
                       ("column sums without the atomic", 16, lambda: ops.gemm_nt(dy, w, ops.EPI_DMUL, aux=fac, colsum=cs, out=o1, tile=24)),
                       ("no column sums", 8, lambda: ops.gemm_nt(dy, w, ops.EPI_DMUL, aux=fac, colsum=cs, out=o1, tile=24)),
                       ("neither", 12, lambda: ops.gemm_nt(dy, w, ops.EPI_DMUL, aux=fac, colsum=cs, out=o1, tile=24))):
    L.vitamd_set_debug(bits); rows[name] = statistics.median(t(fn) for _ in range(5)); L.vitamd_set_debug(0)
print("   ".join(f"{k} {v:6.1f} us" for k, v in rows.items()), flush=True)
