"""fc1+GELU on the seam kernel (M = 50432, K = 768, N = 3072): what the erf/exp arithmetic and the second output cost (timing-only ablation bits
0 and 1 of experimental builds), beside the plain-bias epilogue of the same GEMM."""
import os, sys, statistics, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import ops, lib
lib.use_experimental(); L = lib.load(); L.vitamd_set_debug.argtypes = [ctypes.c_int]
dev = torch.device("cuda")
M, D = 256 * 197, 768
g = torch.Generator(device="cpu").manual_seed(0)
x = torch.randn(M, D, generator=g).to(dev, torch.bfloat16); w = (torch.randn(4 * D, D, generator=g) * 0.03).to(dev, torch.bfloat16)
bias = torch.randn(4 * D, device=dev)
o1 = torch.empty(M, 4 * D, device=dev, dtype=torch.bfloat16); o2 = torch.empty_like(o1)
def t(fn, n=10):
    for _ in range(2): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
tiles = [int(a) for a in sys.argv[1:]] or [24, 30]
for tile in tiles:
    rows = {}
    for name, bits, fn in (("bias only", 0, lambda: ops.gemm_nt(x, w, ops.EPI_BIAS_BF16, bias=bias, out=o1, tile=tile)),
                           ("gelu + gelu'", 0, lambda: ops.gemm_nt(x, w, ops.EPI_GELU_DG, bias=bias, out=o1, out2=o2, tile=tile)),
                           ("  no erf/exp", 1, lambda: ops.gemm_nt(x, w, ops.EPI_GELU_DG, bias=bias, out=o1, out2=o2, tile=tile)),
                           ("  no 2nd store", 2, lambda: ops.gemm_nt(x, w, ops.EPI_GELU_DG, bias=bias, out=o1, out2=o2, tile=tile)),
                           ("  neither", 3, lambda: ops.gemm_nt(x, w, ops.EPI_GELU_DG, bias=bias, out=o1, out2=o2, tile=tile))):
        L.vitamd_set_debug(bits); rows[name] = statistics.median(t(fn) for _ in range(5)); L.vitamd_set_debug(0)
    print(f"tile code {tile}: " + "   ".join(f"{k.strip()} {v:6.1f} us" for k, v in rows.items()), flush=True)
