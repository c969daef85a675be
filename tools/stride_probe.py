"""Does the operand row stride (K*2 bytes) matter?  L2-channel camping probe."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import ops
from vitamd import lib as _explib; _explib.use_experimental()
dev = torch.device("cuda")
M = 256 * 197
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for N in (768, 2304):
    for K in (704, 768, 832, 2944, 3008, 3072, 3136, 3200):
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        b = (torch.randn(N, K, device=dev) * 0.03).to(torch.bfloat16)
        for tile in (2, 4):
            us = t(lambda: ops.gemm_nt(a, b, ops.EPI_BIAS_BF16, tile=tile))
            print(f"N={N:5d} K={K:5d} tile={tile}  {us:8.1f} us  {2.0*M*N*K/us/1e6:8.1f} TFLOP/s", flush=True)
