"""The weight-gradient GEMM with its split-K reduce pass at the step's split factors (the reduce pass alone is what differs between
splits=1 and splits=4 beyond the GEMM's own time): quick timing of gemm_tn on the three ViT-B shapes."""
import os, sys, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import ops, functions as F
dev = torch.device("cuda")
R = 256 * 197
g = torch.Generator(device="cpu").manual_seed(3)
for name, P, Q in (("dWqkv", 2304, 768), ("dW1", 3072, 768), ("dW2", 768, 3072)):
    l = torch.randn(R, P, generator=g).to(dev, torch.bfloat16); r = torch.randn(R, Q, generator=g).to(dev, torch.bfloat16)
    out = torch.empty(P, Q, device=dev)
    sp = F._tn_splits(out)
    def t(n=10):
        ops.gemm_tn(l, r, out, accumulate=False, splits=sp); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for _ in range(n): ops.gemm_tn(l, r, out, accumulate=False, splits=sp)
        e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
    print(f"{name}: splits {sp}: {statistics.median(t() for _ in range(5)):.1f} us (GEMM + reduce pass)")
