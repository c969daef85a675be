"""TN GEMM time against the split-K factor (= active workgroups): is the kernel bound per CU or by something chip-wide?"""
import os, sys, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import ops
dev = torch.device("cuda")
R = 256 * 197
g = torch.Generator(device="cpu").manual_seed(3)
for name, P, Q in [("dW1", 3072, 768), ("dWqkv", 2304, 768)]:
    l = torch.randn(R, P, generator=g).to(dev, torch.bfloat16)
    r = torch.randn(R, Q, generator=g).to(dev, torch.bfloat16)
    out = torch.empty(P, Q, device=dev)
    ntile = (P // 256) * (Q // 256)
    for splits in (1, 2, 3, 4, 5, 6, 7, 9, 14):
        ts = []
        for rnd in range(3):
            ops.gemm_tn(l, r, out, accumulate=False, splits=splits)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); s.record()
            for _ in range(10): ops.gemm_tn(l, r, out, accumulate=False, splits=splits)
            e.record(); torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) / 10 * 1e3)
        med = statistics.median(ts)
        print(f"{name} splits {splits:2d} -> {ntile * splits:4d} workgroups: {med:7.1f} us  {2.0 * R * P * Q / med / 1e6:7.1f} TF   per active CU {2.0 * R * P * Q / med / 1e6 / min(256, ntile * splits):5.2f} TF", flush=True)
