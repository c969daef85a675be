"""Schedule sweep of the NT GEMM (plain bias epilogue) on the four ViT-B (N, K) shapes.  usage: bench_nt_bias.py name=tile ..."""
import os, sys, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import ops
import ctypes
from vitamd import lib
lib.use_experimental(); L = lib.load(); L.vitamd_set_debug.argtypes = [ctypes.c_int]
cfgs = {}
for a in sys.argv[1:]:                      # name=tile[:dbgbits]
    k, v = a.split("=")
    t, _, d = v.partition(":")
    cfgs[k] = (int(t, 0), int(d, 0) if d else 0)
dev = torch.device("cuda")
M = 256 * 197
g = torch.Generator(device="cpu").manual_seed(1)
tot = {k: 0.0 for k in cfgs}
for (N, K) in [(2304, 768), (3072, 768), (768, 2304), (768, 3072)]:
    a = torch.randn(M, K, generator=g).to(dev, torch.bfloat16)
    b = (torch.randn(N, K, generator=g) * 0.03).to(dev, torch.bfloat16)
    ref = None
    res = {k: [] for k in cfgs}
    for rnd in range(5):
        for k, (t, dbg) in cfgs.items():
            L.vitamd_set_debug(dbg)
            out = ops.gemm_nt(a, b, ops.EPI_BIAS_BF16, tile=t)
            if rnd == 0:
                torch.cuda.synchronize()
                if ref is None: ref = out.clone()
                elif not torch.equal(out, ref): print(f"  N={N} K={K} {k}: MISMATCH", flush=True)
            del out
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); s.record()
            for _ in range(10): ops.gemm_nt(a, b, ops.EPI_BIAS_BF16, tile=t)
            e.record(); torch.cuda.synchronize()
            res[k].append(s.elapsed_time(e) / 10 * 1e3)
    L.vitamd_set_debug(0)
    fl = 2.0 * M * N * K
    for k in cfgs:
        med = statistics.median(res[k]); tot[k] += med
        print(f"N={N:5d} K={K:5d} {k:14s} {med:7.1f} us  {fl / med / 1e6:7.1f} TF", flush=True)
print({k: round(v, 1) for k, v in tot.items()})
