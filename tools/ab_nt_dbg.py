"""Isolated NT GEMM launches (the six per-layer shapes) under vitamd_set_debug knobs, interleaved.  usage: ab_nt_dbg.py name=bits ..."""
import os, sys, statistics, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import ops, lib
lib.use_experimental(); L = lib.load(); L.vitamd_set_debug.argtypes = [ctypes.c_int]
cfgs = {"production": 0}
for a in sys.argv[1:]:
    k, v = a.split("="); cfgs[k] = int(v, 0)
dev = torch.device("cuda")
M, D = 256 * 197, 768
g = torch.Generator(device="cpu").manual_seed(1)
def rb(*s, scale=1.0): return (torch.randn(*s, generator=g) * scale).to(dev, torch.bfloat16)
x1, x3, x4 = rb(M, D), rb(M, 3 * D), rb(M, 4 * D)
wqkv, w1, w2 = rb(3 * D, D, scale=0.03), rb(4 * D, D, scale=0.03), rb(D, 4 * D, scale=0.03)
wqkv_t, w1_t, w2_t = rb(D, 3 * D, scale=0.03), rb(D, 4 * D, scale=0.03), rb(4 * D, D, scale=0.03)
b3, b4, b1 = torch.randn(3 * D, device=dev), torch.randn(4 * D, device=dev), torch.randn(D, device=dev)
res_in = torch.randn(M, D, device=dev)
cs = torch.zeros(4 * D, device=dev)
calls = [
    ("qkv", lambda: ops.gemm_nt(x1, wqkv, ops.EPI_BIAS_BF16, bias=b3), 2.0 * M * D * 3 * D),
    ("fc1+gelu", lambda: ops.gemm_nt(x1, w1, ops.EPI_GELU_DG, bias=b4), 2.0 * M * D * 4 * D),
    ("fc2+resid", lambda: ops.gemm_nt(x4, w2, ops.EPI_RESID_F32, bias=b1, aux=res_in), 2.0 * M * D * 4 * D),
    ("dgrad_fc2", lambda: ops.gemm_nt(x1, w2_t, ops.EPI_DMUL, aux=x4, colsum=cs), 2.0 * M * D * 4 * D),
    ("dgrad_fc1", lambda: ops.gemm_nt(x4, w1_t, ops.EPI_BIAS_BF16), 2.0 * M * D * 4 * D),
    ("dgrad_qkv", lambda: ops.gemm_nt(x3, wqkv_t, ops.EPI_BIAS_BF16), 2.0 * M * D * 3 * D),
]
tot = {k: 0.0 for k in cfgs}
for name, fn, fl in calls:
    res = {k: [] for k in cfgs}
    for rnd in range(5):
        for k, bits in cfgs.items():
            L.vitamd_set_debug(bits)
            fn()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); s.record()
            for _ in range(10): fn()
            e.record(); torch.cuda.synchronize()
            res[k].append(s.elapsed_time(e) / 10 * 1e3)
    L.vitamd_set_debug(0)
    for k in cfgs:
        med = statistics.median(res[k]); tot[k] += med
        print(f"{name:10s} {k:16s} {med:7.1f} us  {fl / med / 1e6:7.1f} TF", flush=True)
print({k: round(v, 1) for k, v in tot.items()}, "us per layer")
