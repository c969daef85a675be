"""NT GEMM variants on the six per-layer launches of ViT-B (M = 50 432): interleaved rounds in one process, random data, medians;
outputs compared with the first variant (same accumulation order -> bit-exact).
usage: bench_nt.py [name=tile ...]   tile: 0 auto (round-1 pipe kernel, 256/320-row tiles), 7/8/9 ping-pong LEAD 4/5/6"""
import os, sys, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import ops
from vitamd import lib as _explib; _explib.use_experimental()
cfgs = {"pp_auto": (0, 0), "pers": (20, 0), "pers_xcd2_u12": (20, 0x1200c00), "pers_xcd4_u6": (20, 0x1400600), "pers_xcd8_u3": (20, 0x1800300)}
import ctypes
from vitamd import lib as _l2
_L = _l2.load()
for a in sys.argv[1:]:                      # name=tile[:dbgbits]
    k, v = a.split("="); t, _, d = v.partition(":"); cfgs[k] = (int(t, 0), int(d, 0) if d else 0)
dev = torch.device("cuda")
M, D = 256 * 197, 768
g = torch.Generator(device="cpu").manual_seed(1)
def rb(*s, scale=1.0): return (torch.randn(*s, generator=g) * scale).to(dev, torch.bfloat16)
x1, x3, x4 = rb(M, D), rb(M, 3 * D), rb(M, 4 * D)
wqkv, w1, w2 = rb(3 * D, D, scale=0.03), rb(4 * D, D, scale=0.03), rb(D, 4 * D, scale=0.03)
wqkv_t, w1_t, w2_t = rb(D, 3 * D, scale=0.03), rb(D, 4 * D, scale=0.03), rb(4 * D, D, scale=0.03)
b3, b4, b1 = torch.randn(3 * D, device=dev), torch.randn(4 * D, device=dev), torch.randn(D, device=dev)
res_in = torch.randn(M, D, device=dev)
calls = [
    ("qkv", lambda t: ops.gemm_nt(x1, wqkv, ops.EPI_BIAS_BF16, bias=b3, tile=t), 2.0 * M * D * 3 * D),
    ("fc1+gelu", lambda t: ops.gemm_nt(x1, w1, ops.EPI_GELU_DG, bias=b4, tile=t), 2.0 * M * D * 4 * D),
    ("fc2+resid", lambda t: ops.gemm_nt(x4, w2, ops.EPI_RESID_F32, bias=b1, aux=res_in, tile=t), 2.0 * M * D * 4 * D),
    ("dgrad_fc2", lambda t: ops.gemm_nt(x1, w2_t, ops.EPI_DMUL, aux=x4, colsum=torch.zeros(4 * D, device=dev), tile=t), 2.0 * M * D * 4 * D),
    ("dgrad_fc1", lambda t: ops.gemm_nt(x4, w1_t, ops.EPI_BIAS_BF16, tile=t), 2.0 * M * D * 4 * D),
    ("dgrad_qkv", lambda t: ops.gemm_nt(x3, wqkv_t, ops.EPI_BIAS_BF16, tile=t), 2.0 * M * D * 3 * D),
]
tot = {k: 0.0 for k in cfgs}
for name, fn, fl in calls:
    ref = None
    res = {k: [] for k in cfgs}
    for rnd in range(5):
        for k, (t, dbg) in cfgs.items():
            _L.vitamd_set_debug(dbg)
            out = fn(t)
            if rnd == 0:
                torch.cuda.synchronize()
                o = out[1] if isinstance(out, tuple) else out
                o0 = out[0] if isinstance(out, tuple) else out
                if ref is None: ref = (o.float().clone(), o0.float().clone())
                else:
                    same = torch.equal(o.float(), ref[0]) and torch.equal(o0.float(), ref[1])
                    if not same: print(f"  {name} {k}: MISMATCH rel {float((o.float() - ref[0]).norm() / ref[0].norm()):.3e}", flush=True)
            del out
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); s.record()
            for _ in range(10): fn(t)
            e.record(); torch.cuda.synchronize()
            res[k].append(s.elapsed_time(e) / 10 * 1e3)
            _L.vitamd_set_debug(0)
    for k in cfgs:
        med = statistics.median(res[k]); tot[k] += med
        print(f"{name:10s} {k:10s} {med:7.1f} us  {fl / med / 1e6:7.1f} TF  {['%.0f' % v for v in res[k]]}", flush=True)
print({k: round(v, 1) for k, v in tot.items()}, "us per layer (sum of the six launches)")
