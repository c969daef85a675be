"""Weight-gradient (TN) GEMM variants on the three ViT-B shapes: interleaved rounds in one process, random data, medians.
usage: bench_tn.py [name=dbgbits ...]   (bit 23 = the loader-wave form, bits 16-17 its timing-only ablations: csrc/gemm_tn.hip)"""
import os, sys, statistics, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import ops, lib
lib.use_experimental(); L = lib.load(); L.vitamd_set_debug.argtypes = [ctypes.c_int]
cfgs = {"pp_d4": 0, "loader_waves": 0x800000, "ld_no_mfma(!)": 0x800000 | 1 << 16, "ld_no_lds_reads(!)": 0x800000 | 2 << 16, "ld_neither(!)": 0x800000 | 3 << 16} 
for a in sys.argv[1:]:
    k, v = a.split("="); cfgs[k] = int(v, 0)
dev = torch.device("cuda")
R = 256 * 197
shapes = [("dWqkv", 2304, 768), ("dW1", 3072, 768), ("dW2", 768, 3072)]
g = torch.Generator(device="cpu").manual_seed(3)
for name, P, Q in shapes:
    l = torch.randn(R, P, generator=g).to(dev, torch.bfloat16)
    r = torch.randn(R, Q, generator=g).to(dev, torch.bfloat16)
    out = torch.empty(P, Q, device=dev)
    ref = None
    res = {k: [] for k in cfgs}
    for rnd in range(5):
        for k, bits in cfgs.items():
            L.vitamd_set_debug(bits)
            ops.gemm_tn(l, r, out, accumulate=False)
            if rnd == 0:
                torch.cuda.synchronize()
                if ref is None: ref = out.clone()
                else:
                    err = float((out - ref).norm() / ref.norm())
                    print(f"  {name} {k}: rel diff vs first variant {err:.2e}", flush=True)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); s.record()
            for _ in range(10): ops.gemm_tn(l, r, out, accumulate=False)
            e.record(); torch.cuda.synchronize()
            res[k].append(s.elapsed_time(e) / 10 * 1e3)
    L.vitamd_set_debug(0)
    fl = 2.0 * R * P * Q
    for k in cfgs:
        med = statistics.median(res[k])
        print(f"{name:6s} {k:16s} {med:7.1f} us  {fl / med / 1e6:7.1f} TF  {['%.0f' % v for v in res[k]]}", flush=True)
