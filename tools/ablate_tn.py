"""Weight-gradient (TN) GEMM: what bounds the main loop?  interleaved medians of the ablated kernels."""
import os, sys, torch, ctypes, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import ops, lib
lib.use_experimental(); L = lib.load(); L.vitamd_set_debug.argtypes = [ctypes.c_int]
dev = torch.device("cuda")
M, D = 256 * 197, 768
g = torch.Generator(device="cpu").manual_seed(0)
rb = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dev, torch.bfloat16)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for name, (P, Q) in {"dW_fc1 [3072 x 768]": (4 * D, D), "dW_fc2 [768 x 3072]": (D, 4 * D), "dW_qkv [2304 x 768]": (3 * D, D)}.items():
    l, r = rb(M, P), rb(M, Q)
    out = torch.zeros(P, Q, device=dev)
    fn = lambda: ops.gemm_tn(l, r, out, accumulate=False)
    for _ in range(10): fn()
    cfg = {"full": 0, "no MFMA": 1 << 26, "no DMA": 2 << 26, "no tr-reads": 3 << 26, "via VGPR": 4 << 26}
    res = {k: [] for k in cfg}
    for rr in range(5):
        for k, bits in cfg.items():
            L.vitamd_set_debug(bits); res[k].append(t(fn))
    L.vitamd_set_debug(0)
    ref = out.clone(); L.vitamd_set_debug(4 << 26); fn(); L.vitamd_set_debug(0)
    print("   via-VGPR result equals LDS-DMA result:", bool(torch.equal(ref, out)))
    fl = 2.0 * M * P * Q
    print(f"{name}: " + "  ".join(f"{k} {statistics.median(v):6.1f} us" for k, v in res.items()) + f"   ({fl / statistics.median(res['full']) / 1e6:.0f} TF incl. reduce)", flush=True)
