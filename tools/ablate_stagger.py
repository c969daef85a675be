"""Careful A/B of the first-round phase stagger: configs interleaved round-robin, median of 7 rounds of 10 launches."""
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
shapes = {"fc1 gelu  N=3072 K=768": (4 * D, D, ops.EPI_GELU), "qkv bias  N=2304 K=768": (3 * D, D, ops.EPI_BIAS_BF16),
          "dfc1 bias N=768 K=3072": (D, 4 * D, ops.EPI_BIAS_BF16), "dqkv bias N=768 K=2304": (D, 3 * D, ops.EPI_BIAS_BF16),
          "fc2 resid N=768 K=3072": (D, 4 * D, ops.EPI_RESID_F32), "dgelu     N=3072 K=768": (4 * D, D, ops.EPI_DGELU)}
cfgs = {"off": 255 << 8, "P2u8A(prod)": 0, "P4u4B": (4 << 8) | (4 << 20) | (1 << 24), "P8u2B": (2 << 8) | (8 << 20) | (1 << 24),
        "P8u3B": (3 << 8) | (8 << 20) | (1 << 24), "P15u1B": (1 << 8) | (15 << 20) | (1 << 24), "P15u2B": (2 << 8) | (15 << 20) | (1 << 24)}
tot = {k: 0.0 for k in cfgs}
for name, (N, K, epi) in shapes.items():
    x, w = rb(M, K), rb(N, K, scale=0.03)
    bias = torch.randn(N, device=dev)
    aux = torch.randn(M, N, device=dev) if epi == ops.EPI_RESID_F32 else (rb(M, N) if epi == ops.EPI_DGELU else None)
    cs = torch.zeros(N, device=dev) if epi == ops.EPI_DGELU else None
    fn = lambda: ops.gemm_nt(x, w, epi, bias=None if epi == ops.EPI_DGELU else bias, aux=aux, colsum=cs, tile=2)
    for _ in range(20): fn()
    res = {k: [] for k in cfgs}
    for r in range(7):
        for k, bits in cfgs.items():
            L.vitamd_set_debug(bits); res[k].append(t(fn))
    L.vitamd_set_debug(0)
    med = {k: statistics.median(v) for k, v in res.items()}
    for k in cfgs: tot[k] += med[k]
    print(f"{name}: " + "  ".join(f"{k}={med[k]:.0f}" for k in cfgs), flush=True)
    del x, w, aux
print("sum:", "  ".join(f"{k}={v:.0f}" for k, v in tot.items()))
