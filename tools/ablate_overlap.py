"""Where does a K=768 GEMM's time go: main loop alone (stores suppressed), epilogue alone (one K-tile), both."""
import os, sys, torch, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import ops, lib
lib.use_experimental(); L = lib.load(); L.vitamd_set_debug.argtypes = [ctypes.c_int]
dev = torch.device("cuda")
M, D = 256 * 197, 768
g = torch.Generator(device="cpu").manual_seed(0)
rb = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dev, torch.bfloat16)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
shapes = {"fc1 gelu  N=3072 K=768": (4 * D, D, ops.EPI_GELU), "qkv bias  N=2304 K=768": (3 * D, D, ops.EPI_BIAS_BF16),
          "dfc1 bias N=768 K=3072": (D, 4 * D, ops.EPI_BIAS_BF16), "fc2 resid N=768 K=3072": (D, 4 * D, ops.EPI_RESID_F32),
          "dgelu     N=3072 K=768": (4 * D, D, ops.EPI_DGELU)}
for name, (N, K, epi) in shapes.items():
    x, w = rb(M, K), rb(N, K, scale=0.03)
    bias = torch.randn(N, device=dev)
    aux = torch.randn(M, N, device=dev) if epi == ops.EPI_RESID_F32 else (rb(M, N) if epi == ops.EPI_DGELU else None)
    cs = torch.zeros(N, device=dev) if epi == ops.EPI_DGELU else None
    fn = lambda: ops.gemm_nt(x, w, epi, bias=None if epi == ops.EPI_DGELU else bias, aux=aux, colsum=cs, tile=2)
    res = []
    for bits in (0, 0x10000, 0x20000, 0x30000):
        L.vitamd_set_debug(bits | (255 << 8)); res.append(t(fn))      # stagger off for a clean decomposition
    L.vitamd_set_debug(0); st = t(fn)
    print(f"{name}: full {res[0]:6.1f}  no-stores {res[1]:6.1f}  one-K-tile {res[2]:6.1f}  one-K-tile+no-stores {res[3]:6.1f}   (production, stagger on: {st:6.1f}) us", flush=True)
    del x, w, aux

print("multi-phase stagger: P phases, unit u (~us): delay = phase*u; map A = blockIdx%P, map B = (blockIdx>>3)%P")
for name in ("fc1 gelu  N=3072 K=768", "dgelu     N=3072 K=768", "fc2 resid N=768 K=3072", "qkv bias  N=2304 K=768"):
    N, K, epi = shapes[name]
    x, w = rb(M, K), rb(N, K, scale=0.03)
    bias = torch.randn(N, device=dev)
    aux = torch.randn(M, N, device=dev) if epi == ops.EPI_RESID_F32 else (rb(M, N) if epi == ops.EPI_DGELU else None)
    cs = torch.zeros(N, device=dev) if epi == ops.EPI_DGELU else None
    fn = lambda: ops.gemm_nt(x, w, epi, bias=None if epi == ops.EPI_DGELU else bias, aux=aux, colsum=cs, tile=2)
    out = []
    for (P, u, mapb) in ((2, 8, 0), (2, 8, 1), (2, 12, 1), (4, 4, 0), (4, 4, 1), (4, 6, 1), (8, 2, 0), (8, 2, 1), (8, 3, 1), (8, 4, 1), (15, 2, 1)):
        L.vitamd_set_debug((u << 8) | (P << 20) | (mapb << 24)); out.append(f"P{P}u{u}{'B' if mapb else 'A'}={t(fn):.0f}")
    L.vitamd_set_debug(0)
    print(f"{name}: " + "  ".join(out), flush=True)
    del x, w, aux
