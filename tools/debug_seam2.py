import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import ops
dev = torch.device("cuda")
M, D = 256 * 197, 768
g = torch.Generator(device="cpu").manual_seed(1)
def rb(*s, scale=1.0): return (torch.randn(*s, generator=g) * scale).to(dev, torch.bfloat16)
x1, x4 = rb(M, D), rb(M, 4 * D)
w2_t = rb(4 * D, D, scale=0.03)
cs = torch.zeros(4 * D, device=dev)
def t(fn, n=10):
    fn(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
for rep in range(3):
    a = t(lambda: ops.gemm_nt(x1, w2_t, ops.EPI_DMUL, aux=x4, colsum=cs))
    b = t(lambda: ops.gemm_nt(x1, w2_t, ops.EPI_DMUL, aux=x4, colsum=torch.zeros(4 * D, device=dev)))
    c = t(lambda: ops.gemm_nt(x1, w2_t, ops.EPI_DMUL, aux=x4, colsum=cs, tile=512))
    print(f"persistent colsum {a:.1f} us | fresh colsum {b:.1f} us | one-wg-per-tile, persistent colsum {c:.1f} us | cs absmax {float(cs.abs().max()):.3g} nan {int(torch.isnan(cs).sum())}", flush=True)
out = torch.empty(M, 4 * D, device=dev, dtype=torch.bfloat16)
d = t(lambda: ops.gemm_nt(x1, w2_t, ops.EPI_DMUL, aux=x4, colsum=cs, out=out))
print(f"persistent colsum + persistent out {d:.1f} us")
