"""Seam form (tile 0) against the plain persistent form (tile 1024) of the NT GEMM across row counts: where does the rule of
gemm_nt.hip::dispatch_tile (K <= 1536, >= 3 tiles per CU) pay?  Interleaved, medians of 5 x 8 launches, production library."""
import os, sys, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import ops
ops.NT_SEAM = True
dev = torch.device("cuda")
g = torch.Generator(device="cpu").manual_seed(1)
def rb(*s, scale=1.0): return (torch.randn(*s, generator=g) * scale).to(dev, torch.bfloat16)
def t(fn, n=8):
    fn(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
for D in (768, 512, 1024):
    for rows in (12608, 18912, 25216, 37824, 50432, 75648):
        x1, x4 = rb(rows, D), rb(rows, 4 * D)
        shapes = {"qkv": (lambda tl: ops.gemm_nt(x1, wq, ops.EPI_BIAS_BF16, bias=b3, tile=tl), 3 * D),
                  "fc1+gelu": (lambda tl: ops.gemm_nt(x1, w1, ops.EPI_GELU_DG, bias=b4, tile=tl), 4 * D),
                  "dgrad_fc2": (lambda tl: ops.gemm_nt(x1, w2t, ops.EPI_DMUL, aux=x4, colsum=cs, tile=tl), 4 * D)}
        wq, w1, w2t = rb(3 * D, D, scale=0.03), rb(4 * D, D, scale=0.03), rb(4 * D, D, scale=0.03)
        b3, b4, cs = torch.randn(3 * D, device=dev), torch.randn(4 * D, device=dev), torch.zeros(4 * D, device=dev)
        line = f"D {D:4d} rows {rows:6d}:"
        for name, (fn, N) in shapes.items():
            a, b = [], []
            for _ in range(5):
                a.append(t(lambda: fn(0))); b.append(t(lambda: fn(1024)))
            tiles = ((rows + 255) // 256) * ((N + 255) // 256)
            line += f"  {name} tiles/CU {tiles / 256:4.1f} seam {statistics.median(a):6.1f} plain {statistics.median(b):6.1f} ({statistics.median(a) / statistics.median(b) - 1:+.1%})"
        print(line, flush=True)
        del x1, x4
