"""tools/check_seam.py - exactness of the seam kernel (experimental tile codes 24 / 25) against one-workgroup-per-tile launches of the ping-pong kernel: integer-valued
operands, every epilogue, three repetitions each (a race shows as a count that changes between repetitions)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import lib as _explib; _explib.use_experimental()
from vitamd import ops
dev = torch.device("cuda")
g = torch.Generator(device="cpu").manual_seed(1)
shapes = ((512, 512, 128), (256 * 40, 768, 768), (256 * 197, 2304, 768), (320 * 30 + 64, 768, 3072))
for (M, N, K) in shapes:
    a = torch.randint(-2, 3, (M, K), generator=g).to(dev, torch.bfloat16)
    b = torch.randint(-2, 3, (N, K), generator=g).to(dev, torch.bfloat16)
    bias = torch.randint(-4, 5, (N,), generator=g).float().to(dev)
    fac = (torch.randint(-2, 3, (M, N), generator=g).float() * 0.5).to(dev, torch.bfloat16)
    def run(epi, tile):
        if epi == "bias": return (ops.gemm_nt(a, b, ops.EPI_BIAS_BF16, bias=bias, tile=tile),)
        if epi == "nobias": return (ops.gemm_nt(a, b, ops.EPI_BIAS_BF16, tile=tile),)
        if epi == "gelu": return ops.gemm_nt(a * 0.125, b * 0.125, ops.EPI_GELU_DG, bias=bias * 0.25, tile=tile)
        cs = torch.zeros(N, device=dev)
        return (ops.gemm_nt(a, b, ops.EPI_DMUL, aux=fac, colsum=cs, tile=tile), cs)
    for epi in ("nobias", "bias", "gelu", "dmul"):
        ref = [t.float() for t in run(epi, 512)]
        for tile in ((24, 2048) if epi == "dmul" else (24, 25, 2048) if epi in ("bias", "nobias") else (24, 2048)):
            if epi == "dmul" and N % 256: continue
            counts = []
            for rep in range(3):
                out = [t.float() for t in run(epi, tile)]
                torch.cuda.synchronize()
                counts.append([int(((o != r) | torch.isnan(o)).sum()) for o, r in zip(out, ref)])
            print(f"M{M} N{N} K{K} {epi:7s} tile{tile}: mismatches per output, 3 reps: {counts}", flush=True)
