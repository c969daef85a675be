"""Exactness matrix of the loader-wave NT GEMM (tile code 2048) against gemm_nt_pp_kernel on one workgroup per tile (tile code 256): random data,
bit for bit (same per-accumulator k order), every epilogue of the loader form, ragged M / N, one and several tiles per workgroup, three repetitions."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import ops, lib as _lib
if "--exp" in sys.argv: _lib.use_experimental()      # also checks the 320-row form on ten compute waves (tile code 4096: experimental library only)
EXP = "--exp" in sys.argv
dev = torch.device("cuda")
g = torch.Generator(device="cpu").manual_seed(7)
def rb(*s, scale=1.0): return (torch.randn(*s, generator=g) * scale).to(dev, torch.bfloat16)
bad = 0
shapes = [(1024, 768, 256), (1000, 264, 128), (256 * 40, 2304, 768), (153600, 256, 128), (50432, 768, 3072), (50432, 2304, 768), (50000, 3072, 768)]
_a = [v for v in sys.argv[1:] if not v.startswith("--")]
if _a: shapes = shapes[: int(_a[0])]
for (M, N, K) in shapes:
    a, b = rb(M, K), rb(N, K, scale=0.05)
    bias = torch.randn(N, device=dev)
    aux = rb(M, N)
    for epi, name in ((ops.EPI_BIAS_BF16, "bias"), (ops.EPI_GELU_DG, "gelu_dg"), (ops.EPI_GELU, "gelu"), (ops.EPI_DMUL, "dmul")):
        if epi == ops.EPI_DMUL and N % 256: continue
        for rep in range(3):
            outs = {}
            for t in (256, 2048):
                cs = torch.zeros(N, device=dev)
                kw = dict(bias=bias) if epi != ops.EPI_DMUL else dict(aux=aux, colsum=cs)
                o = ops.gemm_nt(a, b, epi, tile=t, **kw)
                torch.cuda.synchronize()
                o = o if isinstance(o, tuple) else (o,)
                outs[t] = [x.clone() for x in o] + ([cs] if epi == ops.EPI_DMUL else [])
            ok = True
            if EXP and epi == ops.EPI_BIAS_BF16 and K % 128 == 0:             # the 320-row form (ten compute + two loader waves; tile code 4096): plain-bias epilogue only
                y320 = ops.gemm_nt(a, b, epi, tile=4096, bias=bias)
                torch.cuda.synchronize()
                if not torch.equal(y320, outs[256][0]):
                    d = (y320.float() - outs[256][0].float()); rows = torch.nonzero((d != 0).any(dim=1)).flatten()
                    print(f"   ld10: {int((d != 0).sum())} elements differ, rows {rows[:6].tolist()}..{rows[-3:].tolist()} nan {int(torch.isnan(y320.float()).sum())}")
                    ok = False
            for idx, (x, y) in enumerate(zip(outs[256], outs[2048])):
                if x.dtype == torch.float32:      # column sums: atomics order differs
                    if not torch.allclose(x, y, rtol=1e-4, atol=1e-2): ok = False; print("   colsum diff", float((x - y).abs().max()))
                elif not torch.equal(x, y):
                    d = (x.float() - y.float())
                    nbad = int((d != 0).sum())
                    # the table GELU of the loader / seam forms differs from the formula of gemm_nt_pp_kernel by one bf16 ulp on ~1e-4 of the elements
                    lim = 2e-4 * x.numel() if epi in (ops.EPI_GELU, ops.EPI_GELU_DG) else 0
                    if nbad > lim or not torch.isfinite(y.float()).all():
                        ok = False
                        rows = torch.nonzero((d != 0).any(dim=1)).flatten()
                        print(f"   out{idx}: {nbad} elements differ, rel {float(d.norm() / x.float().norm()):.3e}, rows {rows[:6].tolist()}..{rows[-3:].tolist()} nan {int(torch.isnan(y.float()).sum())}")
            bad += 0 if ok else 1
            print(f"M {M} N {N} K {K} {name:8s} rep {rep}: {'ok' if ok else 'MISMATCH'}", flush=True)
print("check_ld:", "ALL OK" if bad == 0 else f"{bad} FAILED")
sys.exit(1 if bad else 0)
