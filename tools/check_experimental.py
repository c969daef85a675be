"""Exactness checks of the measured alternative kernels of libvitamd_exp.so (integer data: every result must be bit-exact).
Run by tests/test_gpu_kernels.py::test_experimental_library_alternatives in a child process; exit code 0 = all good."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import lib
lib.use_experimental()
from vitamd import ops
L = lib.load()
dev, BF16 = torch.device("cuda"), torch.bfloat16
bad = []


def ints(shape, lo, hi, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi + 1, shape, generator=g).float()


# NT alternatives: 1 ring, 2 pipe, 3-5 deep, 6 persistent, 257 plain double-buffered 256x256
for (M, N, K, tile) in [(256, 128, 32, 1), (512, 768, 768, 1), (1000, 2304, 96, 1), (333, 200, 160, 1), (256, 256, 32, 4), (512, 768, 768, 4),
                        (1000, 2304, 96, 3), (333, 200, 160, 5), (197 * 64, 768, 3072, 4), (256, 256, 64, 2), (512, 768, 768, 2), (1000, 2304, 128, 2), (197 * 64, 768, 3072, 2),
                        (300, 1024, 192, 257)]:
    a, b = ints((M, K), -3, 3, 1), ints((N, K), -2, 2, 2)
    out = ops.gemm_nt(a.to(dev, BF16), b.to(dev, BF16), ops.EPI_F32, tile=tile)
    if not torch.equal(out.cpu(), a @ b.t()):
        bad.append(("nt", M, N, K, tile))
# the persistent kernel has no fp32 epilogue: bf16 outputs of {-1,0,1} data with K <= 256 are exact integers
for (M, N, K) in [(256, 256, 64), (512, 768, 256), (1000, 2304, 128), (333, 200, 192), (50432, 768, 128)]:
    a, b = ints((M, K), -1, 1, 3), ints((N, K), -1, 1, 4)
    out = ops.gemm_nt(a.to(dev, BF16), b.to(dev, BF16), ops.EPI_BIAS_BF16, tile=6)
    if not torch.equal(out.float().cpu(), a @ b.t()):
        bad.append(("nt-persist", M, N, K))
# the pipe kernel (tile 2) with the fused epilogues against the production kernel
M, N, K = 20380, 3072, 128
g = torch.Generator().manual_seed(5)
a = torch.randn(M, K, generator=g).to(dev, BF16); b = (torch.randn(N, K, generator=g) * 0.1).to(dev, BF16)
bias = torch.randn(N, generator=g).to(dev); aux = (torch.randn(M, N, generator=g) * 0.5).to(dev, BF16)
for epi in (ops.EPI_GELU, ops.EPI_GELU_DG):
    o1, h1 = ops.gemm_nt(a, b, epi, bias=bias, tile=256); o2, h2 = ops.gemm_nt(a, b, epi, bias=bias, tile=2)      # (256: the formula GELU, as the pipe kernel; tile 0 is the table form)
    if not (torch.equal(o1, o2) and torch.equal(h1, h2)):
        bad.append(("nt-epi", epi))
y1 = ops.gemm_nt(a, b, ops.EPI_DMUL, aux=aux, colsum=torch.zeros(N, device=dev)); y2 = ops.gemm_nt(a, b, ops.EPI_DMUL, aux=aux, colsum=torch.zeros(N, device=dev), tile=2)
if not torch.equal(y1, y2):
    bad.append(("nt-epi", "dmul"))
# TN alternatives behind vitamd_set_debug
for bits in (64, 5 << 26, 6 << 26, 7 << 26):
    for (R, P, Q) in [(1000, 256, 256), (4133, 768, 512), (300, 200, 136)]:
        l, r = ints((R, P), -2, 2, 61), ints((R, Q), -3, 3, 62)
        L.vitamd_set_debug(bits)
        out = torch.full((P, Q), 7.0, device=dev)
        ops.gemm_tn(l.to(dev, BF16), r.to(dev, BF16), out, accumulate=False)
        torch.cuda.synchronize()
        L.vitamd_set_debug(0)
        if not torch.equal(out.cpu(), l.t() @ r):
            bad.append(("tn", bits, R, P, Q))
L.vitamd_set_debug(1 << 25)      # 256x384-tile kernel
for (R, P, Q) in [(4096, 256, 384), (5000, 512, 768), (8197, 768, 768), (4100, 256, 1152)]:
    l, r, init = ints((R, P), -2, 2, 31), ints((R, Q), -3, 3, 32), ints((P, Q), -5, 5, 33)
    out = torch.full((P, Q), 123.0, device=dev)
    ops.gemm_tn(l.to(dev, BF16), r.to(dev, BF16), out, accumulate=False)
    out2 = init.to(dev)
    ops.gemm_tn(l.to(dev, BF16), r.to(dev, BF16), out2, accumulate=True)
    if not (torch.equal(out.cpu(), l.t() @ r) and torch.equal(out2.cpu(), init + l.t() @ r)):
        bad.append(("tn-wide", R, P, Q))
L.vitamd_set_debug(0)
# fused attention backward (bit 9) against the production two-kernel form
for (B, N, H, causal) in [(3, 197, 4, False), (2, 65, 2, True), (2, 256, 2, False), (1, 5, 2, False)]:
    g = torch.Generator().manual_seed(9)
    qkv = torch.randn(B * N, 3 * H * 64, generator=g).to(dev, BF16); d_o = torch.randn(B * N, H * 64, generator=g).to(dev, BF16)
    o, lse = ops.attention_fwd(qkv, B, N, H, causal)
    ref = ops.attention_bwd(qkv, o, lse, d_o, B, N, H, causal).float()
    L.vitamd_set_debug(0x200)
    got = ops.attention_bwd(qkv, o, lse, d_o, B, N, H, causal).float()
    L.vitamd_set_debug(0)
    err = float((got - ref).norm() / ref.norm())
    if not err < 1e-3:
        bad.append(("attn-fused", B, N, H, causal, err))
# the software-pipelined attention backward kernels (production, 33 <= N <= 256 non-causal) against the plain loops (bit 17): bit-identical
for (B, N, H) in [(2, 33, 2), (3, 64, 1), (2, 100, 3), (1, 160, 2), (2, 197, 4), (1, 224, 2), (2, 256, 1)]:
    g = torch.Generator().manual_seed(10 + N)
    qkv = torch.randn(B * N, 3 * H * 64, generator=g).to(dev, BF16); d_o = torch.randn(B * N, H * 64, generator=g).to(dev, BF16)
    o, lse = ops.attention_fwd(qkv, B, N, H, False)
    db = torch.zeros(3 * H * 64, device=dev); got = ops.attention_bwd(qkv, o, lse, d_o, B, N, H, False, dbias=db)
    L.vitamd_set_debug(0x20000)
    db2 = torch.zeros(3 * H * 64, device=dev); ref = ops.attention_bwd(qkv, o, lse, d_o, B, N, H, False, dbias=db2)
    L.vitamd_set_debug(0)
    if not torch.equal(got, ref) or not torch.allclose(db, db2, rtol=1e-5, atol=1e-5):
        bad.append(("attn-bwd-pipe", B, N, H))
# round 3: explicit seam-kernel codes (24 = 256-row tiles, 25 = 320-row) and the 5-slot-ring weight-gradient kernel (debug bit 31)
for (M, N, K, tile) in [(512, 512, 128, 24), (256 * 40, 768, 768, 24), (320 * 30 + 64, 768, 768, 25), (512, 512, 128, 25)]:
    a, b = ints((M, K), -1, 1, 5), ints((N, K), -1, 1, 6)
    out = ops.gemm_nt(a.to(dev, BF16), b.to(dev, BF16), ops.EPI_BIAS_BF16, tile=tile)
    if not torch.equal(out.float().cpu(), (a @ b.t()).to(BF16).float()):
        bad.append(("nt-seam", M, N, K, tile))
L.vitamd_set_debug(-2147483648)
for (R, P, Q) in [(4096, 768, 3072), (1000, 512, 768)]:
    l, r = ints((R, P), -2, 2, 7), ints((R, Q), -2, 2, 8)
    out = torch.full((P, Q), float("nan"), device=dev)
    ops.gemm_tn(l.to(dev, BF16), r.to(dev, BF16), out, accumulate=False)
    if not torch.equal(out.cpu(), l.t() @ r):
        bad.append(("tn-ring5", R, P, Q))
L.vitamd_set_debug(0)
print("experimental checks:", "ok" if not bad else bad)
sys.exit(1 if bad else 0)
