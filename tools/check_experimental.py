"""Exactness checks of what libvitamd_exp.so still carries beyond the production library (the measured alternative kernels of rounds 1-3 were deleted in
round 4; their numbers live in DESIGN.md and profiles/r02, r03): explicit seam-kernel codes, the loader kernel's first request schedule, the round-3
weight-gradient loaders, the split-role attention backward, the plain-loop attention backward.  Every result must be bit-identical to production.
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


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev, BF16)


# explicit seam-kernel codes (24 = 256-row tiles, 25 = 320-row, 30 = 256-row with the GELU table), the loader kernel's burst schedule (2049) and the
# 320-row loader form on ten compute waves (4096)
for (M, N, K, tile) in [(512, 512, 128, 24), (256 * 40, 768, 768, 24), (320 * 30 + 64, 768, 768, 25), (512, 512, 128, 25), (256 * 40, 768, 256, 2049), (1000, 264, 128, 2049), (320 * 30 + 7, 768, 256, 4096), (1000, 264, 128, 4096)]:
    a, b = ints((M, K), -1, 1, 5), ints((N, K), -1, 1, 6)
    out = ops.gemm_nt(a.to(dev, BF16), b.to(dev, BF16), ops.EPI_BIAS_BF16, tile=tile)
    if not torch.equal(out.float().cpu(), (a @ b.t()).to(BF16).float()):
        bad.append(("nt-explicit", M, N, K, tile))
a, b, bias = rnd((256 * 30 + 5, 256), 1), rnd((1024, 256), 2, 0.1), torch.randn(1024, device=dev)
ref = ops.gemm_nt(a, b, ops.EPI_GELU_DG, bias=bias, tile=256)
for tile in (30, 2048, 2049):
    got = ops.gemm_nt(a, b, ops.EPI_GELU_DG, bias=bias, tile=tile)
    if not (torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])):
        bad.append(("nt-gelu", tile))
# the round-3 weight-gradient loaders (vitamd_set_debug2 bit 6) against the shipped ones (requests split around the first barrier)
for (R, P, Q) in [(4096, 768, 3072), (1000, 512, 768)]:
    l, r = ints((R, P), -2, 2, 7), ints((R, Q), -2, 2, 8)
    for bits in (0, 64):
        L.vitamd_set_debug2(bits)
        out = torch.full((P, Q), float("nan"), device=dev)
        ops.gemm_tn(l.to(dev, BF16), r.to(dev, BF16), out, accumulate=False, form=ops.TN_FORM_EXCLUSIVE)
        if not torch.equal(out.cpu(), l.t() @ r):
            bad.append(("tn-loaders", bits, R, P, Q))
L.vitamd_set_debug2(0)
# attention backward: the split-role kernel (debug2 bit 4) and the plain loops (debug bit 17) against the pipelined two-kernel production form
for (B, N, H) in [(2, 33, 2), (3, 64, 1), (2, 100, 3), (1, 160, 2), (2, 197, 4), (1, 224, 2)]:
    g = torch.Generator().manual_seed(10 + N)
    qkv = torch.randn(B * N, 3 * H * 64, generator=g).to(dev, BF16); d_o = torch.randn(B * N, H * 64, generator=g).to(dev, BF16)
    o, lse = ops.attention_fwd(qkv, B, N, H, False)
    db = torch.zeros(3 * H * 64, device=dev); got = ops.attention_bwd(qkv, o, lse, d_o, B, N, H, False, dbias=db)
    L.vitamd_set_debug(0x20000)
    db2 = torch.zeros(3 * H * 64, device=dev); ref = ops.attention_bwd(qkv, o, lse, d_o, B, N, H, False, dbias=db2)
    L.vitamd_set_debug(0)
    if not torch.equal(got, ref) or not torch.allclose(db, db2, rtol=1e-5, atol=1e-5):
        bad.append(("attn-bwd-pipe", B, N, H))
    if N <= 224:
        L.vitamd_set_debug2(16)
        db3 = torch.zeros(3 * H * 64, device=dev); spl = ops.attention_bwd(qkv, o, lse, d_o, B, N, H, False, dbias=db3)
        L.vitamd_set_debug2(0)
        if not torch.equal(got, spl) or not torch.allclose(db, db3, rtol=1e-5, atol=1e-5):
            bad.append(("attn-bwd-split", B, N, H))
print("experimental checks:", "ok" if not bad else bad)
sys.exit(1 if bad else 0)
