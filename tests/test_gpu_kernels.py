"""Per-kernel checks of libvitamd.so through the C ABI on a real MI355X.
Integer-valued operands make the MFMA results exact, so fragment-layout mistakes show up as
exact mismatches (asymmetric operands: a transposed result cannot pass)."""
import math

import pytest
import torch

import vit_oracle as O

pytestmark = pytest.mark.gpu

BF16, F32 = torch.bfloat16, torch.float32


def dev():
    return torch.device("cuda")


def ints(shape, lo, hi, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi + 1, shape, generator=g).float()


def randn(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def r16(x):
    return x.to(BF16).float()


# ------------------------------------------------------------------------------------------ gemm_nt
@pytest.mark.parametrize("M,N,K,tile", [(256, 256, 64, 0), (512, 768, 768, 0), (1000, 2304, 128, 0), (333, 200, 192, 0), (197 * 4, 768, 3072, 0),
                                        (64, 64, 64, 0), (197 * 64, 768, 3072, 0), (197 * 64, 3072, 768, 0), (50432, 2304, 768, 0),
                                        (256, 256, 64, 128), (512, 768, 768, 128), (333, 200, 192, 128), (1000, 2304, 128, 128),
                                        (256, 256, 64, 256), (512, 768, 768, 256), (1000, 2304, 128, 256), (333, 200, 192, 256), (300, 1024, 192, 256),
                                        (1000, 2304, 768, 256), (197 * 64, 768, 3072, 256), (50432, 768, 768, 256)])
def test_gemm_nt_exact_integers(hip, M, N, K, tile):
    from vitamd import ops
    a = ints((M, K), -3, 3, 1)
    b = ints((N, K), -2, 2, 2)
    ref = a @ b.t()
    out = ops.gemm_nt(a.to(dev(), BF16), b.to(dev(), BF16), ops.EPI_F32, tile=tile)
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), ref)


def test_gemm_nt_persistent_launch_equals_one_workgroup_per_tile(hip):
    """The automatic launch of a problem with more tiles than CUs is persistent (one workgroup per CU walks a strided tile list);
    tile code 512 is the same automatic choice with one workgroup per tile.  Same arithmetic: bit-identical outputs (GELU too: every kernel reads
    the one erf-GELU table, round 4), for every fused epilogue of the training step, with a ragged last tile row."""
    from vitamd import ops
    M, D = 320 * 300 + 77, 768          # 301 x 3 = 903 tiles of 320 rows (or 1 128 of 256): more than any CU count
    a = r16(randn((M, D), 71)).to(dev(), BF16)
    w = r16(randn((3 * D, D), 72, 0.05)).to(dev(), BF16)
    wt = r16(randn((D, 3 * D), 73, 0.05)).to(dev(), BF16)
    a3 = r16(randn((M, 3 * D), 74)).to(dev(), BF16)
    bias3, bias1 = randn((3 * D,), 75).to(dev()), randn((D,), 76).to(dev())
    res = randn((M, D), 77).to(dev())
    cases = [
        (a, w, ops.EPI_BIAS_BF16, dict(bias=bias3)),
        (a, w, ops.EPI_GELU_DG, dict(bias=bias3)),
        (a3, wt, ops.EPI_RESID_F32, dict(bias=bias1, aux=res)),
        (a3, wt, ops.EPI_BIAS_BF16, dict()),
        (a, w, ops.EPI_DMUL, dict(aux=a3)),
    ]
    for x, wgt, epi, kw in cases:
        cs1 = torch.zeros(wgt.shape[0], device=dev()) if epi == ops.EPI_DMUL else None
        cs2 = torch.zeros(wgt.shape[0], device=dev()) if epi == ops.EPI_DMUL else None
        y1 = ops.gemm_nt(x, wgt, epi, tile=0, colsum=cs1, **kw)
        y2 = ops.gemm_nt(x, wgt, epi, tile=512, colsum=cs2, **kw)
        for u, v in zip(y1 if isinstance(y1, tuple) else (y1,), y2 if isinstance(y2, tuple) else (y2,)):
            assert torch.equal(u, v), epi
        if cs1 is not None:
            assert O.rel_l2(cs1.cpu(), cs2.cpu()) < 1e-5          # column sums are accumulated with atomics: order differs


@pytest.mark.parametrize("tile", [128, 256, 320])      # the small-problem kernel and the ping-pong kernel on 256- / 320-row tiles
def test_gemm_nt_epilogues(hip, tile):
    import functools
    from vitamd import ops as _ops

    class ops:  # same API with the tile selector pinned
        gemm_nt = staticmethod(functools.partial(_ops.gemm_nt, tile=tile))
        EPI_BIAS_BF16, EPI_GELU, EPI_RESID_F32, EPI_DGELU, EPI_PATCH_F32, EPI_F32, EPI_GELU_DG, EPI_DMUL = range(8)
    M, N, K = 333, 512, 256
    a, b = r16(randn((M, K), 3)), r16(randn((N, K), 4, 0.1))
    bias = randn((N,), 5)
    acc = a @ b.t()
    ad, bd, biasd = a.to(dev(), BF16), b.to(dev(), BF16), bias.to(dev())
    # bias -> bf16
    y = ops.gemm_nt(ad, bd, ops.EPI_BIAS_BF16, bias=biasd).float().cpu()
    ref = r16(acc + r16(bias))
    assert O.rel_l2(y, ref) < 3.2e-5
    # gelu (two outputs)
    pre, act = ops.gemm_nt(ad, bd, ops.EPI_GELU, bias=biasd)
    assert O.rel_l2(pre.float().cpu(), ref) < 3.2e-5
    assert O.rel_l2(act.float().cpu(), r16(O.gelu_erf(pre.float().cpu()))) < 2e-3
    # residual fp32
    res = randn((M, N), 6)
    y = ops.gemm_nt(ad, bd, ops.EPI_RESID_F32, bias=biasd, aux=res.to(dev())).cpu()
    assert O.rel_l2(y, res + ref) < 2.8e-5
    # dgelu + column sums
    prez = r16(randn((M, N), 7))
    cs = torch.zeros(N, device=dev())
    y = ops.gemm_nt(ad, bd, ops.EPI_DGELU, aux=prez.to(dev(), BF16), colsum=cs).float().cpu()
    x = prez.clone().requires_grad_(True)
    O.gelu_erf(x).backward(r16(acc))
    assert O.rel_l2(y, r16(x.grad)) < 5.8e-5
    assert O.rel_l2(cs.cpu(), y.sum(0)) < 1.0e-6
    # stored-derivative pair: GELU_DG writes gelu'(pre) (bf16) beside gelu(pre); DMUL multiplies by it as stored
    dgl, act2 = ops.gemm_nt(ad, bd, ops.EPI_GELU_DG, bias=biasd)
    assert torch.equal(act2, act)
    xp = ref.clone().requires_grad_(True)
    O.gelu_erf(xp).backward(torch.ones_like(ref))
    assert O.rel_l2(dgl.float().cpu(), r16(xp.grad)) < 4.5e-5
    assert float((dgl.float().cpu() - xp.grad).abs().max()) < 6e-3          # bf16 rounding of values up to 1.13
    cs2 = torch.zeros(N, device=dev())
    dg_in = r16(randn((M, N), 17, 0.5))
    y2 = ops.gemm_nt(ad, bd, ops.EPI_DMUL, aux=dg_in.to(dev(), BF16), colsum=cs2).float().cpu()
    assert O.rel_l2(y2, r16(r16(acc) * dg_in)) < 2.2e-5
    assert O.rel_l2(cs2.cpu(), y2.sum(0)) < 1.0e-6
    # patch epilogue: row remap + pos add
    n_p, extra = 9, 2
    Bn = 37
    a2 = r16(randn((Bn * n_p, K), 8))
    pos = randn((n_p, N), 9)
    out = torch.full((Bn * (n_p + extra), N), 7.0, device=dev())
    patch_gemm = functools.partial(_ops.gemm_nt, tile=256) if tile == 320 else ops.gemm_nt      # (the 320-row form has no patch epilogue)
    patch_gemm(a2.to(dev(), BF16), bd, ops.EPI_PATCH_F32, bias=biasd, aux=pos.to(dev()), out=out, n_patches=n_p, seq=n_p + extra, extra=extra)
    got = out.cpu().view(Bn, n_p + extra, N)
    want = r16(a2 @ b.t() + r16(bias)).view(Bn, n_p, N) + pos
    assert torch.all(got[:, :extra] == 7.0)
    assert O.rel_l2(got[:, extra:], want) < 7.7e-6


@pytest.mark.parametrize("M,N,K", [(320 * 300 + 77, 2312, 192), (256 * 300 + 5, 1288, 128), (50432, 3072, 1536)])
def test_gemm_nt_seam_form_exact_on_ragged_shapes(hip, M, N, K):
    """The seam form of the persistent kernel (gemm_nt_seam.h: short K loop, >= 3 tiles per CU) on shapes whose last tile row AND last tile
    column are ragged (N % 8 == 0 only): masked rows / columns are out-of-range stores, bias columns past N read as zero, the pipeline
    fill of a tile is requested during the previous tile's epilogue.  Integer operands: exact, and equal to one workgroup per tile."""
    from vitamd import ops
    a, b = ints((M, K), -2, 2, 61), ints((N, K), -2, 2, 62)
    bias = ints((N,), -4, 4, 63)
    ad, bd, biasd = a.to(dev(), BF16), b.to(dev(), BF16), bias.to(dev())
    ref = ((ad.float() @ bd.float().t()) + biasd).to(BF16).float()   # exact in fp32 (|sum| <= 4 K), then the output's one bf16 rounding
    for _ in range(2):                                             # twice: a race would not repeat
        y = ops.gemm_nt(ad, bd, ops.EPI_BIAS_BF16, bias=biasd)
        assert torch.equal(y.float(), ref)
    assert torch.equal(y, ops.gemm_nt(ad, bd, ops.EPI_BIAS_BF16, bias=biasd, tile=512))
    pre, h = ops.gemm_nt(ad, bd, ops.EPI_GELU_DG, bias=biasd)
    pre2, h2 = ops.gemm_nt(ad, bd, ops.EPI_GELU_DG, bias=biasd, tile=512)
    assert torch.equal(pre, pre2) and torch.equal(h, h2)                     # one GELU rounding per model: the table in every launch form
    if N % 256 == 0:
        fac = (ints((M, N), -2, 2, 64) * 0.5).to(dev(), BF16)
        c1, c2 = torch.zeros(N, device=dev()), torch.zeros(N, device=dev())
        d1 = ops.gemm_nt(ad, bd, ops.EPI_DMUL, aux=fac, colsum=c1)
        d2 = ops.gemm_nt(ad, bd, ops.EPI_DMUL, aux=fac, colsum=c2, tile=512)
        assert torch.equal(d1, d2)
        assert torch.equal(d1.float(), ((ad.float() @ bd.float().t()).to(BF16).float() * fac.float()).to(BF16).float())
        assert O.rel_l2(c1.cpu(), c2.cpu()) < 1.0e-6               # column sums: atomics, order differs


@pytest.mark.parametrize("M", [256 * 197, 50000, 320 * 100 + 7])
def test_gemm_nt_tall_tile_exact(hip, M):
    """N = 768 outputs at large M are dispatched to the 320x256-tile kernel (fewer rounds of the 256 CUs); integer operands
    make the bf16 / fp32 results exact, so any mis-mapped row or column shows.  Checked against the 256x256 kernel too."""
    from vitamd import ops
    N, K = 768, 128
    a, b = ints((M, K), -1, 1, 41), ints((N, K), -1, 1, 42)
    bias = ints((N,), -3, 3, 43)
    res = ints((M, N), -7, 7, 44)
    acc = a @ b.t()
    ad, bd = a.to(dev(), BF16), b.to(dev(), BF16)
    y = ops.gemm_nt(ad, bd, ops.EPI_BIAS_BF16, bias=bias.to(dev()))
    assert torch.equal(y.float().cpu(), acc + bias)
    assert torch.equal(y, ops.gemm_nt(ad, bd, ops.EPI_BIAS_BF16, bias=bias.to(dev()), tile=256))
    y = ops.gemm_nt(ad, bd, ops.EPI_RESID_F32, bias=bias.to(dev()), aux=res.to(dev()))
    assert torch.equal(y.cpu(), acc + bias + res)


def test_gemm_nt_tall_tile_gelu_epilogues_match_256(hip):
    """N = 3072 at large M also takes the 320-row tile (a tie in rounds x rows, higher FLOP per staged byte): the GELU / stored
    derivative / multiply epilogues must give bit-identical results to the 256-row kernel (same per-element arithmetic)."""
    from vitamd import ops
    M, N, K = 20380, 3072, 128
    a, b = r16(randn((M, K), 51)).to(dev(), BF16), r16(randn((N, K), 52, 0.1)).to(dev(), BF16)
    bias = randn((N,), 53).to(dev())
    aux = r16(randn((M, N), 54, 0.5)).to(dev(), BF16)
    for epi in (ops.EPI_GELU, ops.EPI_GELU_DG):
        o1, h1 = ops.gemm_nt(a, b, epi, bias=bias, tile=320)
        o2, h2 = ops.gemm_nt(a, b, epi, bias=bias, tile=256)
        assert torch.equal(o1, o2) and torch.equal(h1, h2)
        o0, h0 = ops.gemm_nt(a, b, epi, bias=bias)                    # automatic: the seam kernel (the table from LDS; the others gather it from the device image)
        assert torch.equal(o0, o2) and torch.equal(h0, h2)
        o3, h3 = ops.gemm_nt(a, b, epi, bias=bias, tile=128)          # ... and the small-problem kernel's direct epilogue
        assert torch.equal(o3, o2) and torch.equal(h3, h2)
    for epi in (ops.EPI_DGELU, ops.EPI_DMUL):
        c1, c2 = torch.zeros(N, device=dev()), torch.zeros(N, device=dev())
        y1 = ops.gemm_nt(a, b, epi, aux=aux, colsum=c1)
        y2 = ops.gemm_nt(a, b, epi, aux=aux, colsum=c2, tile=256)
        assert torch.equal(y1, y2)
        assert O.rel_l2(c1.cpu(), c2.cpu()) < 1.0e-6          # column sums: atomics, order differs


def _bf16_rne_exact(v):
    """float64 array -> bf16 bit patterns, nearest-even decided on exact distances (no float32 intermediate rounding)."""
    import numpy as np
    u = v.astype(np.float32).view(np.uint32).astype(np.int64)
    first = (u + 0x7fff + ((u >> 16) & 1)) >> 16
    val = lambda b: (b.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    best, bd = first.copy(), np.abs(val(first) - v)
    for dl in (-1, 1):
        n = first + dl
        dd = np.abs(val(n) - v)
        take = (dd < bd) | ((dd == bd) & (n % 2 == 0) & (best % 2 == 1))
        best, bd = np.where(take, n, best), np.where(take, dd, bd)
    return best.astype(np.int64)


def test_gelu_table_exact_on_every_bf16_input(hip):
    """fc1+GELU on the seam kernel reads gelu / gelu' from a table indexed by the bf16 pre-activation (gemm_nt_seam.h::gelu_lookup8).  A one-hot
    GEMM puts EVERY finite bf16 pattern through it: inside the table (2^-13 <= |x| < 8) both outputs must be the correctly rounded values of
    x Phi(x) and Phi(x) + x phi(x) computed in float64; outside it the formula path must agree to one unit in the last place (or 1e-12)."""
    import numpy as np
    from scipy.special import erfc
    from vitamd import ops
    M, N, K = 65536, 1024, 256
    bits = (torch.arange(N).view(N, 1) % 256) * 256 + torch.arange(K).view(1, K)                     # B[n][k]: pattern (n & 255) << 8 | k
    finite = ((bits >> 7) & 0xff) != 0xff
    b = torch.where(finite, bits, torch.zeros_like(bits)).to(torch.int16).view(BF16)
    a = torch.zeros((M, K), dtype=BF16)
    a[torch.arange(M), torch.arange(M) % K] = 1.0                                                   # out[m][n] = B[n][m & 255]
    dg, g = ops.gemm_nt(a.to(dev()), b.to(dev()), ops.EPI_GELU_DG)
    torch.cuda.synchronize()
    pat = np.where(finite.numpy(), bits.numpy(), 0).astype(np.int64)                                 # [N, K] input pattern of out[k (+256 j)][n]
    x = (pat.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    cdf = 0.5 * erfc(-x * 0.7071067811865476)
    want_g = _bf16_rne_exact(x * cdf)
    want_dg = _bf16_rne_exact(cdf + x * 0.3989422804014327 * np.exp(-0.5 * x * x))
    mag = pat & 0x7fff
    inside = (mag >= 0x3900) & (mag < 0x4100)
    for rows in (slice(0, 256), slice(M - 256, M)):                                                  # first and last 256 rows: every pattern twice
        got_g = (g[rows].cpu().view(torch.int16).numpy().astype(np.int64) & 0xffff).T                # [N, 256] as [n][k]
        got_dg = (dg[rows].cpu().view(torch.int16).numpy().astype(np.int64) & 0xffff).T
        assert np.array_equal(got_g[inside], want_g[inside] & 0xffff) and np.array_equal(got_dg[inside], want_dg[inside] & 0xffff)
        val = lambda q: ((q & 0xffff).astype(np.uint32) << 16).view(np.float32).astype(np.float64)
        for got, want in ((got_g, want_g), (got_dg, want_dg)):
            o = ~inside
            ok = (np.abs((got[o] & 0x7fff) - (want[o] & 0x7fff)) <= 1) & ((got[o] >> 15) == ((want[o] & 0xffff) >> 15))
            ok |= np.abs(val(got[o]) - val(want[o])) <= 1e-12
            assert ok.all()
    assert int(inside.sum()) == 4 * 4096                                                             # N = 4 x 256: each pattern in four columns


def test_seam_probe_and_tile_code_1024(hip):
    """ops.seam_probe times the seam form against the plain persistent form and records the decision PER DEVICE (ops.SEAM_PROBE); tile code 1024
    (persistent, no seams) gives the same bits as the automatic choice and as one workgroup per tile; with the form switched off the host
    wrapper asks for code 1024 for EVERY automatic launch (QKV at batch 128, 891 tiles, sat in the gap of the old pre-filter), and
    the library then plans no seam kernel."""
    from vitamd import ops, lib
    idx = torch.cuda.current_device()
    keep, keep_rec = ops.NT_SEAM, ops.SEAM_PROBE.pop(idx, None)
    try:
        ops.NT_SEAM = None
        assert ops.seam_enabled(dev()) is None
        on = ops.seam_probe(dev(), rows=12288, reps=1)
        rec = ops.SEAM_PROBE[idx]
        assert on == rec["enabled"] == ops.seam_enabled(dev()) and rec["seam_us"] > 0 and rec["plain_us"] > 0
        a, b = r16(randn((256 * 120 + 3, 768), 91)).to(dev(), BF16), r16(randn((2304, 768), 92, 0.05)).to(dev(), BF16)
        bias = randn((2304,), 93).to(dev())
        y0 = ops.gemm_nt(a, b, ops.EPI_BIAS_BF16, bias=bias, tile=0)
        assert torch.equal(y0, ops.gemm_nt(a, b, ops.EPI_BIAS_BF16, bias=bias, tile=1024))
        assert torch.equal(y0, ops.gemm_nt(a, b, ops.EPI_BIAS_BF16, bias=bias, tile=512))
        plan = lib.load().vitamd_gemm_nt_plan
        assert plan(25216, 2304, 768, 2304, ops.EPI_BIAS_BF16, 0) & 0x7f == ops.NT_FORM_LOADER        # the library's own rule: the loader-wave form (an odd number of K-tiles: the seam kernel)
        assert plan(25216, 2304, 704, 2304, ops.EPI_BIAS_BF16, 0) & 0x7f == ops.NT_FORM_SEAM
        ops.SEAM_PROBE[idx] = dict(rec, enabled=False)               # switched off on this device
        assert ops.auto_tile(dev(), 25216, 2304, 768, ops.EPI_BIAS_BF16) == 1024 and ops.auto_tile(dev(), 512, 512, 64, ops.EPI_F32) == 1024
        assert plan(25216, 2304, 768, 2304, ops.EPI_BIAS_BF16, 1024) & 0x7f == 3                      # VITAMD_NT_FORM_PP_PERSISTENT
        assert torch.equal(y0, ops.gemm_nt(a, b, ops.EPI_BIAS_BF16, bias=bias))
        ops.NT_SEAM = True                                           # the process-wide override wins over the per-device record
        assert ops.auto_tile(dev(), 25216, 2304, 768, ops.EPI_BIAS_BF16) == 0
    finally:
        ops.NT_SEAM = keep
        ops.SEAM_PROBE.pop(idx, None)
        if keep_rec is not None:
            ops.SEAM_PROBE[idx] = keep_rec


@pytest.mark.parametrize("M,N,K", [(1000, 264, 128), (256 * 40, 2304, 768), (153600, 256, 128), (50432, 768, 3072), (50000, 3072, 768)])
def test_gemm_nt_loader_form_bit_identical(hip, M, N, K):
    """The loader-wave form (tile code 2048, gemm_nt_ld.h: eight compute waves that issue no vector-memory instruction in the K loop + four
    loader waves, K-half phases) against gemm_nt_pp_kernel on one workgroup per tile: RANDOM data, every epilogue of the form, ragged M and N,
    one and several tiles per workgroup, twice (a race would not repeat) - bit for bit (same per-accumulator k order, same GELU table)."""
    from vitamd import ops, lib
    a, b = r16(randn((M, K), 7)).to(dev(), BF16), r16(randn((N, K), 8, 0.05)).to(dev(), BF16)
    bias = randn((N,), 9).to(dev())
    aux = r16(randn((M, N), 10)).to(dev(), BF16)
    assert lib.load().vitamd_gemm_nt_plan(M, N, K, N, ops.EPI_BIAS_BF16, 2048) & 0x7f == 5           # VITAMD_NT_FORM_LOADER
    for epi in (ops.EPI_BIAS_BF16, ops.EPI_GELU_DG, ops.EPI_GELU, ops.EPI_DMUL):
        if epi == ops.EPI_DMUL and N % 256:
            continue
        for rep in range(2):
            outs = {}
            for t in (256, 2048):
                cs = torch.zeros(N, device=dev())
                kw = dict(bias=bias) if epi != ops.EPI_DMUL else dict(aux=aux, colsum=cs)
                o = ops.gemm_nt(a, b, epi, tile=t, **kw)
                outs[t] = list(o if isinstance(o, tuple) else (o,)) + ([cs] if epi == ops.EPI_DMUL else [])
            for x, y in zip(outs[256], outs[2048]):
                if x.dtype == F32:
                    assert O.rel_l2(x.cpu(), y.cpu()) < 1.0e-5      # column sums: atomics, order differs
                else:
                    assert torch.equal(x, y), (epi, rep)
    with pytest.raises(lib.VitamdError):
        ops.gemm_nt(a[:, :64].contiguous(), b[:, :64].contiguous(), ops.EPI_BIAS_BF16, bias=bias, tile=2048)      # an odd number of K-tiles: refused, not mis-computed


def test_gelu_needs_vitamd_init_and_init_is_idempotent(hip):
    """C-ABI contract (include/vitamd.h): vitamd_init is the only entry point that allocates; it is idempotent; the host wrapper calls it before
    the first GELU launch on a device."""
    from vitamd import ops, lib
    L = lib.load()
    idx = torch.cuda.current_device()
    assert L.vitamd_init(idx, torch.cuda.current_stream().cuda_stream) == 0
    assert L.vitamd_init(-1, 0) == 0 and L.vitamd_init(idx, 0) == 0
    a, b = r16(randn((300, 128), 1)).to(dev(), BF16), r16(randn((256, 128), 2)).to(dev(), BF16)
    pre, act = ops.gemm_nt(a, b, ops.EPI_GELU)
    assert idx in ops._INITIALISED and torch.isfinite(act.float()).all()
    assert L.vitamd_init(99, 0) != 0                                 # no such device


def test_gemm_nt_rejects_bad_shapes(hip):
    from vitamd import ops, lib
    a = torch.zeros((64, 100), device=dev(), dtype=BF16)
    b = torch.zeros((64, 100), device=dev(), dtype=BF16)
    with pytest.raises(lib.VitamdError):
        ops.gemm_nt(a, b, ops.EPI_F32)           # K % 64 != 0
    with pytest.raises(lib.VitamdError):
        ops.gemm_nt(a.cpu(), b, ops.EPI_F32)     # host tensor


# ------------------------------------------------------------------------------------------ gemm_tn
@pytest.mark.parametrize("R,P,Q,splits", [(64, 256, 256, 1), (128, 256, 256, 2), (1000, 512, 768, 0), (111, 128, 64, 0),
                                          (197 * 8, 2304, 768, 0), (320, 1536, 512, 3), (70, 16, 512, 1), (4096, 768, 3072, 0)])
def test_gemm_tn_exact_integers(hip, R, P, Q, splits):
    from vitamd import ops
    l = ints((R, P), -2, 2, 11)
    r = ints((R, Q), -3, 3, 12)
    init = ints((P, Q), -5, 5, 13)
    ref = init + l.t() @ r
    ld, rd = l.to(dev(), BF16), r.to(dev(), BF16)
    for atomic in (True, False):                 # fp32 atomics / workspace + reduce pass
        out = init.to(dev())
        ops.gemm_tn(ld, rd, out, splits=splits, atomic=atomic)
        torch.cuda.synchronize()
        assert torch.equal(out.cpu(), ref), atomic
    out = torch.full((P, Q), 123.0, device=dev())
    ops.gemm_tn(ld, rd, out, splits=splits, accumulate=False)   # overwrite mode needs no zeroing
    assert torch.equal(out.cpu(), l.t() @ r)


@pytest.mark.parametrize("R,P,Q,splits", [(50432, 768, 3072, 7), (50432, 2304, 768, 14), (4100, 520, 264, 0), (70, 256, 256, 0), (197 * 64, 3072, 768, 5)])
def test_gemm_tn_exclusive_form_equals_shared_form(hip, R, P, Q, splits):
    """VITAMD_TN_FORM_EXCLUSIVE (12 waves: four loader waves issue every LDS-DMA request, eight compute waves only read LDS and multiply) walks the
    same ring in the same order as the 8-wave kernel: bit-identical outputs, ragged P / Q / R and both accumulate modes included."""
    from vitamd import ops
    l, r = r16(randn((R, P), 81)).to(dev(), BF16), r16(randn((R, Q), 82)).to(dev(), BF16)
    for acc in (False, True):
        o0 = torch.full((P, Q), 3.0, device=dev()); o1 = torch.full((P, Q), 3.0, device=dev())
        ops.gemm_tn(l, r, o0, splits=splits, accumulate=acc, form=ops.TN_FORM_SHARED)
        ops.gemm_tn(l, r, o1, splits=splits, accumulate=acc, form=ops.TN_FORM_EXCLUSIVE)
        assert torch.equal(o0, o1)
    if R <= 5000:
        assert O.rel_l2(o0.cpu() - 3.0, l.float().cpu().t() @ r.float().cpu()) < 2.0e-6
    st = torch.cuda.current_stream().cuda_stream
    assert hip.vitamd_gemm_tn_bf16_ws(l.data_ptr(), r.data_ptr(), o0.data_ptr(), R, P, Q, P, Q, Q, 0, None, 0, 0, 2, st) == 2      # unknown form


@pytest.mark.parametrize("R,P,Q", [(50432, 768, 3072), (50432, 3072, 768), (50432, 2304, 768)])
def test_gemm_tn_exact_at_the_step_reduction_length_with_the_step_splits(hip, R, P, Q):
    """VERDICT r2 item 4-iii / ADVICE r2: the weight-gradient GEMMs of the ViT-B step reduce over R = 50 432 rows with the split-K factors of
    functions._tn_splits (7 / 9 at 252 workgroups, 4 / 5 at 128) and OVERWRITE an uninitialised gradient buffer.  Integer-valued
    operands make every partial sum exact in fp32, so the result must equal the integer product bit for bit, and a NaN-filled `out`
    proves that the split-K reduce pass writes every element."""
    from vitamd import ops, functions as F
    l = ints((R, P), -2, 2, 81).to(dev(), BF16)
    r = ints((R, Q), -2, 2, 82).to(dev(), BF16)
    ref = (l.double().t() @ r.double()).float()            # |sum| <= 4 R < 2^24: exact in fp32 too; fp64 product on the device as the yardstick
    assert float(ref.abs().max()) < 2 ** 24
    try:
        for target in (252, 128):
            F.TN_TARGET_WGS = target
            out = torch.full((P, Q), float("nan"), device=dev())
            splits = F._tn_splits(out)
            assert splits == max(1, round(target / (((P + 255) // 256) * ((Q + 255) // 256))))
            ops.gemm_tn(l, r, out, accumulate=False, splits=splits)
            torch.cuda.synchronize()
            assert torch.equal(out, ref), (target, splits)
    finally:
        F.TN_TARGET_WGS = None


def test_gemm_tn_overwrite_mode_without_workspace_is_refused(hip):
    """with accumulate = 0 and no (or a too small) split-K workspace the call used to fall through to the atomic kernel,
    which ADDS into an un-zeroed `out`; it must fail loudly instead."""
    from vitamd import ops
    l, r = ints((128, 256), -2, 2, 71).to(dev(), BF16), ints((128, 256), -2, 2, 72).to(dev(), BF16)
    out = torch.full((256, 256), 5.0, device=dev())
    st = torch.cuda.current_stream().cuda_stream
    assert hip.vitamd_gemm_tn_bf16_ws(l.data_ptr(), r.data_ptr(), out.data_ptr(), 128, 256, 256, 256, 256, 256, 0, None, 0, 0, 0, st) == 2
    small = torch.empty(16, device=dev())
    assert hip.vitamd_gemm_tn_bf16_ws(l.data_ptr(), r.data_ptr(), out.data_ptr(), 128, 256, 256, 256, 256, 256, 0, small.data_ptr(), 64, 0, 0, st) == 2
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), torch.full((256, 256), 5.0))          # untouched


def test_experimental_library_alternatives(hip):
    """What libvitamd_exp.so (make EXPERIMENTAL=1) carries beyond the product - explicit seam-kernel codes, the loader kernel's first request
    schedule, the round-3 weight-gradient loaders, the split-role and plain-loop attention backward - is checked bit for bit against production by
    tools/check_experimental.py, in a child process so that this process only ever maps the production library."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "vit-is-all-you-need_amd", "vitamd", "libvitamd_exp.so")):
        pytest.skip("libvitamd_exp.so not built (make EXPERIMENTAL=1)")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_experimental.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


# ------------------------------------------------------------------------------------------ layernorm
@pytest.mark.parametrize("M,D", [(1, 768), (777, 768), (320, 512), (111, 128), (50, 1024), (33, 200)])
def test_layernorm_fwd_bwd(hip, M, D):
    from vitamd import ops
    x = randn((M, D), 21, 2.0) + 0.5
    add = r16(randn((M, D), 22))
    xs, y, mean, rstd = ops.layernorm_fwd(x.to(dev()), addend=add.to(dev(), BF16))
    xr = (x + add).requires_grad_(True)
    yr = O.layer_norm(xr)
    assert O.rel_l2(xs.cpu(), xr) < 1e-6
    assert O.rel_l2(y.float().cpu(), r16(yr)) < 2.9e-5
    assert O.rel_l2(mean.cpu(), xr.mean(-1)) < 1.0e-6
    _, y2, _, _ = ops.layernorm_fwd(x.to(dev()))
    assert O.rel_l2(y2.float().cpu(), r16(O.layer_norm(x))) < 3.0e-5
    dy = r16(randn((M, D), 23))
    gres = randn((M, D), 24)
    cs = torch.zeros(D, device=dev())
    g, gb = ops.layernorm_bwd(dy.to(dev(), BF16), xs, mean, rstd, g_res=gres.to(dev()), want_bf16=True, colsum=cs)
    yr.backward(dy)
    want = gres + xr.grad
    assert O.rel_l2(g.cpu(), want) < 1.0e-6
    assert torch.equal(gb.float().cpu(), r16(g.cpu()))
    assert O.rel_l2(cs.cpu(), gb.float().cpu().sum(0)) < 1.0e-6
    g2, none = ops.layernorm_bwd(dy.to(dev(), BF16), xs, mean, rstd)
    assert none is None and O.rel_l2(g2.cpu(), xr.grad) < 1.0e-6


@pytest.mark.parametrize("M,D", [(777, 768), (320, 512), (50, 1024), (5, 256)])
def test_layernorm_bwd_xhat_mode(hip, M, D):
    """LayerNorm backward reading xhat from the forward's bf16 output instead of recomputing it from fp32 x: the bf16 rounding of
    xhat only enters the xhat * mean(dy * xhat) term, so the result must stay within 1e-3 of the recomputing kernel and of the
    fp32 formula (it is ~2e-4 in practice); the bf16 copy and its column sums follow."""
    from vitamd import ops
    x = (randn((M, D), 71, 2.0) + 0.3).to(dev())
    dy = r16(randn((M, D), 72)).to(dev(), BF16)
    gres = randn((M, D), 73).to(dev())
    _, y, mean, rstd = ops.layernorm_fwd(x)
    c1, c2 = torch.zeros(D, device=dev()), torch.zeros(D, device=dev())
    g_ref, gb_ref = ops.layernorm_bwd(dy, x, mean, rstd, g_res=gres, want_bf16=True, colsum=c1)
    g, gb = ops.layernorm_bwd(dy, x, mean, rstd, g_res=gres, want_bf16=True, colsum=c2, xhat=y)
    assert O.rel_l2(g.cpu(), g_ref.cpu()) < 7.2e-5
    assert O.rel_l2(gb.float().cpu(), gb_ref.float().cpu()) < 5.3e-4          # bf16 roundings may flip the last bit
    assert O.rel_l2(c2.cpu(), gb.float().sum(0).cpu()) < 1.0e-6
    xr = x.cpu().clone().requires_grad_(True)
    O.layer_norm(xr).backward(dy.float().cpu())
    assert O.rel_l2(g.cpu(), xr.grad + gres.cpu()) < 7.2e-5


# ------------------------------------------------------------------------------------------ attention
def _attn_ref(qkv, B, N, H, causal, d_o=None):
    t = qkv.view(B, N, 3 * H * 64).clone().requires_grad_(True)
    q, k, v = O.split_qkv(t, H)
    o = O.sdpa(q, k, v, causal, lowp=True).permute(0, 2, 1, 3).reshape(B * N, H * 64)
    if d_o is None:
        return o.detach(), None
    o.backward(d_o)
    return o.detach(), t.grad.view(B * N, -1)


@pytest.mark.parametrize("B,N,H,causal", [(2, 5, 2, False), (1, 32, 1, False), (3, 37, 2, True), (2, 197, 3, False),
                                          (1, 288, 2, False), (2, 64, 2, True), (1, 197, 2, True), (1, 512, 1, False),
                                          # every tile count of the pipelined backward kernels (33 <= N <= 256, non-causal), ragged and exact
                                          (2, 33, 1, False), (1, 64, 2, False), (2, 96, 1, False), (1, 130, 2, False), (1, 160, 1, False),
                                          (1, 224, 1, False), (2, 256, 2, False)])
def test_attention_fwd_bwd(hip, B, N, H, causal):
    from vitamd import ops
    qkv = r16(randn((B * N, 3 * H * 64), 31 + N, 1.5))
    d_o = r16(randn((B * N, H * 64), 32 + N))
    o_ref, dqkv_ref = _attn_ref(qkv, B, N, H, causal, d_o)
    qd = qkv.to(dev(), BF16)
    o, lse = ops.attention_fwd(qd, B, N, H, causal)
    assert O.rel_l2(o.float().cpu(), o_ref) < 2.2e-3
    # log-sum-exp (log2 domain) against the definition
    q, k, _ = O.split_qkv(qkv.view(B, N, -1), H)
    s = (q @ k.transpose(-1, -2)) * 0.125
    if causal:
        s = s.masked_fill(torch.triu(torch.ones(N, N, dtype=torch.bool), 1), float("-inf"))
    assert O.rel_l2(lse.cpu(), torch.logsumexp(s, -1) / math.log(2.0)) < 1.0e-6
    dbias = torch.full((3 * H * 64,), 1.0, device=dev())
    dqkv = ops.attention_bwd(qd, o, lse, d_o.to(dev(), BF16), B, N, H, causal, dbias=dbias).float().cpu()
    assert O.rel_l2(dbias.cpu() - 1.0, dqkv.sum(0)) < 1.0e-6     # fused QKV-bias gradient = column sums of what was stored
    D = H * 64
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        assert O.rel_l2(dqkv[:, sl], dqkv_ref[:, sl]) < 6.4e-3, name


def test_attention_against_the_reference_sdpa_bf16_floor(hip):
    """tests/golden/sdpa_b197.pt holds F.scaled_dot_product_attention as the reference calls it (transformer.py:27-28) at
    the ViT-B head geometry (B 2, N 197, H 12), forward and dq / dk / dv, in fp32 and in bf16 (what the reference's SDPA computes under
    autocast), from the same bf16-representable inputs; `ref_bf16_floor` = the reference's own bf16-vs-fp32 distance (o 1.9e-3,
    dq 2.8e-3, dk 3.5e-3, dv 3.3e-3).  north_star's 1e-3 is below that floor for this op, so the HIP kernels are held to the floor:
    distance to the fp32 result <= 1.5 x the reference's own."""
    import weights as W
    from vitamd import ops
    from conftest import load_golden
    g = load_golden("sdpa_b197.pt")
    c = g["config"]
    B, N, H, D = c["B"], c["N"], c["H"], c["H"] * c["head_dim"]
    qkv = (W.normal(c["seed"], "qkv", (B, N, 3 * D)) * c["input_scale"]).bfloat16().view(B * N, 3 * D).to(dev())
    d_o = W.normal(c["seed"], "d_o", (B, N, D)).bfloat16().view(B * N, D).to(dev())
    o, lse = ops.attention_fwd(qkv, B, N, H, False)
    dqkv = ops.attention_bwd(qkv, o, lse, d_o, B, N, H, False).float().cpu()
    got = {"o": o.float().cpu(), "dq": dqkv[:, :D].contiguous(), "dk": dqkv[:, D:2 * D].contiguous(), "dv": dqkv[:, 2 * D:].contiguous()}
    for k, t in got.items():
        s = t.flatten()[::c["sample_stride"]]
        e32, e16 = O.rel_l2(s, g["fp32"][k]), O.rel_l2(s, g["bf16"][k])
        floor = g["ref_bf16_floor"][k]
        assert e32 < 1.5 * floor, (k, e32, floor)            # no further from fp32 than 1.5 x the reference's own bf16 SDPA
        assert e16 < 2.0 * floor, (k, e16, floor)            # two independent bf16 flows of one fp32 function


@pytest.mark.parametrize("B,N,H,causal", [(1, 513, 1, False), (2, 577, 2, False), (1, 1024, 2, True), (1, 640, 1, True), (1, 1500, 1, False)])
def test_attention_long_sequences(hip, B, N, H, causal):
    """N > 512: the two-sided-tiled kernels (K/V or Q/dO streamed through LDS in 512-row chunks, online softmax state and
    the dQ / dK / dV accumulators carried across chunks).  Same checks as the single-chunk kernels."""
    from vitamd import ops
    qkv = r16(randn((B * N, 3 * H * 64), 131 + N, 1.5))
    d_o = r16(randn((B * N, H * 64), 132 + N))
    o_ref, dqkv_ref = _attn_ref(qkv, B, N, H, causal, d_o)
    qd = qkv.to(dev(), BF16)
    o, lse = ops.attention_fwd(qd, B, N, H, causal)
    assert O.rel_l2(o.float().cpu(), o_ref) < 2.3e-3
    q, k, _ = O.split_qkv(qkv.view(B, N, -1), H)
    s = (q @ k.transpose(-1, -2)) * 0.125
    if causal:
        s = s.masked_fill(torch.triu(torch.ones(N, N, dtype=torch.bool), 1), float("-inf"))
    assert O.rel_l2(lse.cpu(), torch.logsumexp(s, -1) / math.log(2.0)) < 1.0e-6
    dbias = torch.zeros((3 * H * 64,), device=dev())
    dqkv = ops.attention_bwd(qd, o, lse, d_o.to(dev(), BF16), B, N, H, causal, dbias=dbias).float().cpu()
    assert O.rel_l2(dbias.cpu(), dqkv.sum(0)) < 1.0e-6
    D = H * 64
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        assert O.rel_l2(dqkv[:, sl], dqkv_ref[:, sl]) < 5.3e-3, name


@pytest.mark.parametrize("B,N,H,causal", [(2, 197, 3, False), (1, 256, 2, True), (3, 37, 2, False), (2, 5, 1, False)])
def test_attention_fwd_fused_residual(hip, B, N, H, causal):
    """Forward with the layer's residual add in its epilogue (N <= 256): o and lse identical to the plain kernel, and
    x1 = x0 + o bit-exact against the same sum done by torch (fp32 add of the bf16-rounded o)."""
    from vitamd import ops, lib
    qkv = r16(randn((B * N, 3 * H * 64), 231 + N, 1.5)).to(dev(), BF16)
    x0 = randn((B * N, H * 64), 232 + N, 2.0).to(dev())
    o_ref, lse_ref = ops.attention_fwd(qkv, B, N, H, causal)
    o, lse, x1 = ops.attention_fwd(qkv, B, N, H, causal, resid=x0)
    assert torch.equal(o, o_ref) and torch.equal(lse, lse_ref)
    assert torch.equal(x1, x0 + o_ref.float())
    with pytest.raises(lib.VitamdError):
        ops.attention_fwd(torch.zeros((300, 192), device=dev(), dtype=BF16), 1, 300, 1, False, resid=torch.zeros((300, 64), device=dev()))


def test_attention_softmax_spike(hip):
    """A key that dominates late in the sequence forces the online-softmax rescale branch."""
    from vitamd import ops
    B, N, H = 1, 197, 1
    qkv = r16(randn((B * N, 3 * 64), 77, 0.5))
    qkv[150, 64:128] = r16(qkv[10, 0:64] * 40.0)  # key 150 aligned with query 10
    o_ref, _ = _attn_ref(qkv, B, N, H, False)
    o, _ = ops.attention_fwd(qkv.to(dev(), BF16), B, N, H, False)
    # the 8-wave forward rounds P to bf16 against the RUNNING maximum (online softmax), the reference against the final one: same bf16 floor as
    # test_attention_fwd_bwd for the ordinary rows; the spiked row must come out as the value row of the dominating key (its earlier tiles are
    # rescaled by exp2(-huge) = 0)
    assert O.rel_l2(o.float().cpu(), o_ref) < 2.2e-3
    assert O.rel_l2(o[10].float().cpu(), qkv[150, 128:192]) < 1.0e-6
    # causal keeps the 4-wave kernel with the register-resident score row, whose P rounding the reference restates exactly on a one-hot row
    o_c, _ = ops.attention_fwd(qkv.to(dev(), BF16), B, N, H, True)
    assert O.rel_l2(o_c.float().cpu(), _attn_ref(qkv, B, N, H, True)[0]) < 2.2e-3


def test_batched_weight_cast_vector_and_scalar_paths(hip):
    """One launch casts (and transposes) every weight of a model: the 16-B path (N, K multiples of 4) and the scalar tail path
    must both equal torch's own rounding, tile edges included."""
    from vitamd.functions import WeightCache
    ws = [randn(s, 60 + i).to(dev()) for i, s in enumerate(((768, 768), (2304, 768), (1000, 768), (100, 36), (70, 130), (33, 7), (64, 64), (4, 4)))]
    for want_t in (True, False):
        cache = WeightCache()
        cache.prepare(ws, want_t)
        torch.cuda.synchronize()
        for w in ws:
            wb, wbt = cache.get(w, want_t)
            assert torch.equal(wb, w.to(BF16))
            assert (wbt is None) == (not want_t)
            if want_t:
                assert torch.equal(wbt, w.t().contiguous().to(BF16))


# ------------------------------------------------------------------------------------------ helpers
def test_cast_transpose_im2col_colsum_embed(hip):
    from vitamd import ops
    w = randn((1000, 768), 41)
    wb, wbt = ops.cast_weight(w.to(dev()), True, True)
    assert torch.equal(wb.float().cpu(), r16(w)) and torch.equal(wbt.float().cpu(), r16(w).t())
    x = randn((100003,), 42)
    assert torch.equal(ops.cast_bf16(x.to(dev())).float().cpu(), r16(x))
    for (B, C, H, W, p) in ((3, 3, 32, 32, 16), (2, 3, 224, 224, 16), (2, 8, 5, 1, 1), (1, 3, 24, 36, 12)):
        img = randn((B, C, H, W), 43)
        assert torch.equal(ops.im2col(img.to(dev()), p).float().cpu(), r16(O.patchify(img, p)).reshape(-1, C * p * p))
    m = r16(randn((5000, 770), 44))
    assert O.rel_l2(ops.colsum(m.to(dev(), BF16)).cpu(), m.sum(0)) < 1e-5
    for B, seq, extra, D in ((7, 11, 2, 256), (40, 197, 1, 768), (3, 9, 0, 512), (5, 6, 2, 66)):      # 66: the scalar path (D % 4 != 0)
        g = randn((B * seq, D), 45 + D)
        dpos, dextra, dyp, dbias = ops.embed_bwd(g.to(dev()), B, seq, extra, D)
        g3 = g.view(B, seq, D)
        assert O.rel_l2(dpos.cpu(), g3[:, extra:].sum(0)) < 1e-6
        if extra:
            assert O.rel_l2(dextra.cpu(), g3[:, :extra].sum(0)) < 1e-6
        assert torch.equal(dyp.float().cpu(), r16(g3[:, extra:]).reshape(-1, D))
        assert O.rel_l2(dbias.cpu(), r16(g3[:, extra:]).sum((0, 1))) < 1.0e-6
