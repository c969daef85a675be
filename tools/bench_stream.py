"""Stream-kernel experiments (gemm_nt_stream.h) on the six per-layer NT launches of ViT-B (M = 50 432): interleaved rounds in one
process, random data, medians.  Variants per shape:
  prod        production automatic choice (persistent ping-pong kernel, 320- or 256-row tiles)
  s8/s6/s10   stream kernel, 256/192/320-row tiles, burst epilogue (s8 must be bit-identical to the production 256-row kernel)
  *_nost      the same without any output store (timing only: the floor that perfect overlap could reach)
  *_trk       no epilogue stores, but the same number of 16-B-per-lane store instructions issued one per phase from inside the
              main loop to the PREVIOUS tile's output rows (timing only): what trickled stores cost beside the main loop
  *_trkr      the same instruction stream with every store aimed at one tile's rows (L2-resident: no HBM write traffic)
  *_trkd      (DIRECT=1) accumulator-layout addresses (16 rows x 64 B per instruction) instead of whole 128-B row segments
  sp8/sp6     split roles: wave row 0 issues all LDS-DMA and does the counted waits, wave row 1 carries the store slots and never waits
usage: bench_stream.py [rounds]"""
import os, sys, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import lib as _explib; _explib.use_experimental()
from vitamd import ops
_L = _explib.load()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda")
M, D = 256 * 197, 768
g = torch.Generator(device="cpu").manual_seed(1)
def rb(*s, scale=1.0): return (torch.randn(*s, generator=g) * scale).to(dev, torch.bfloat16)
x1, x3, x4 = rb(M, D), rb(M, 3 * D), rb(M, 4 * D)
wqkv, w1, w2 = rb(3 * D, D, scale=0.03), rb(4 * D, D, scale=0.03), rb(D, 4 * D, scale=0.03)
wqkv_t, w1_t, w2_t = rb(D, 3 * D, scale=0.03), rb(D, 4 * D, scale=0.03), rb(4 * D, D, scale=0.03)
b3, b4, b1 = torch.randn(3 * D, device=dev), torch.randn(4 * D, device=dev), torch.randn(D, device=dev)
res_in = torch.randn(M, D, device=dev)
cs = torch.zeros(4 * D, device=dev)
NOST = 1 << 16
def variants(outs_per_row_mt):          # outs_per_row_mt: live stores per wave and tile = this x MT
    v = {"prod": (0, 0)}
    sel = os.environ.get("VARIANTS", "s8,sp8").split(",")
    for name, code, tcode, mt in (("s8", 30, 33, 8), ("s6", 31, 34, 6), ("s10", 32, 35, 10), ("sp8", 36, 36, 8), ("sp6", 37, 37, 6)):
        if name not in sel: continue
        ns = outs_per_row_mt * mt
        v[name] = (code, 0)
        v[name + "_nost"] = (code, NOST)
        v[name + "_trk"] = (tcode, NOST | (ns << 8))                    # dummy stores to the previous tile's rows (HBM write traffic)
        v[name + "_trkr"] = (tcode, NOST | (ns << 8) | (1 << 25))       # the same instruction stream, stores to L2-resident rows
        if name == "s8" and os.environ.get("BISECT"):
            v["s8_trk_oob"] = (tcode, NOST)                                   # every store slot out of range
            v["s8_trk_plain"] = (tcode, NOST | (ns << 8) | (1 << 26))         # default cache policy
            v["s8_trk_skip"] = (tcode, NOST | (ns << 8) | (2 << 26))          # no instruction when nothing is due
        if name == "s8" and os.environ.get("LENIENT"):
            for ln, c in ((4, 38), (8, 39), (12, 40)): v["s8_trk_mid%d" % ln] = (c, NOST | (ns << 8))     # store slot behind the k-th MFMA of the matrix section
        if outs_per_row_mt == 2 and os.environ.get("DIRECT"): v[name + "_trkd"] = (tcode, NOST | (ns << 8) | (1 << 24))
    return v
calls = [
    ("qkv", lambda t: ops.gemm_nt(x1, wqkv, ops.EPI_BIAS_BF16, bias=b3, tile=t), 2.0 * M * D * 3 * D, 2),
    ("fc1+gelu", lambda t: ops.gemm_nt(x1, w1, ops.EPI_GELU_DG, bias=b4, tile=t), 2.0 * M * D * 4 * D, 4),
    ("fc2+resid", lambda t: ops.gemm_nt(x4, w2, ops.EPI_RESID_F32, bias=b1, aux=res_in, tile=t), 2.0 * M * D * 4 * D, 4),
    ("dgrad_fc2", lambda t: ops.gemm_nt(x1, w2_t, ops.EPI_DMUL, aux=x4, colsum=cs, tile=t), 2.0 * M * D * 4 * D, 2),
    ("dgrad_fc1", lambda t: ops.gemm_nt(x4, w1_t, ops.EPI_BIAS_BF16, tile=t), 2.0 * M * D * 4 * D, 2),
    ("dgrad_qkv", lambda t: ops.gemm_nt(x3, wqkv_t, ops.EPI_BIAS_BF16, tile=t), 2.0 * M * D * 3 * D, 2),
]
only = os.environ.get("SHAPES")
tot = {}
for name, fn, fl, opr in calls:
    if only and name not in only.split(","): continue
    cfgs = variants(opr)
    ref = None
    res = {k: [] for k in cfgs}
    for rnd in range(rounds):
        for k, (t, dbg) in cfgs.items():
            _L.vitamd_set_debug(dbg)
            if rnd == 0 and dbg == 0:
                out = fn(t); torch.cuda.synchronize()
                o = [x.float().clone() for x in (out if isinstance(out, tuple) else (out,))]
                if ref is None: ref = o
                elif not all(torch.equal(a, b) for a, b in zip(o, ref)):
                    print(f"  {name} {k}: MISMATCH rel {float((o[-1] - ref[-1]).norm() / ref[-1].norm()):.3e}", flush=True)
                else: print(f"  {name} {k}: bit-identical to prod", flush=True)
                del out, o
            fn(t)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); s.record()
            for _ in range(10): fn(t)
            e.record(); torch.cuda.synchronize()
            res[k].append(s.elapsed_time(e) / 10 * 1e3)
            _L.vitamd_set_debug(0)
    for k in cfgs:
        med = statistics.median(res[k]); tot[k] = tot.get(k, 0.0) + med
        print(f"{name:10s} {k:10s} {med:7.1f} us  {fl / med / 1e6:7.1f} TF  {['%.0f' % v for v in res[k]]}", flush=True)
print({k: round(v, 1) for k, v in tot.items()}, "us per layer (sum over the shapes run)")
