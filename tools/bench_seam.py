"""Seam kernel (csrc/gemm_nt_seam.h) against the production choice on the per-layer NT launches of ViT-B (M = 50 432): interleaved rounds in
one process, random data, medians; outputs must be bit-identical (same accumulation order, same epilogue arithmetic).
usage: bench_seam.py [rounds]   (experimental library: explicit tile codes 24 = seam kernel on 256-row tiles, 25 = on 320-row tiles)"""
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
calls = [
    ("qkv", lambda t: ops.gemm_nt(x1, wqkv, ops.EPI_BIAS_BF16, bias=b3, tile=t), 2.0 * M * D * 3 * D, (0, 1024, 24, 25, 30, 2048)),
    ("fc1+gelu", lambda t: ops.gemm_nt(x1, w1, ops.EPI_GELU_DG, bias=b4, tile=t), 2.0 * M * D * 4 * D, (0, 1024, 24, 2048)),            # (320-row seam kernel: plain-bias epilogue only)
    ("dgrad_fc2", lambda t: ops.gemm_nt(x1, w2_t, ops.EPI_DMUL, aux=x4, colsum=torch.zeros(4 * D, device=dev), tile=t), 2.0 * M * D * 4 * D, (0, 1024, 24, 2048)),
    ("dgrad_fc1", lambda t: ops.gemm_nt(x4, w1_t, ops.EPI_BIAS_BF16, tile=t), 2.0 * M * D * 4 * D, (0, 1024, 24, 25, 30, 2048)),
    ("dgrad_qkv", lambda t: ops.gemm_nt(x3, wqkv_t, ops.EPI_BIAS_BF16, tile=t), 2.0 * M * D * 3 * D, (0, 1024, 24, 25, 30, 2048)),
]
names = {0: "auto", 512: "one-wg-per-tile", 1024: "persistent-no-seam", 24: "seam256", 25: "seam320", 30: "seam256+table", 2048: "loader-waves"}
only = os.environ.get("SHAPES")
for name, fn, fl, tiles in calls:
    if only and name not in only.split(","): continue
    ref = None
    res = {t: [] for t in tiles}
    for rnd in range(rounds):
        for t in tiles:
            if rnd == 0:
                out = fn(t); torch.cuda.synchronize()
                o = [x.float().clone() for x in (out if isinstance(out, tuple) else (out,))]
                if ref is None: ref = o
                else:
                    same = all(torch.equal(a, b) for a, b in zip(o, ref))
                    print(f"  {name} {names[t]}: " + ("bit-identical" if same else "MISMATCH rel %.3e / %.3e" % tuple(float((a - b).norm() / b.norm()) for a, b in zip((o + o)[:2], (ref + ref)[:2]))), flush=True)
                del out, o
            fn(t)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); s.record()
            for _ in range(10): fn(t)
            e.record(); torch.cuda.synchronize()
            res[t].append(s.elapsed_time(e) / 10 * 1e3)
    for t in tiles:
        med = statistics.median(res[t])
        print(f"{name:10s} {names[t]:20s} {med:7.1f} us  {fl / med / 1e6:7.1f} TF  {['%.0f' % v for v in res[t]]}", flush=True)
