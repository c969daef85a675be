"""Loader-wave NT GEMM (tile code 2048, csrc/gemm_nt_ld.h) against the automatic choice (gemm_nt_pp_kernel / gemm_nt_seam_kernel) on the six
per-layer launches of ViT-B (M = 50 432, plain-bias fc2 as inside the stack): interleaved rounds in one process, random data, medians; outputs
compared bit for bit (same per-accumulator k order; GELU: table in both forms).
usage: bench_ld.py [rows]"""
import os, sys, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import ops, lib as _lib
EXP = "--exp" in sys.argv            # experimental library: the burst schedule (tile 2049) and the timing-only ablations (dbg bits 0 / 1, results garbage)
if EXP: _lib.use_experimental()
_args = [a for a in sys.argv[1:] if not a.startswith("--")]
dev = torch.device("cuda")
M, D = (int(_args[0]) if _args else 256 * 197), 768
g = torch.Generator(device="cpu").manual_seed(1)
def rb(*s, scale=1.0): return (torch.randn(*s, generator=g) * scale).to(dev, torch.bfloat16)
x1, x3, x4 = rb(M, D), rb(M, 3 * D), rb(M, 4 * D)
wqkv, w1, w2 = rb(3 * D, D, scale=0.03), rb(4 * D, D, scale=0.03), rb(D, 4 * D, scale=0.03)
wqkv_t, w1_t, w2_t = rb(D, 3 * D, scale=0.03), rb(D, 4 * D, scale=0.03), rb(4 * D, D, scale=0.03)
b3, b4, b1 = torch.randn(3 * D, device=dev), torch.randn(4 * D, device=dev), torch.randn(D, device=dev)
cs = torch.zeros(4 * D, device=dev)
calls = [
    ("qkv", lambda t: ops.gemm_nt(x1, wqkv, ops.EPI_BIAS_BF16, bias=b3, tile=t), 2.0 * M * D * 3 * D),
    ("fc1+gelu", lambda t: ops.gemm_nt(x1, w1, ops.EPI_GELU_DG, bias=b4, tile=t), 2.0 * M * D * 4 * D),
    ("fc2", lambda t: ops.gemm_nt(x4, w2, ops.EPI_BIAS_BF16, bias=b1, tile=t), 2.0 * M * D * 4 * D),
    ("dgrad_fc2", lambda t: ops.gemm_nt(x1, w2_t, ops.EPI_DMUL, aux=x4, colsum=cs.zero_(), tile=t), 2.0 * M * D * 4 * D),
    ("dgrad_fc1", lambda t: ops.gemm_nt(x4, w1_t, ops.EPI_BIAS_BF16, tile=t), 2.0 * M * D * 4 * D),
    ("dgrad_qkv", lambda t: ops.gemm_nt(x3, wqkv_t, ops.EPI_BIAS_BF16, tile=t), 2.0 * M * D * 3 * D),
]
cfgs = {"auto": (0, 0), "loader": (2048, 0)}
if EXP: cfgs.update({"ld10_320": (4096, 0), "ld_burst": (2049, 0), "ld_oob": (2048, 1), "ld_noreq": (2048, 2), "ld_Ares": (2048, 4), "ld_Bres": (2048, 8), "ld_ABres": (2048, 12), "ld_Acont": (2048, 16), "ld_ABcont": (2048, 48)})
def setdbg(d):
    if EXP: _lib.load().vitamd_set_debug(d & 0xff); _lib.load().vitamd_set_debug2(d >> 8 << 8)
tot = {k: 0.0 for k in cfgs}
for name, fn, fl in calls:
    ref = None
    res = {k: [] for k in cfgs}
    for rnd in range(5):
        for k, (t, dbg) in cfgs.items():
            setdbg(dbg)
            try:
                out = fn(t)
            except _lib.VitamdError:
                setdbg(0); res[k].append(float("nan")); continue
            if rnd == 0 and dbg == 0:
                torch.cuda.synchronize()
                outs = out if isinstance(out, tuple) else (out,)
                got = [o.float().clone() for o in outs] + ([cs.clone()] if name == "dgrad_fc2" else [])
                if ref is None: ref = got
                else:
                    for a, b in zip(got, ref):
                        if not torch.equal(a, b):
                            if a.dim() == 1 and torch.allclose(a, b, rtol=1e-4, atol=1e-2): continue      # column sums: atomics order
                            print(f"  {name} {k}: MISMATCH rel {float((a - b).norm() / b.norm()):.3e} max {float((a - b).abs().max()):.3e} nan {int(torch.isnan(a).sum())}", flush=True)
            del out
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); s.record()
            for _ in range(10): fn(t)
            e.record(); torch.cuda.synchronize()
            res[k].append(s.elapsed_time(e) / 10 * 1e3)
            setdbg(0)
    for k in cfgs:
        med = statistics.median(res[k]); tot[k] += med
        print(f"{name:10s} {k:8s} {med:7.1f} us  {fl / med / 1e6:7.1f} TF  {['%.0f' % v for v in res[k]]}", flush=True)
print({k: round(v, 1) for k, v in tot.items()}, "us per layer (sum of the six launches)")
