"""Weight-gradient (TN) loader-wave kernel with its requests split around the phase's first barrier + loader priority (shipped since round 4) against the
round-3 loaders ("shipped" in the log; vitamd_set_debug2 bit 6 of the experimental library), on the three ViT-B shapes with the step's split factors: interleaved, medians, bit-for-bit compare."""
import os, sys, statistics, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import ops, lib, functions as F
lib.use_experimental(); L = lib.load()
dev = torch.device("cuda")
R = 256 * 197
g = torch.Generator(device="cpu").manual_seed(3)
for name, P, Q in (("dWqkv", 2304, 768), ("dW1", 3072, 768), ("dW2", 768, 3072)):
    l = torch.randn(R, P, generator=g).to(dev, torch.bfloat16)
    r = torch.randn(R, Q, generator=g).to(dev, torch.bfloat16)
    out = torch.empty(P, Q, device=dev)
    for splits in (0, F._tn_splits(out)):
        res = {"split": [], "shipped": []}; ref = None
        for rnd in range(5):
            for k, bits in (("split", 0), ("shipped", 64)):
                L.vitamd_set_debug2(bits)
                ops.gemm_tn(l, r, out, accumulate=False, splits=splits, form=ops.TN_FORM_EXCLUSIVE)
                if rnd == 0:
                    torch.cuda.synchronize()
                    if ref is None: ref = out.clone()
                    elif not torch.equal(out, ref): print("  MISMATCH", name, float((out - ref).abs().max()))
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); s.record()
                for _ in range(10): ops.gemm_tn(l, r, out, accumulate=False, splits=splits, form=ops.TN_FORM_EXCLUSIVE)
                e.record(); torch.cuda.synchronize()
                res[k].append(s.elapsed_time(e) / 10 * 1e3)
        L.vitamd_set_debug2(0)
        fl = 2.0 * R * P * Q
        print(f"{name:6s} splits {splits}: " + "  ".join(f"{k} {statistics.median(v):7.1f} us ({fl / statistics.median(v) / 1e6:6.0f} TF)" for k, v in res.items()), flush=True)
