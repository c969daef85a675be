"""Attention backward on the ViT shapes: one staging of the head with split roles (attn_bwd_split_kernel, vitamd_set_debug2 bit 4 of the experimental
library) against the two pipelined kernels: bit-for-bit comparison of dqkv / dbias / delta, then interleaved timing (medians)."""
import os, sys, statistics, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import ops, lib
lib.use_experimental(); L = lib.load(); L.vitamd_set_debug2.argtypes = [ctypes.c_int]
dev = torch.device("cuda")
def t(fn, n=10):
    fn(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
for (B, N, H) in ((256, 197, 12), (64, 170, 8), (32, 65, 12), (256, 224, 12), (8, 33, 4)):
    g = torch.Generator(device="cpu").manual_seed(5)
    qkv = torch.randn(B * N, 3 * H * 64, generator=g).to(dev, torch.bfloat16)
    d_o = torch.randn(B * N, H * 64, generator=g).to(dev, torch.bfloat16)
    o, lse = ops.attention_fwd(qkv, B, N, H)
    outs = {}
    for name, bits in (("two_kernels", 0), ("split", 16)):
        L.vitamd_set_debug2(bits)
        db = torch.zeros(3 * H * 64, device=dev)
        dq = ops.attention_bwd(qkv, o, lse, d_o, B, N, H, dbias=db); torch.cuda.synchronize()
        outs[name] = (dq.clone(), db.clone())
    L.vitamd_set_debug2(0)
    same = torch.equal(outs["two_kernels"][0], outs["split"][0])
    dbd = float((outs["two_kernels"][1] - outs["split"][1]).abs().max() / outs["two_kernels"][1].abs().max())
    res = {"two_kernels": [], "split": []}
    for r in range(5):
        for name, bits in (("two_kernels", 0), ("split", 16)):
            L.vitamd_set_debug2(bits); res[name].append(t(lambda: ops.attention_bwd(qkv, o, lse, d_o, B, N, H)))
    L.vitamd_set_debug2(0)
    print(f"B {B} N {N} H {H}: dqkv bit-identical {same}, nan {int(torch.isnan(outs['split'][0].float()).sum())}, dbias rel diff {dbd:.1e} | two kernels {statistics.median(res['two_kernels']):7.1f} us, split {statistics.median(res['split']):7.1f} us  {['%.0f' % v for v in res['split']]}", flush=True)
