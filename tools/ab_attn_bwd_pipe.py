"""Attention backward: the software-pipelined kernels against the plain loops (vitamd_set_debug bit 17 of the experimental library) per sequence length,
interleaved, medians.  usage: ab_attn_bwd_pipe.py [B N H ...]"""
import os, sys, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import ops, lib
lib.use_experimental(); L = lib.load()
dev = torch.device("cuda")
def t(fn, n=10):
    fn(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
a = [int(v) for v in sys.argv[1:]]
shapes = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)] or [(256, 288, 8), (256, 197, 12), (128, 256, 12)]
for (B, N, H) in shapes:
    g = torch.Generator(device="cpu").manual_seed(5)
    qkv = torch.randn(B * N, 3 * H * 64, generator=g).to(dev, torch.bfloat16)
    d_o = torch.randn(B * N, H * 64, generator=g).to(dev, torch.bfloat16)
    o, lse = ops.attention_fwd(qkv, B, N, H)
    res = {"pipelined": [], "plain": []}
    for r in range(5):
        for name, bits in (("pipelined", 0), ("plain", 0x20000)):
            L.vitamd_set_debug(bits); res[name].append(t(lambda: ops.attention_bwd(qkv, o, lse, d_o, B, N, H)))
    L.vitamd_set_debug(0)
    print(f"B {B} N {N} H {H}: pipelined {statistics.median(res['pipelined']):7.1f} us, plain loops {statistics.median(res['plain']):7.1f} us", flush=True)
