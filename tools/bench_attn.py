"""Attention forward / backward on the headline shape (B=256, N=197, H=12): fused backward against the two-kernel form (dbg bit 9)."""
import os, sys, statistics, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import ops, lib
lib.use_experimental(); L = lib.load(); L.vitamd_set_debug.argtypes = [ctypes.c_int]
dev = torch.device("cuda")
B, N, H = 256, 197, 12
g = torch.Generator(device="cpu").manual_seed(5)
qkv = torch.randn(B * N, 3 * H * 64, generator=g).to(dev, torch.bfloat16)
d_o = torch.randn(B * N, H * 64, generator=g).to(dev, torch.bfloat16)
o, lse = ops.attention_fwd(qkv, B, N, H)
def t(fn, n=10):
    fn(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
cfgs = {"two_kernel": 0}
for a in sys.argv[1:]:
    k, v = a.split("="); cfgs[k] = int(v, 0)
res = {k: [] for k in cfgs}; fw = []; fw_old = []
ref = None
for r in range(5):
    L.vitamd_set_debug(0x2000); fw.append(t(lambda: ops.attention_fwd(qkv, B, N, H))); L.vitamd_set_debug(0)
    L.vitamd_set_debug(0); fw_old.append(t(lambda: ops.attention_fwd(qkv, B, N, H))); L.vitamd_set_debug(0)
    for k, bits in cfgs.items():
        L.vitamd_set_debug(bits)
        if r == 0:
            dq = ops.attention_bwd(qkv, o, lse, d_o, B, N, H); torch.cuda.synchronize()
            if ref is None: ref = dq.float()
            else: print(k, "rel diff vs first", float((dq.float() - ref).norm() / ref.norm()))
        res[k].append(t(lambda: ops.attention_bwd(qkv, o, lse, d_o, B, N, H)))
L.vitamd_set_debug(0)
o_old, lse_old = ops.attention_fwd(qkv, B, N, H)
L.vitamd_set_debug(0x2000); o_new, lse_new = ops.attention_fwd(qkv, B, N, H); L.vitamd_set_debug(0); torch.cuda.synchronize()
print("fwd persistent vs per-head: o equal", torch.equal(o_new, o_old), "lse equal", torch.equal(lse_new, lse_old))
print(f"fwd persistent {statistics.median(fw):7.1f} us, per-head kernel {statistics.median(fw_old):7.1f} us   (HBM floor 52 us)")
L.vitamd_set_debug(0x4000); one = statistics.median(t(lambda: ops.attention_fwd(qkv, B, N, H)) for _ in range(5)); L.vitamd_set_debug(0)
print(f"fwd per-head kernel at ONE workgroup per CU: {one:7.1f} us")
for k in cfgs: print(f"bwd {k:12s} {statistics.median(res[k]):7.1f} us  {['%.0f' % v for v in res[k]]}")
