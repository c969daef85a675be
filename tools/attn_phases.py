"""Where the attention kernels' waves spend their cycles (experimental library, dbg bit 15): per-phase shader-clock sums per wave
at the headline shape (B=256, N=197, H=12); the kernels are persistent, so a wave's sums cover the 6 heads it walks.  The ticks drain the wave's own loads, so the probed kernel is a little slower."""
import os, sys, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import ops, lib
lib.use_experimental(); L = lib.load()
L.vitamd_debug_attn_probe.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
dev = torch.device("cuda")
B, N, H = 256, 197, 12
g = torch.Generator(device="cpu").manual_seed(5)
qkv = torch.randn(B * N, 3 * H * 64, generator=g).to(dev, torch.bfloat16)
d_o = torch.randn(B * N, H * 64, generator=g).to(dev, torch.bfloat16)
o, lse = ops.attention_fwd(qkv, B, N, H)
buf = (ctypes.c_ulonglong * 16)()

def t(fn, n=5):
    fn(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3

def probe(fn, names, base=0):
    torch.cuda.synchronize(); L.vitamd_debug_attn_probe(buf)          # clear
    L.vitamd_set_debug(0x8000); fn(); torch.cuda.synchronize(); L.vitamd_set_debug(0)
    L.vitamd_debug_attn_probe(buf)
    return list(buf)

print(f"fwd  plain {t(lambda: ops.attention_fwd(qkv, B, N, H)):.1f} us")
L.vitamd_set_debug(0x8000); print(f"fwd  probed {t(lambda: ops.attention_fwd(qkv, B, N, H)):.1f} us"); L.vitamd_set_debug(0)
v = probe(lambda: ops.attention_fwd(qkv, B, N, H), None)
def show(title, v, names):
    waves = max(1, v[6]); tot = sum(c for n, c in zip(names, v[:6]) if n)
    print(f"{title}: {waves} waves, {tot / waves:.0f} cycles per wave")
    for n, c in zip(names, v[:6]):
        if n: print(f"    {n:34s} {c / waves:9.0f} cycles/wave  {100.0 * c / tot:5.1f} %")
show("forward", v, ["K/V staging + both Q loads + barrier", None, "S = K.Q^T (28 MFMA) + row max", "exp, P.V (28 MFMA)", "O store", None])
print(f"   (wave life on the 100-MHz wall clock: {v[5] / max(1, v[6]) * 10:.0f} ns -> the cycle counter runs at {sum(v[:5]) / max(1, v[5]) * 0.1:.2f} GHz)")
print(f"bwd  plain {t(lambda: ops.attention_bwd(qkv, o, lse, d_o, B, N, H)):.1f} us")
L.vitamd_set_debug(0x8000); print(f"bwd  probed {t(lambda: ops.attention_bwd(qkv, o, lse, d_o, B, N, H)):.1f} us"); L.vitamd_set_debug(0)
v = probe(lambda: ops.attention_bwd(qkv, o, lse, d_o, B, N, H), None)
show("backward dQ", v[:8], ["K/V staging + barrier", "Q, dO, O fragment loads, delta", "the pipelined tile loop (12 MFMA + exp, dS per tile)", None, None, "dQ store"])
show("backward dK/dV", v[8:], ["Q/dO staging + barrier", "K, V fragment loads", "the pipelined tile loop (16 MFMA + exp, P, dS per tile)", None, None, "dK, dV store"])
