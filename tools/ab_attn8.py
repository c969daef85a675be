"""Attention forward, headline shape (B=256, N=197, H=12): 8-wave online-softmax workgroups (16 waves per CU) against the 4-wave kernel with the
register-resident score row (dbg bit 15 selects the latter in experimental builds)."""
import os, sys, statistics, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import ops, lib
lib.use_experimental(); L = lib.load(); L.vitamd_set_debug.argtypes = [ctypes.c_int]
dev = torch.device("cuda")
for B, N, H in ((256, 197, 12), (256, 256, 12), (64, 170, 12)):
    g = torch.Generator(device="cpu").manual_seed(5)
    qkv = torch.randn(B * N, 3 * H * 64, generator=g).to(dev, torch.bfloat16)
    def t(n=20):
        ops.attention_fwd(qkv, B, N, H); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for _ in range(n): ops.attention_fwd(qkv, B, N, H)
        e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
    new, old = [], []
    for r in range(5):
        L.vitamd_set_debug(0); new.append(t()); L.vitamd_set_debug(0x8000); old.append(t())
    L.vitamd_set_debug(0); o1, l1 = ops.attention_fwd(qkv, B, N, H); L.vitamd_set_debug(0x8000); o0, l0 = ops.attention_fwd(qkv, B, N, H); L.vitamd_set_debug(0)
    q, k, v = (qkv.view(B, N, 3, H, 64)[:, :, i].permute(0, 2, 1, 3).float() for i in range(3))
    ref = torch.nn.functional.scaled_dot_product_attention(q, k, v).permute(0, 2, 1, 3).reshape(B * N, H * 64)
    rel = lambda a: float((a.float() - ref).norm() / ref.norm())
    x = torch.randn(B * N, H * 64, generator=g).to(dev)
    o2, l2, x1 = ops.attention_fwd(qkv, B, N, H, resid=x); torch.cuda.synchronize()
    assert torch.equal(o2, o1) and torch.equal(l2, l1) and torch.equal(x1, x + o1.float()), "fused residual form differs"
    print(f"B{B} N{N}: 8-wave {statistics.median(new):7.1f} us   4-wave {statistics.median(old):7.1f} us   rel-L2 vs fp32 SDPA: 8-wave {rel(o1):.3e} 4-wave {rel(o0):.3e}"
          f"   max|lse diff| {float((l1 - l0).abs().max()):.2e}  finite {bool(torch.isfinite(o1.float()).all())}", flush=True)
