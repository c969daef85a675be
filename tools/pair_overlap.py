"""What does a weight-gradient (TN) GEMM cost when it runs BESIDE another kernel of the backward pass?  For each partner kernel:
time of k partner launches alone, of k TN launches alone, and of both sequences enqueued on two streams at once.
overlap gain = alone_P + alone_TN - together."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
if os.environ.get("TN_SMALL_RING"):        # co-residency experiment: the 5-slot-ring (80 KiB) weight-gradient kernel of the experimental library
    from vitamd import lib as _explib; _explib.use_experimental()
from vitamd import ops, functions as F
if os.environ.get("TN_SMALL_RING"):
    _explib.load().vitamd_set_debug(-2147483648)     # bit 31
dev = torch.device("cuda")
B, N, H, D = 256, 197, 12, 768
M = B * N
g = torch.Generator(device="cpu").manual_seed(3)
rb = lambda *s: torch.randn(*s, generator=g).to(dev, torch.bfloat16)
x1, x3, x4 = rb(M, D), rb(M, 3 * D), rb(M, 4 * D)
w1_t = (torch.randn(D, 4 * D, generator=g) * 0.03).to(dev, torch.bfloat16)
dW1 = torch.empty(4 * D, D, device=dev)
qkv = rb(M, 3 * D); d_o = rb(M, D)
o, lse = ops.attention_fwd(qkv, B, N, H)
xf = torch.randn(M, D, device=dev); mean = torch.zeros(M, device=dev); rstd = torch.ones(M, device=dev); gres = torch.randn(M, D, device=dev)
side = torch.cuda.Stream()
partners = {
    "LN backward": lambda: ops.layernorm_bwd(x1, xf, mean, rstd, g_res=gres, want_bf16=True, xhat=x1),
    "attention backward": lambda: ops.attention_bwd(qkv, o, lse, d_o, B, N, H),
    "NT dgrad-fc1": lambda: ops.gemm_nt(x4, w1_t, ops.EPI_BIAS_BF16),
    "attention forward": lambda: ops.attention_fwd(qkv, B, N, H),
}
def tn(splits):
    return lambda: ops.gemm_tn(x4, x1, dW1, accumulate=False, splits=splits)
def run(main_fn, side_fn, k=6, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        if side_fn is not None:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(k): side_fn()
        if main_fn is not None:
            for _ in range(k): main_fn()
        if side_fn is not None: torch.cuda.current_stream().wait_stream(side)
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / k * 1e3)
    return best
ref = None
for wgs in (128, 252):
    F.TN_TARGET_WGS = wgs
    t = tn(F._tn_splits(dW1))
    t(); torch.cuda.synchronize()
    if ref is None: ref = dW1.clone(); print("dW1 checksum %.6e" % float(dW1.double().abs().sum()))
    ta = run(None, t)
    print(f"TN dW1 cut into ~{wgs} workgroups alone: {ta:.0f} us per launch")
    for name, p in partners.items():
        p(); torch.cuda.synchronize()
        pa = run(p, None)
        both = run(p, t)
        print(f"   beside {name:20s}: partner alone {pa:6.0f}  together {both:6.0f}  sum {pa + ta:6.0f}  -> overlap gain {pa + ta - both:6.0f} us per pair")
