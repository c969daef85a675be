"""Where does the cost of a resident foreign kernel (tools/cu_thief.py) come from?  Three main-stream sequences timed with and without
ONE spinning workgroup resident on another stream: (a) 50 back-to-back large GEMMs (few kernel boundaries per unit time),
(b) 2000 tiny dependent kernels (nothing but boundaries), (c) 200 LayerNorm-sized HBM-bound kernels."""
import os, sys, time, ctypes, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import ops
_so = os.path.join(ROOT, "tools", "probes", "libcuthief.so")
if not os.path.exists(_so):        # hipcc cross-compiles without a GPU; the .so then travels with the snapshot
    import subprocess
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(ROOT, "tools", "probes", "cu_thief.hip"), "-o", _so])
lib = ctypes.CDLL(_so)
lib.thief_launch.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda")
M, D = 256 * 197, 768
g = torch.Generator(device="cpu").manual_seed(1)
x4 = torch.randn(M, 4 * D, generator=g).to(dev, torch.bfloat16)
w1_t = (torch.randn(D, 4 * D, generator=g) * 0.03).to(dev, torch.bfloat16)
xf = torch.randn(M, D, device=dev)
small = torch.zeros(256, device=dev)
third = torch.cuda.Stream(); sink = torch.zeros(4, dtype=torch.int32, device=dev)
B, N, H = 256, 197, 12
qkv = torch.randn(B * N, 3 * H * 64, generator=g).to(dev, torch.bfloat16); d_o = torch.randn(B * N, H * 64, generator=g).to(dev, torch.bfloat16)
o, lse = ops.attention_fwd(qkv, B, N, H)
x1 = torch.randn(M, D, generator=g).to(dev, torch.bfloat16); dW1 = torch.empty(4 * D, D, device=dev)
w1 = (torch.randn(4 * D, D, generator=g) * 0.03).to(dev, torch.bfloat16); b4 = torch.randn(4 * D, device=dev)
xs = {r: torch.randn(r, 4 * D, generator=g).to(dev, torch.bfloat16) for r in (320 * 8, 320 * 32, 320 * 80, 320 * 85, 320 * 86)}
seqs = {
    **{f"100 NT GEMMs of {r // 320 * 3} tiles": ((lambda r=r: ops.gemm_nt(xs[r], w1_t, ops.EPI_BIAS_BF16)), 100) for r in xs},
    "50 fc1+GELU GEMMs (1896 tiles)": (lambda: ops.gemm_nt(x1, w1, ops.EPI_GELU_DG, bias=b4), 50),
    "50 weight-gradient GEMMs (144 workgroups)": (lambda: ops.gemm_tn(x4, x1, dW1, accumulate=False, splits=4), 50),
    "50 attention forwards (3072 workgroups)": (lambda: ops.attention_fwd(qkv, B, N, H), 50),
    "30 attention backwards": (lambda: ops.attention_bwd(qkv, o, lse, d_o, B, N, H), 30),
    "50 large NT GEMMs (180 us each)": (lambda: ops.gemm_nt(x4, w1_t, ops.EPI_BIAS_BF16), 50),
    "2000 tiny dependent kernels": (lambda: small.add_(1.0), 2000),
    "200 LayerNorm forwards (36 us each)": (lambda: ops.layernorm_fwd(xf), 200),
}
def timed(fn, n, thief):
    fn(); torch.cuda.synchronize()
    if thief:
        lib.thief_launch(1, 400.0, sink.data_ptr(), third.cuda_stream); time.sleep(0.002)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); e.synchronize()
    ms = s.elapsed_time(e)
    torch.cuda.synchronize()
    return ms * 1e3 / n
for name, (fn, n) in seqs.items():
    a = statistics.median(timed(fn, n, False) for _ in range(3))
    b = statistics.median(timed(fn, n, True) for _ in range(3))
    print(f"{name:40s} alone {a:8.2f} us per kernel   beside one resident workgroup {b:8.2f} us   (+{b - a:.2f} us)", flush=True)
