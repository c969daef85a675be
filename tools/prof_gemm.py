"""Launch each hot GEMM shape a few times (for rocprofv3 --pmc / --kernel-trace runs)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import ops
from vitamd import lib as _explib; _explib.use_experimental()
dev = torch.device("cuda")
B, N, D = 256, 197, 768
M = B * N
g = torch.Generator(device="cpu").manual_seed(0)
rb = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dev, torch.bfloat16)
x768, x3072, x2304 = rb(M, D), rb(M, 4 * D), rb(M, 3 * D)
wqkv, w1, w2 = rb(3 * D, D, scale=0.03), rb(4 * D, D, scale=0.03), rb(D, 4 * D, scale=0.03)
tiles = [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "1,256").split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for tile in tiles:
    for _ in range(reps):
        ops.gemm_nt(x768, wqkv, ops.EPI_BIAS_BF16, tile=tile)      # K=768  N=2304
        ops.gemm_nt(x3072, w2, ops.EPI_BIAS_BF16, tile=tile)       # K=3072 N=768
dW = torch.zeros(3 * D, D, device=dev)
for _ in range(reps):
    ops.gemm_tn(x2304, x768, dW)
torch.cuda.synchronize()
