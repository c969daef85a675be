"""NT GEMM variants against torch matmul on the device (fp32 reference of bf16 operands), several shapes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import ops
from vitamd import lib as _explib; _explib.use_experimental()
dev = torch.device("cuda")
g = torch.Generator(device="cpu").manual_seed(1)
tiles = [int(t) for t in sys.argv[1:]] or [0, 2, 7, 8, 9]
for (M, N, K) in [(50432, 2304, 768), (50432, 768, 2304), (50432, 3072, 768), (50432, 768, 3072), (1024, 768, 768), (512, 256, 128), (700, 520, 192), (256, 256, 64)]:
    a = torch.randint(-2, 3, (M, K), generator=g).to(dev, torch.bfloat16)
    b = torch.randint(-2, 3, (N, K), generator=g).to(dev, torch.bfloat16)
    ref = (a.float() @ b.float().t())
    for t in tiles:
        out = ops.gemm_nt(a, b, ops.EPI_F32 if False else ops.EPI_BIAS_BF16, tile=t)
        torch.cuda.synchronize()
        want = ref.to(torch.bfloat16).float()
        bad = (out.float() != want)
        nb = int(bad.sum())
        msg = "ok" if nb == 0 else f"BAD {nb} of {bad.numel()}; rows {bad.any(1).nonzero()[:6].flatten().tolist()} cols {bad.any(0).nonzero()[:6].flatten().tolist()} .. rows bad {int(bad.any(1).sum())} cols bad {int(bad.any(0).sum())}"
        print(f"M={M} N={N} K={K} tile={t}: {msg}", flush=True)
