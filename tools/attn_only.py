"""Attention forward + backward alone at the headline shape (for rocprofv3 --pmc passes over just these kernels)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import ops
dev = torch.device("cuda")
B, N, H = 256, 197, 12
g = torch.Generator(device="cpu").manual_seed(5)
qkv = torch.randn(B * N, 3 * H * 64, generator=g).to(dev, torch.bfloat16)
d_o = torch.randn(B * N, H * 64, generator=g).to(dev, torch.bfloat16)
for _ in range(3):
    o, lse = ops.attention_fwd(qkv, B, N, H)
    ops.attention_bwd(qkv, o, lse, d_o, B, N, H)
torch.cuda.synchronize()
