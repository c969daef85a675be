import os, sys, torch, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import ops, lib
lib.use_experimental(); L = lib.load(); L.vitamd_set_debug.argtypes = [ctypes.c_int]
dev = torch.device("cuda")
M, D = 256 * 197, 768
l = torch.randn(M, 4 * D, device=dev).to(torch.bfloat16); r = torch.randn(M, D, device=dev).to(torch.bfloat16)
out = torch.zeros(4 * D, D, device=dev)
for bits in (64, 5 << 26, 0):
    L.vitamd_set_debug(bits)
    for _ in range(3): ops.gemm_tn(l, r, out, accumulate=False)
    torch.cuda.synchronize()
L.vitamd_set_debug(0)
