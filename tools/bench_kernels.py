"""Per-kernel timing at the ViT-B/16 batch-256 shapes (M = 50432).  Run on the GPU box:
   python tools/bench_kernels.py [--quick]
Prints one line per kernel: avg ms, TFLOP/s or GB/s."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import ops  # noqa: E402
from vitamd import lib as _explib; _explib.use_experimental()

BF16 = torch.bfloat16
dev = torch.device("cuda")


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    B, N, H, D = 256, 197, 12, 768
    M = B * N
    g = torch.Generator(device="cpu").manual_seed(0)

    def rb(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g) * scale).to(dev, BF16)

    x768 = rb(M, D)
    x3072 = rb(M, 4 * D)
    x2304 = rb(M, 3 * D)
    wqkv, w1, w2 = rb(3 * D, D, scale=0.03), rb(4 * D, D, scale=0.03), rb(D, 4 * D, scale=0.03)
    bias3, bias4, bias1 = torch.randn(3 * D, device=dev), torch.randn(4 * D, device=dev), torch.randn(D, device=dev)
    res = torch.randn(M, D, device=dev)
    rows = []

    def rec(name, ms, flops=None, bytes_=None):
        s = f"{name:34s} {ms:8.3f} ms"
        if flops:
            s += f"  {flops / ms / 1e9:8.1f} TFLOP/s"
        if bytes_:
            s += f"  {bytes_ / ms / 1e6:8.1f} GB/s"
        print(s, flush=True)
        rows.append(s)

    wqkv_t = rb(D, 3 * D, scale=0.03)
    for tile in (2, 256):
        rec(f"gemm_nt qkv  bias   t{tile}", timeit(lambda: ops.gemm_nt(x768, wqkv, ops.EPI_BIAS_BF16, bias=bias3, tile=tile)), 2 * M * D * 3 * D)
        rec(f"gemm_nt fc1  gelu   t{tile}", timeit(lambda: ops.gemm_nt(x768, w1, ops.EPI_GELU, bias=bias4, tile=tile)), 2 * M * D * 4 * D)
        rec(f"gemm_nt fc2  resid  t{tile}", timeit(lambda: ops.gemm_nt(x3072, w2, ops.EPI_RESID_F32, bias=bias1, aux=res, tile=tile)), 2 * M * D * 4 * D)
        rec(f"gemm_nt dfc2 dgelu  t{tile}", timeit(lambda: ops.gemm_nt(x768, w1, ops.EPI_DGELU, aux=x3072, tile=tile)), 2 * M * D * 4 * D)
        rec(f"gemm_nt dfc1 plain  t{tile}", timeit(lambda: ops.gemm_nt(x3072, w2, ops.EPI_BIAS_BF16, tile=tile)), 2 * M * D * 4 * D)
        rec(f"gemm_nt dqkv plain  t{tile}", timeit(lambda: ops.gemm_nt(x2304, wqkv_t, ops.EPI_BIAS_BF16, tile=tile)), 2 * M * D * 3 * D)
    dW = torch.zeros(3 * D, D, device=dev)
    rec("gemm_tn wgrad qkv (atomic)", timeit(lambda: ops.gemm_tn(x2304, x768, dW, atomic=True)), 2 * M * D * 3 * D)
    dWa = torch.zeros(4 * D, D, device=dev)
    rec("gemm_tn wgrad fc1 (atomic)", timeit(lambda: ops.gemm_tn(x3072, x768, dWa, atomic=True)), 2 * M * D * 4 * D)
    rec("gemm_tn wgrad qkv", timeit(lambda: ops.gemm_tn(x2304, x768, dW)), 2 * M * D * 3 * D)
    dW1 = torch.zeros(4 * D, D, device=dev)
    rec("gemm_tn wgrad fc1", timeit(lambda: ops.gemm_tn(x3072, x768, dW1)), 2 * M * D * 4 * D)
    dW2 = torch.zeros(D, 4 * D, device=dev)
    rec("gemm_tn wgrad fc2", timeit(lambda: ops.gemm_tn(x768, x3072, dW2)), 2 * M * D * 4 * D)
    import ctypes
    from vitamd import lib as _lib
    _L = _lib.load(); _L.vitamd_set_debug.argtypes = [ctypes.c_int]
    _L.vitamd_set_debug(64)
    rec("gemm_tn wgrad qkv (16x16x32 variant)", timeit(lambda: ops.gemm_tn(x2304, x768, dW)), 2 * M * D * 3 * D)
    rec("gemm_tn wgrad fc1 (16x16x32 variant)", timeit(lambda: ops.gemm_tn(x3072, x768, dW1)), 2 * M * D * 4 * D)
    _L.vitamd_set_debug(0)
    for sp in (7,):
        rec(f"gemm_tn wgrad fc1 splits={sp}", timeit(lambda: ops.gemm_tn(x3072, x768, dW1, splits=sp)), 2 * M * D * 4 * D)

    xf = torch.randn(M, D, device=dev)
    rec("layernorm_fwd", timeit(lambda: ops.layernorm_fwd(xf)), bytes_=M * D * 6)
    rec("layernorm_fwd + add", timeit(lambda: ops.layernorm_fwd(xf, addend=x768)), bytes_=M * D * 12)
    _, y, mean, rstd = ops.layernorm_fwd(xf)
    cs = torch.zeros(D, device=dev)
    rec("layernorm_bwd (+res,+bf16,+colsum)", timeit(lambda: ops.layernorm_bwd(x768, xf, mean, rstd, g_res=res, want_bf16=True, colsum=cs)), bytes_=M * D * 16)
    qkv = rb(M, 3 * D)
    o, lse = ops.attention_fwd(qkv, B, N, H)
    aflops = 4 * B * H * N * N * 64
    rec("attention_fwd", timeit(lambda: ops.attention_fwd(qkv, B, N, H)), aflops)
    rec("attention_bwd", timeit(lambda: ops.attention_bwd(qkv, o, lse, x768, B, N, H)), aflops * 2.5)
    rec("cast_bf16 [M,768]", timeit(lambda: ops.cast_bf16(xf)), bytes_=M * D * 6)
    rec("colsum [M,2304]", timeit(lambda: ops.colsum(qkv)), bytes_=M * 3 * D * 2)
    img = torch.randn(B, 3, 224, 224, device=dev)
    rec("im2col", timeit(lambda: ops.im2col(img, 16)), bytes_=B * 3 * 224 * 224 * 6)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_kernels.txt"), "w") as f:
        f.write("\n".join(rows) + "\n")


if __name__ == "__main__":
    main()
