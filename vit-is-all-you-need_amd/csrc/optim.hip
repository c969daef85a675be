// Fused AdamW step (decoupled weight decay), the optimiser of the reference's loops
// (train_vit.py:82,105 `torch.optim.AdamW` through GradScaler.step; bf16 needs no scaler).
// HBM-bound: 16 B read + 12 B written per parameter, one pass, 16-B accesses.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, size_t n4, size_t n, float lr, float b1, float b2,
                                                    float eps, float wd, float inv_bc1, float inv_sqrt_bc2) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const float decay = 1.0f - lr * wd, step = lr * inv_bc1;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 pv = *(const f32x4*)(p + 4 * i);
    const f32x4 gv = *(const f32x4*)(g + 4 * i);
    f32x4 mv = *(const f32x4*)(m + 4 * i), vv = *(const f32x4*)(v + 4 * i);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      pv[c] *= decay;
      mv[c] = b1 * mv[c] + (1.0f - b1) * gv[c];
      vv[c] = b2 * vv[c] + (1.0f - b2) * gv[c] * gv[c];
      pv[c] -= step * mv[c] / (sqrtf(vv[c]) * inv_sqrt_bc2 + eps);
    }
    *(f32x4*)(p + 4 * i) = pv;
    *(f32x4*)(m + 4 * i) = mv;
    *(f32x4*)(v + 4 * i) = vv;
  }
  if (blockIdx.x == 0) {
    for (size_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) {
      float pv = p[i] * decay;
      const float gv = g[i];
      const float mv = b1 * m[i] + (1.0f - b1) * gv, vv = b2 * v[i] + (1.0f - b2) * gv * gv;
      pv -= step * mv / (sqrtf(vv) * inv_sqrt_bc2 + eps);
      p[i] = pv; m[i] = mv; v[i] = vv;
    }
  }
}

}  // namespace

extern "C" int vitamd_adamw_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                                 float eps, float weight_decay, int step, void* stream) {
  if (n < 0 || step < 1) return VITAMD_ERR_SHAPE;
  if (n == 0) return VITAMD_OK;
  if (!p || !g || !m || !v) return VITAMD_ERR_ARG;
  if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return VITAMD_ERR_ARG;
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const size_t n4 = (size_t)n / 4;
  int grid = (int)((n4 + 255) / 256);
  grid = grid < 1 ? 1 : (grid > 2048 ? 2048 : grid);
  hipLaunchKernelGGL(adamw_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n4, (size_t)n, lr, beta1, beta2, eps,
                     weight_decay, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)));
  return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
}
