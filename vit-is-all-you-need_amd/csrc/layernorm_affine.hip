// Affine LayerNorm (nn.LayerNorm with weight/bias, eps as given) forward / backward for the
// `blocks.py` surface of the reference (blocks.py:43,48,179,184: ln_1/ln_2, norm1/norm2).  Same
// conventions as layernorm.hip: fp32 residual-stream input, bf16 normalised output for the next GEMM,
// one wave per row.  Not on the measured ViT path (which uses the non-affine kernels), so this is the
// straightforward any-width form (row re-read through L1/L2 instead of register-resident).
//   forward : y = bf16(((x - mean) * rstd) * gamma + beta), mean/rstd saved
//   backward: dxhat = dy * gamma;  g_out = (g_res or 0) + rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat))
//             dgamma += sum_rows dy * xhat ; dbeta += sum_rows dy ; optional bf16 copy of g_out + column sums
#include "common.h"

namespace {

constexpr int WAVES = 4;

__device__ __forceinline__ void put(__bf16* p, float v) { *p = f2bf(v); }
__device__ __forceinline__ void put(float* p, float v) { *p = v; }
__device__ __forceinline__ float get(const __bf16* p) { return bf2f(*p); }
__device__ __forceinline__ float get(const float* p) { return *p; }

// YT = __bf16 (feeds a GEMM) or float (ln_pre / ln_post of the tokenizers, whose output stays in the fp32 stream)
template <typename YT>
__global__ __launch_bounds__(256) void ln_affine_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, YT* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd, int M, int D, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int row = blockIdx.x * WAVES + wave; row < M; row += gridDim.x * WAVES) {
    const float* xr = x + (size_t)row * D;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += xr[c];
    const float mu = wave_sum(s) / D;
    float q = 0.f;
    for (int c = lane; c < D; c += 64) { const float d = xr[c] - mu; q += d * d; }
    const float rs = rsqrtf(wave_sum(q) / D + eps);
    for (int c = lane; c < D; c += 64) put(y + (size_t)row * D + c, (xr[c] - mu) * rs * gamma[c] + beta[c]);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
  }
}

template <typename DT>
__global__ __launch_bounds__(256) void ln_affine_bwd_kernel(const DT* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ g_res,
                                                            float* __restrict__ g_out, __bf16* __restrict__ g_bf16,
                                                            float* __restrict__ colsum, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int M, int D) {
  extern __shared__ float part[];   // [3][D] per-block partials: dgamma, dbeta, colsum
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c = threadIdx.x; c < 3 * D; c += 256) part[c] = 0.f;
  __syncthreads();
  for (int row = blockIdx.x * WAVES + wave; row < M; row += gridDim.x * WAVES) {
    const size_t base = (size_t)row * D;
    const float mu = mean[row], rs = rstd[row];
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < D; c += 64) {
      const float d = get(dy + base + c) * gamma[c];
      const float xh = (x[base + c] - mu) * rs;
      s1 += d;
      s2 += d * xh;
    }
    const float m1 = wave_sum(s1) / D, m2 = wave_sum(s2) / D;
    for (int c = lane; c < D; c += 64) {
      const float dyr = get(dy + base + c);
      const float xh = (x[base + c] - mu) * rs;
      float g = rs * (dyr * gamma[c] - m1 - xh * m2);
      if (g_res) g += g_res[base + c];
      g_out[base + c] = g;
      atomicAdd(&part[c], dyr * xh);          // LDS atomics: 4 waves share the block partials
      atomicAdd(&part[D + c], dyr);
      if (g_bf16) {
        const __bf16 gb = f2bf(g);
        g_bf16[base + c] = gb;
        if (colsum) atomicAdd(&part[2 * D + c], bf2f(gb));
      }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    atomicAdd(dgamma + c, part[c]);
    atomicAdd(dbeta + c, part[D + c]);
    if (colsum && g_bf16) atomicAdd(colsum + c, part[2 * D + c]);
  }
}

}  // namespace

extern "C" int vitamd_layernorm_affine_fwd(const float* x, const float* gamma, const float* beta, void* y_bf16, float* mean,
                                           float* rstd, int M, int D, float eps, void* stream) {
  if (M <= 0 || D <= 0) return VITAMD_ERR_SHAPE;
  if (!x || !gamma || !beta || !y_bf16 || !mean || !rstd) return VITAMD_ERR_ARG;
  int grid = (M + WAVES - 1) / WAVES; grid = grid > 2048 ? 2048 : grid;
  hipLaunchKernelGGL(ln_affine_fwd_kernel<__bf16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, (__bf16*)y_bf16, mean, rstd, M, D, eps);
  return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
}

extern "C" int vitamd_layernorm_affine_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                               float* rstd, int M, int D, float eps, void* stream) {
  if (M <= 0 || D <= 0) return VITAMD_ERR_SHAPE;
  if (!x || !gamma || !beta || !y || !mean || !rstd) return VITAMD_ERR_ARG;
  int grid = (M + WAVES - 1) / WAVES; grid = grid > 2048 ? 2048 : grid;
  hipLaunchKernelGGL(ln_affine_fwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, mean, rstd, M, D, eps);
  return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
}

extern "C" int vitamd_layernorm_affine_bwd(const void* dy_bf16, const float* x, const float* mean, const float* rstd,
                                           const float* gamma, const float* g_res, float* g_out, void* g_bf16, float* colsum,
                                           float* dgamma, float* dbeta, int M, int D, void* stream) {
  if (M <= 0 || D <= 0 || D > 4096) return VITAMD_ERR_SHAPE;
  if (!dy_bf16 || !x || !mean || !rstd || !gamma || !g_out || !dgamma || !dbeta) return VITAMD_ERR_ARG;
  int grid = (M + WAVES - 1) / WAVES; grid = grid > 1024 ? 1024 : grid;
  hipLaunchKernelGGL(ln_affine_bwd_kernel<__bf16>, dim3(grid), dim3(256), 3 * D * sizeof(float), (hipStream_t)stream, (const __bf16*)dy_bf16, x,
                     mean, rstd, gamma, g_res, g_out, (__bf16*)g_bf16, colsum, dgamma, dbeta, M, D);
  return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
}

extern "C" int vitamd_layernorm_affine_bwd_f32(const float* dy, const float* x, const float* mean, const float* rstd,
                                               const float* gamma, float* g_out, float* dgamma, float* dbeta, int M, int D, void* stream) {
  if (M <= 0 || D <= 0 || D > 4096) return VITAMD_ERR_SHAPE;
  if (!dy || !x || !mean || !rstd || !gamma || !g_out || !dgamma || !dbeta) return VITAMD_ERR_ARG;
  int grid = (M + WAVES - 1) / WAVES; grid = grid > 1024 ? 1024 : grid;
  hipLaunchKernelGGL(ln_affine_bwd_kernel<float>, dim3(grid), dim3(256), 3 * D * sizeof(float), (hipStream_t)stream, dy, x, mean, rstd, gamma,
                     (const float*)nullptr, g_out, (__bf16*)nullptr, (float*)nullptr, dgamma, dbeta, M, D);
  return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
}
