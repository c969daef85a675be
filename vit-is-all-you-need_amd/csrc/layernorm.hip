// Non-affine LayerNorm (eps 1e-5, biased variance) forward / backward on the fp32 residual stream,
// reference transformer.py:43-44 `F.layer_norm(x, (n_embd,))`, fused with the residual adds either
// side of it (transformer.py:43-44 `x = x + f(LN(x))`).  HBM-bound: one wave per row, 16-B loads,
// the row lives in registers between the statistics and the normalisation (one read of x).
//   forward : x = x_in (+ addend_bf16)   -> x_out fp32 (optional), y = bf16(LN(x)), mean, rstd
//   backward: g = g_res + LNbwd(dy_bf16; x, mean, rstd) -> g_out fp32, optional bf16(g) copy and
//             per-column sums of that bf16 copy (the bias gradient of the Linear whose output
//             gradient it is).
#include "common.h"

namespace {

constexpr int ROWS_PER_BLOCK = 4;  // 4 waves, one row each per iteration
constexpr int MAXV = 4;            // up to 4 float4 per lane: D <= 1024, D % 256 == 0

template <int NV, bool HAS_ADD>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x_in, const __bf16* __restrict__ addend,
                                                     float* __restrict__ x_out, __bf16* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int M, float eps) {
  constexpr int D = NV * 256;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int row = blockIdx.x * ROWS_PER_BLOCK + wave; row < M; row += gridDim.x * ROWS_PER_BLOCK) {
    f32x4 v[NV];
    const size_t base = (size_t)row * D;
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] = *(const f32x4*)(x_in + base + j * 256 + lane * 4);
    if constexpr (HAS_ADD) {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const u32x2 a = *(const u32x2*)(addend + base + j * 256 + lane * 4);
        v[j][0] += bf16lo(a[0]); v[j][1] += bf16hi(a[0]); v[j][2] += bf16lo(a[1]); v[j][3] += bf16hi(a[1]);
        *(f32x4*)(x_out + base + j * 256 + lane * 4) = v[j];
      }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    const float mu = wave_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
      for (int c = 0; c < 4; ++c) { const float d = v[j][c] - mu; q += d * d; }
    const float rs = rsqrtf(wave_sum(q) * (1.0f / D) + eps);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      u32x2 o = {pack_bf16x2((v[j][0] - mu) * rs, (v[j][1] - mu) * rs), pack_bf16x2((v[j][2] - mu) * rs, (v[j][3] - mu) * rs)};
      *(u32x2*)(y + base + j * 256 + lane * 4) = o;
    }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
  }
}

// generic-width fallback (D % 4 == 0): three passes over the row through L1/L2
template <bool HAS_ADD>
__global__ __launch_bounds__(256) void ln_fwd_generic(const float* __restrict__ x_in, const __bf16* __restrict__ addend,
                                                      float* __restrict__ x_out, __bf16* __restrict__ y,
                                                      float* __restrict__ mean, float* __restrict__ rstd, int M, int D, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int row = blockIdx.x * ROWS_PER_BLOCK + wave; row < M; row += gridDim.x * ROWS_PER_BLOCK) {
    const size_t base = (size_t)row * D;
    const float* xr = x_in + base;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) {
      float t = xr[c];
      if constexpr (HAS_ADD) { t += bf2f(addend[base + c]); x_out[base + c] = t; }
      s += t;
    }
    if constexpr (HAS_ADD) xr = x_out + base;
    const float mu = wave_sum(s) / D;
    float q = 0.f;
    for (int c = lane; c < D; c += 64) { const float d = xr[c] - mu; q += d * d; }
    const float rs = rsqrtf(wave_sum(q) / D + eps);
    for (int c = lane; c < D; c += 64) y[base + c] = f2bf((xr[c] - mu) * rs);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
  }
}

// XH: xhat is read back from the forward's bf16 output y (= LN(x) exactly for this non-affine LayerNorm, saved anyway as the
// weight-gradient operand of the following Linear) through the `x` pointer, instead of being recomputed from the fp32 input:
// 2 B instead of 4 B per element of an HBM-bound kernel (16 -> 14 B/elem), `mean` unused.  The bf16 rounding of xhat only
// touches the xhat * mean(dy * xhat) term (|.| ~ 0.1 |dy|): ~2e-4 relative on g, far below the bf16 roundings around it.
template <int NV, bool XH = false>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const __bf16* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ g_res, float* __restrict__ g_out,
                                                     __bf16* __restrict__ g_bf16, float* __restrict__ colsum, int M,
                                                     unsigned dthresh, float dscale, unsigned dseed_lo, unsigned dseed_hi) {
  constexpr int D = NV * 256;
  __shared__ float red[ROWS_PER_BLOCK][D];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 cs[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) cs[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int row = blockIdx.x * ROWS_PER_BLOCK + wave; row < M; row += gridDim.x * ROWS_PER_BLOCK) {
    const size_t base = (size_t)row * D;
    const float mu = XH ? 0.f : mean[row], rs = rstd[row];
    f32x4 d[NV], xh[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const u32x2 a = *(const u32x2*)(dy + base + j * 256 + lane * 4);
      d[j] = (f32x4){bf16lo(a[0]), bf16hi(a[0]), bf16lo(a[1]), bf16hi(a[1])};
      if constexpr (XH) {
        const u32x2 yb = *(const u32x2*)((const __bf16*)x + base + j * 256 + lane * 4);
        xh[j] = (f32x4){bf16lo(yb[0]), bf16hi(yb[0]), bf16lo(yb[1]), bf16hi(yb[1])};
      } else {
        const f32x4 xv = *(const f32x4*)(x + base + j * 256 + lane * 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) xh[j][c] = (xv[c] - mu) * rs;
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        s1 += d[j][c];
        s2 += d[j][c] * xh[j][c];
      }
    }
    const float m1 = wave_sum(s1) * (1.0f / D), m2 = wave_sum(s2) * (1.0f / D);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      f32x4 g;
#pragma unroll
      for (int c = 0; c < 4; ++c) g[c] = rs * (d[j][c] - m1 - xh[j][c] * m2);
      if (g_res) g += *(const f32x4*)(g_res + base + j * 256 + lane * 4);
      *(f32x4*)(g_out + base + j * 256 + lane * 4) = g;
      if (g_bf16) {
        u32x2 o = {pack_bf16x2(g[0], g[1]), pack_bf16x2(g[2], g[3])};
        if (dthresh) {   // the bf16 copy is the gradient of a dropped-out Linear output: same mask as the forward
          const unsigned long long e0 = (unsigned long long)base + j * 256 + lane * 4;
          o[0] = pack_bf16x2(bf16lo(o[0]) * dropout_keep(e0, dseed_lo, dseed_hi, dthresh, dscale),
                             bf16hi(o[0]) * dropout_keep(e0 + 1, dseed_lo, dseed_hi, dthresh, dscale));
          o[1] = pack_bf16x2(bf16lo(o[1]) * dropout_keep(e0 + 2, dseed_lo, dseed_hi, dthresh, dscale),
                             bf16hi(o[1]) * dropout_keep(e0 + 3, dseed_lo, dseed_hi, dthresh, dscale));
        }
        *(u32x2*)(g_bf16 + base + j * 256 + lane * 4) = o;
        if (colsum) {
          cs[j][0] += bf16lo(o[0]); cs[j][1] += bf16hi(o[0]); cs[j][2] += bf16lo(o[1]); cs[j][3] += bf16hi(o[1]);
        }
      }
    }
  }
  if (colsum && g_bf16) {
#pragma unroll
    for (int j = 0; j < NV; ++j) *(f32x4*)(&red[wave][j * 256 + lane * 4]) = cs[j];
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < ROWS_PER_BLOCK; ++w) s += red[w][c];
      atomicAdd(colsum + c, s);  // 256 contiguous bytes per wave-instruction
    }
  }
}

__global__ __launch_bounds__(256) void ln_bwd_generic(const __bf16* __restrict__ dy, const float* __restrict__ x,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      const float* __restrict__ g_res, float* __restrict__ g_out,
                                                      __bf16* __restrict__ g_bf16, float* __restrict__ colsum, int M, int D,
                                                      unsigned dthresh, float dscale, unsigned dseed_lo, unsigned dseed_hi) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int row = blockIdx.x * ROWS_PER_BLOCK + wave; row < M; row += gridDim.x * ROWS_PER_BLOCK) {
    const size_t base = (size_t)row * D;
    const float mu = mean[row], rs = rstd[row];
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < D; c += 64) {
      const float d = bf2f(dy[base + c]);
      s1 += d;
      s2 += d * (x[base + c] - mu) * rs;
    }
    const float m1 = wave_sum(s1) / D, m2 = wave_sum(s2) / D;
    for (int c = lane; c < D; c += 64) {
      const float d = bf2f(dy[base + c]);
      float g = rs * (d - m1 - (x[base + c] - mu) * rs * m2);
      if (g_res) g += g_res[base + c];
      g_out[base + c] = g;
      if (g_bf16) {
        __bf16 gb = f2bf(g);
        if (dthresh) gb = f2bf(bf2f(gb) * dropout_keep((unsigned long long)base + c, dseed_lo, dseed_hi, dthresh, dscale));
        g_bf16[base + c] = gb;
        if (colsum) atomicAdd(colsum + c, bf2f(gb));
      }
    }
  }
}

// One row per wave where nothing is shared between rows (12 608 workgroups at M = 50 432: 4-7 % faster than a 2 048-workgroup grid-stride
// launch, tools/bench_ln.py); the column-sum form keeps a grid-stride loop - every workgroup ends with D atomics (12 608 of them: 312 us).
int grid_for(int M, bool colsum = false) {
  int blocks = (M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
  int cap = !colsum ? 16384 : 1024;                 // column-sum form: 2048 -> 118 us, 1024 -> 111, 512 -> 110, 4096 -> 135
#ifdef VITAMD_EXPERIMENTAL
  if (colsum && (g_vitamd_debug >> 24)) cap = 256 * (g_vitamd_debug >> 24);     // dbg bits 24-31: the cap in units of 256 blocks (sweep)
#endif
  return blocks < cap ? blocks : cap;
}

}  // namespace

extern "C" int vitamd_layernorm_fwd(const float* x_in, const void* addend_bf16, float* x_out, void* y_bf16, float* mean,
                                    float* rstd, int M, int D, float eps, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (M <= 0 || D <= 0 || D % 4) return VITAMD_ERR_SHAPE;
  if (!x_in || !y_bf16 || !mean || !rstd || (addend_bf16 && !x_out)) return VITAMD_ERR_ARG;
  const __bf16* add = (const __bf16*)addend_bf16;
  __bf16* y = (__bf16*)y_bf16;
  const int grid = grid_for(M);
#define LN_FWD(NV)                                                                                                   \
  if (add) hipLaunchKernelGGL((ln_fwd_kernel<NV, true>), dim3(grid), dim3(256), 0, stream, x_in, add, x_out, y, mean, rstd, M, eps); \
  else hipLaunchKernelGGL((ln_fwd_kernel<NV, false>), dim3(grid), dim3(256), 0, stream, x_in, add, x_out, y, mean, rstd, M, eps)
  if (D == 256) { LN_FWD(1); }
  else if (D == 512) { LN_FWD(2); }
  else if (D == 768) { LN_FWD(3); }
  else if (D == 1024) { LN_FWD(4); }
  else if (add) hipLaunchKernelGGL((ln_fwd_generic<true>), dim3(grid), dim3(256), 0, stream, x_in, add, x_out, y, mean, rstd, M, D, eps);
  else hipLaunchKernelGGL((ln_fwd_generic<false>), dim3(grid), dim3(256), 0, stream, x_in, add, x_out, y, mean, rstd, M, D, eps);
#undef LN_FWD
  return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
}

static int ln_bwd_launch(const void* dy_bf16, const float* x, const float* mean, const float* rstd, const float* g_res, float* g_out,
                         void* g_bf16, float* colsum, int M, int D, float dropout_p, unsigned long long seed, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (M <= 0 || D <= 0 || D % 4) return VITAMD_ERR_SHAPE;
  if (!dy_bf16 || !x || !mean || !rstd || !g_out) return VITAMD_ERR_ARG;
  if (!(dropout_p >= 0.f) || dropout_p >= 1.f) return VITAMD_ERR_ARG;
  unsigned dthresh = dropout_p > 0.f ? (unsigned)((double)dropout_p * 4294967296.0) : 0u;
  if (dropout_p > 0.f && dthresh == 0u) dthresh = 1u;
  const float dscale = 1.0f / (1.0f - dropout_p);
  const unsigned slo = (unsigned)seed, shi = (unsigned)(seed >> 32);
  const __bf16* dy = (const __bf16*)dy_bf16;
  __bf16* gb = (__bf16*)g_bf16;
  const int grid = grid_for(M, colsum != nullptr && g_bf16 != nullptr);
#define LN_BWD(NV) hipLaunchKernelGGL((ln_bwd_kernel<NV>), dim3(grid), dim3(256), 0, stream, dy, x, mean, rstd, g_res, g_out, gb, colsum, M, dthresh, dscale, slo, shi)
  if (D == 256) { LN_BWD(1); }
  else if (D == 512) { LN_BWD(2); }
  else if (D == 768) { LN_BWD(3); }
  else if (D == 1024) { LN_BWD(4); }
  else hipLaunchKernelGGL(ln_bwd_generic, dim3(grid), dim3(256), 0, stream, dy, x, mean, rstd, g_res, g_out, gb, colsum, M, D, dthresh, dscale, slo, shi);
#undef LN_BWD
  return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
}

extern "C" int vitamd_layernorm_bwd(const void* dy_bf16, const float* x, const float* mean, const float* rstd,
                                    const float* g_res, float* g_out, void* g_bf16, float* colsum, int M, int D,
                                    void* stream) {
  return ln_bwd_launch(dy_bf16, x, mean, rstd, g_res, g_out, g_bf16, colsum, M, D, 0.f, 0ull, stream);
}

// same; the bf16 copy (and its column sums) additionally gets the dropout mask (p, seed) of the
// Linear output whose gradient it is (index = row * D + column, as in vitamd_linear_dropout_resid_bf16)
extern "C" int vitamd_layernorm_bwd_dropout(const void* dy_bf16, const float* x, const float* mean, const float* rstd,
                                            const float* g_res, float* g_out, void* g_bf16, float* colsum, int M, int D,
                                            float dropout_p, unsigned long long seed, void* stream) {
  return ln_bwd_launch(dy_bf16, x, mean, rstd, g_res, g_out, g_bf16, colsum, M, D, dropout_p, seed, stream);
}

// LayerNorm backward with xhat taken from the forward's bf16 output (see ln_bwd_kernel<NV, true>): D in {256, 512, 768, 1024}.
extern "C" int vitamd_layernorm_bwd_xhat(const void* dy_bf16, const void* y_bf16, const float* rstd, const float* g_res, float* g_out,
                                         void* g_bf16, float* colsum, int M, int D, float dropout_p, unsigned long long seed,
                                         void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (M <= 0 || (D != 256 && D != 512 && D != 768 && D != 1024)) return VITAMD_ERR_SHAPE;
  if (!dy_bf16 || !y_bf16 || !rstd || !g_out) return VITAMD_ERR_ARG;
  if (!(dropout_p >= 0.f) || dropout_p >= 1.f) return VITAMD_ERR_ARG;
  unsigned dthresh = dropout_p > 0.f ? (unsigned)((double)dropout_p * 4294967296.0) : 0u;
  if (dropout_p > 0.f && dthresh == 0u) dthresh = 1u;
  const float dscale = 1.0f / (1.0f - dropout_p);
  const unsigned slo = (unsigned)seed, shi = (unsigned)(seed >> 32);
  const __bf16* dy = (const __bf16*)dy_bf16;
  const float* yx = (const float*)y_bf16;        // the kernel reinterprets it (XH = true)
  __bf16* gb = (__bf16*)g_bf16;
  const int grid = grid_for(M, colsum != nullptr && g_bf16 != nullptr);
#define LN_BWDX(NV) hipLaunchKernelGGL((ln_bwd_kernel<NV, true>), dim3(grid), dim3(256), 0, stream, dy, yx, (const float*)nullptr, rstd, g_res, g_out, gb, colsum, M, dthresh, dscale, slo, shi)
  if (D == 256) { LN_BWDX(1); }
  else if (D == 512) { LN_BWDX(2); }
  else if (D == 768) { LN_BWDX(3); }
  else { LN_BWDX(4); }
#undef LN_BWDX
  return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
}
