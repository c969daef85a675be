// 3x3, stride 1, padding 1 convolution over NCHW fp32 images with a handful of channels: the `conv_out =
// nn.Conv2d(3, 3, 3, padding=1)` smoothing layer that ends the reference's pixel decoders
// (reference blocks.py:333,355 and :402).  27 MACs per output value: this is HBM-bound streaming work
// (read the image once, write it once), so there is no GEMM here: one thread per pixel, the 3x3 neighbourhood
// of every input channel comes through L1/L2 (rows are contiguous, lanes walk along W), all output channels
// are produced by the same thread.  The weight gradient is a full reduction over B*H*W: per-thread
// register accumulators over a grid-stride loop, then wave shuffles, LDS across waves and one atomic per block.
#include "common.h"
#include "vitamd_internal.h"
#include "../../include/vitamd.h"

namespace {

template <int CI, int CO, bool FLIP>
__global__ __launch_bounds__(256) void conv3x3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ y, int B, int H, int W) {
  // FLIP = false: y[b,co] = bias[co] + sum_ci,kh,kw w[co,ci,kh,kw] * x[b,ci,h+kh-1,w+kw-1]        (x has CI channels)
  // FLIP = true : y[b,ci] = sum_co,kh,kw w[co,ci,kh,kw] * x[b,co,h+1-kh,w+1-kw]  (input gradient; x = dy has CO channels)
  constexpr int CIN = FLIP ? CO : CI, COUT = FLIP ? CI : CO;
  __shared__ float ws[CO * CI * 9];
  for (int i = threadIdx.x; i < CO * CI * 9; i += 256) ws[i] = w[i];
  __syncthreads();
  const size_t hw = (size_t)H * W, total = (size_t)B * hw;
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < total; p += (size_t)gridDim.x * 256) {
    const int b = (int)(p / hw);
    const int r = (int)(p - (size_t)b * hw);
    const int h = r / W, c = r - h * W;
    float acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = (!FLIP && bias) ? bias[o] : 0.f;
#pragma unroll
    for (int i = 0; i < CIN; ++i) {
      const float* xi = x + ((size_t)b * CIN + i) * hw;
#pragma unroll
      for (int dh = -1; dh <= 1; ++dh) {
        const int hh = h + dh;
        if (hh < 0 || hh >= H) continue;
#pragma unroll
        for (int dw = -1; dw <= 1; ++dw) {
          const int cc = c + dw;
          if (cc < 0 || cc >= W) continue;
          const float v = xi[(size_t)hh * W + cc];
#pragma unroll
          for (int o = 0; o < COUT; ++o) {
            // forward: tap (kh,kw) = (dh+1, dw+1); input gradient: x index h+1-kh => kh = 1-dh
            const int widx = FLIP ? ((i * CI + o) * 9 + (1 - dh) * 3 + (1 - dw)) : ((o * CI + i) * 9 + (dh + 1) * 3 + (dw + 1));
            acc[o] += ws[widx] * v;
          }
        }
      }
    }
#pragma unroll
    for (int o = 0; o < COUT; ++o) y[((size_t)b * COUT + o) * hw + r] = acc[o];
  }
}

template <int CI, int CO>
__global__ __launch_bounds__(256) void conv3x3_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ dw, float* __restrict__ db, int B, int H, int W) {
  constexpr int NW = CO * CI * 9, NA = NW + CO;
  float acc[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) acc[i] = 0.f;
  const size_t hw = (size_t)H * W, total = (size_t)B * hw;
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < total; p += (size_t)gridDim.x * 256) {
    const int b = (int)(p / hw);
    const int r = (int)(p - (size_t)b * hw);
    const int h = r / W, c = r - h * W;
    float g[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) { g[o] = dy[((size_t)b * CO + o) * hw + r]; acc[NW + o] += g[o]; }
#pragma unroll
    for (int i = 0; i < CI; ++i) {
      const float* xi = x + ((size_t)b * CI + i) * hw;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int hh = h + kh - 1, cc = c + kw - 1;
          const float v = (hh >= 0 && hh < H && cc >= 0 && cc < W) ? xi[(size_t)hh * W + cc] : 0.f;
#pragma unroll
          for (int o = 0; o < CO; ++o) acc[(o * CI + i) * 9 + kh * 3 + kw] += g[o] * v;
        }
      }
    }
  }
  __shared__ float red[4][NA];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    float v = acc[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) red[wave][i] = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NA; i += 256) {
    const float v = red[0][i] + red[1][i] + red[2][i] + red[3][i];
    if (i < NW) atomicAdd(dw + i, v);
    else if (db) atomicAdd(db + (i - NW), v);
  }
}

int grid_for(size_t total) {
  size_t g = (total + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

extern "C" int vitamd_conv3x3_fwd(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, int H, int W,
                                  void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin != 3 || Cout != 3) return VITAMD_ERR_SHAPE;
  if (!x || !w || !y) return VITAMD_ERR_ARG;
  hipLaunchKernelGGL((conv3x3_kernel<3, 3, false>), dim3(grid_for((size_t)B * H * W)), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, B, H, W);
  return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
}

extern "C" int vitamd_conv3x3_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int B, int Cin, int Cout,
                                  int H, int W, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin != 3 || Cout != 3) return VITAMD_ERR_SHAPE;
  if (!x || !w || !dy || (!dx && !dw)) return VITAMD_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (dx) hipLaunchKernelGGL((conv3x3_kernel<3, 3, true>), dim3(grid_for((size_t)B * H * W)), dim3(256), 0, s, dy, w, nullptr, dx, B, H, W);
  if (dw) {
    size_t g = ((size_t)B * H * W + 256 * 16 - 1) / (256 * 16);   // >= 16 pixels per thread before the block reduction
    hipLaunchKernelGGL((conv3x3_wgrad_kernel<3, 3>), dim3((int)(g < 1 ? 1 : (g > 1024 ? 1024 : g))), dim3(256), 0, s, x, dy, dw, db, B, H, W);
  }
  return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
}
