// Small HBM-bound helpers around the GEMMs: dtype casts, weight transpose, patch gather (im2col),
// column sums (bias gradients) and the patch-embed backward reduction.
#include "common.h"

namespace {

// fp32 -> bf16, 8 elements per lane per iteration (32-B loads, 16-B stores)
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ in, __bf16* __restrict__ out, size_t n8, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    const f32x4 a = *(const f32x4*)(in + i * 8), b = *(const f32x4*)(in + i * 8 + 4);
    u32x4 o = {pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
    *(u32x4*)(out + i * 8) = o;
  }
  if (blockIdx.x == 0) for (size_t i = n8 * 8 + threadIdx.x; i < n; i += blockDim.x) out[i] = f2bf(in[i]);
}

// bf16(x) with a dropout mask on the rounded value: the top layer's fc2 output gradient when dropout is on
__global__ __launch_bounds__(256) void cast_dropout_kernel(const float* __restrict__ in, __bf16* __restrict__ out, size_t n,
                                                           unsigned thresh, float scale, unsigned slo, unsigned shi) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = f2bf(round_bf16(in[i]) * dropout_keep(i, slo, shi, thresh, scale));
}

// out[i] = in[i] * keep(seed, i / group): dropout with one mask decision per `group` consecutive elements.  group = 1 is nn.Dropout
// (reference blocks.py:118,168-170: proj_drop, Mlp.drop); group = elements per sample is DropPath (blocks.py:124-139, one decision per
// sample, scaled by 1/keep_prob).  The backward applies the same call to the gradient (same seed -> same mask).  In place is fine.
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ in, T* __restrict__ out, size_t n, size_t group, unsigned thresh,
                                                      float scale, unsigned slo, unsigned shi) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = (T)((float)in[i] * dropout_keep(group == 1 ? i : i / group, slo, shi, thresh, scale));
}

// W fp32 [N,K] -> Wb bf16 [N,K] (optional) and WbT bf16 [K,N] (optional); 64x64 tiles through LDS
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ w, __bf16* __restrict__ wb,
                                                             __bf16* __restrict__ wbt, int N, int K) {
  __shared__ float tile[64][65];
  const int n0 = blockIdx.y * 64, k0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int n = n0 + r, k = k0 + tx;
    float v = 0.f;
    if (n < N && k < K) {
      v = w[(size_t)n * K + k];
      if (wb) wb[(size_t)n * K + k] = f2bf(v);
    }
    tile[r][tx] = v;
  }
  __syncthreads();
  if (wbt) {
    for (int r = ty; r < 64; r += 4) {
      const int k = k0 + r, n = n0 + tx;
      if (k < K && n < N) wbt[(size_t)k * N + n] = f2bf(tile[tx][r]);
    }
  }
}

// Batched form: one launch casts (and transposes) EVERY weight of the model.  desc[i] =
// {w, wb, wbt, N, K, first_tile}; blockIdx.x walks the concatenated 64x64 tile lists.
struct CastDesc { const float* w; __bf16* wb; __bf16* wbt; int N, K, first_tile, tiles_k; };
__global__ __launch_bounds__(256) void cast_transpose_batched_kernel(const CastDesc* __restrict__ desc, int n) {
  __shared__ float tile[64][65];
  __shared__ int which;
  if (threadIdx.x == 0) {
    int lo = 0, hi = n - 1;           // last descriptor whose first_tile <= blockIdx.x
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (desc[mid].first_tile <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
    which = lo;
  }
  __syncthreads();
  const CastDesc d = desc[which];
  const int t = blockIdx.x - d.first_tile;
  const int n0 = (t / d.tiles_k) * 64, k0 = (t % d.tiles_k) * 64;
  if ((d.K & 3) == 0 && (d.N & 3) == 0) {
    // vector path (every weight of the models here): 16-B fp32 loads, 8-B bf16 stores in both orientations
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = it * 256 + threadIdx.x, r = idx >> 4, c = (idx & 15) * 4;
      const int nn = n0 + r, k = k0 + c;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (nn < d.N && k < d.K) {        // K % 4 == 0: the four columns are in range together
        v = *(const f32x4*)(d.w + (size_t)nn * d.K + k);
        if (d.wb) *(u32x2*)(d.wb + (size_t)nn * d.K + k) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) tile[r][c + j] = v[j];
    }
    __syncthreads();
    if (d.wbt) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int idx = it * 256 + threadIdx.x, kl = idx >> 4, c = (idx & 15) * 4;
        const int k = k0 + kl, nn = n0 + c;
        if (k < d.K && nn < d.N)
          *(u32x2*)(d.wbt + (size_t)k * d.N + nn) = u32x2{pack_bf16x2(tile[c][kl], tile[c + 1][kl]), pack_bf16x2(tile[c + 2][kl], tile[c + 3][kl])};
      }
    }
    return;
  }
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int nn = n0 + r, k = k0 + tx;
    float v = 0.f;
    if (nn < d.N && k < d.K) {
      v = d.w[(size_t)nn * d.K + k];
      if (d.wb) d.wb[(size_t)nn * d.K + k] = f2bf(v);
    }
    tile[r][tx] = v;
  }
  __syncthreads();
  if (d.wbt) {
    for (int r = ty; r < 64; r += 4) {
      const int k = k0 + r, nn = n0 + tx;
      if (k < d.K && nn < d.N) d.wbt[(size_t)k * d.N + nn] = f2bf(tile[tx][r]);
    }
  }
}

// images fp32 [B,C,H,W] -> patches bf16 [B*gh*gw, C*p*p], patch vector order (c,kh,kw)
// (the contraction order of Conv2d(kernel=stride=p), reference train_vit.py:34,39).  One thread per
// 4 consecutive kw: 16-B fp32 loads along an image row, 8-B bf16 stores along the patch vector.
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ img, __bf16* __restrict__ out,
                                                     int B, int C, int H, int W, int p) {
  const int gh = H / p, gw = W / p, pd = C * p * p, p4 = p / 4;
  const size_t total = (size_t)B * gh * gw * (pd / 4);
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    // decompose so consecutive threads walk an image row: (b, c, y = ph*p+kh, pw, kw4)
    size_t t = i;
    const int kw4 = t % p4; t /= p4;
    const int pw = t % gw; t /= gw;
    const int kh = t % p; t /= p;
    const int ph = t % gh; t /= gh;
    const int c = t % C; const int b = t / C;
    const f32x4 v = *(const f32x4*)(img + (((size_t)b * C + c) * H + ph * p + kh) * W + pw * p + kw4 * 4);
    u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    *(u32x2*)(out + ((size_t)(b * gh + ph) * gw + pw) * pd + (c * p + kh) * p + kw4 * 4) = o;
  }
}

__global__ __launch_bounds__(256) void im2col_generic_kernel(const float* __restrict__ img, __bf16* __restrict__ out,
                                                             int B, int C, int H, int W, int p) {
  const int gh = H / p, gw = W / p, pd = C * p * p;
  const size_t total = (size_t)B * gh * gw * pd;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    size_t t = i;
    const int kw = t % p; t /= p;
    const int kh = t % p; t /= p;
    const int c = t % C; t /= C;
    const int pw = t % gw; t /= gw;
    const int ph = t % gh; const int b = t / gh;
    out[i] = f2bf(img[(((size_t)b * C + c) * H + ph * p + kh) * W + pw * p + kw]);
  }
}

// colsum[n] += sum_m X[m,n]  (X bf16 [M, ld]).  16-B loads: a thread owns 8 columns and every 8th
// row of its block's row slab; the 8 row-groups of a block are combined through LDS, then one
// shaped atomic per column.
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const __bf16* __restrict__ x, float* __restrict__ out, int M, int N, int ld, int rows_per_block) {
  __shared__ float red[8][32 * 8];
  const int cg = threadIdx.x & 31, rg = threadIdx.x >> 5;       // 32 column groups x 8 row groups
  const int n = (blockIdx.x * 32 + cg) * 8;
  const int m_lo = blockIdx.y * rows_per_block, m_hi = min(M, m_lo + rows_per_block);
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (n < N) {
    for (int m = m_lo + rg; m < m_hi; m += 8) {
      const u32x4 v = *(const u32x4*)(x + (size_t)m * ld + n);
#pragma unroll
      for (int j = 0; j < 4; ++j) { s[2 * j] += bf16lo(v[j]); s[2 * j + 1] += bf16hi(v[j]); }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[rg][cg * 8 + j] = s[j];
  __syncthreads();
  const int c = threadIdx.x;
  float t = 0.f;
#pragma unroll
  for (int r = 0; r < 8; ++r) t += red[r][c];
  const int nn = blockIdx.x * 256 + c;
  if (nn < N) atomicAdd(out + nn, t);
}

__global__ __launch_bounds__(256) void colsum_bf16_generic_kernel(const __bf16* __restrict__ x, float* __restrict__ out, int M, int N, int ld, int rows_per_block) {
  const int m_lo = blockIdx.y * rows_per_block, m_hi = min(M, m_lo + rows_per_block);
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  float s = 0.f;
  for (int m = m_lo; m < m_hi; ++m) s += bf2f(x[(size_t)m * ld + n]);
  atomicAdd(out + n, s);
}

// Patch-embed backward reduction over the batch (reference train_vit.py:41-44 backward):
//   g fp32 [B, seq, D] (gradient at the transformer input)
//   dpos[p]   = sum_b g[b, extra+p]      dextra[e] = sum_b g[b, e]
//   dyp bf16 [B*np, D] = bf16(g[b, extra+p])  (compact rows for the conv weight-gradient GEMM)
//   dbias[n] += sum_{b,p} bf16(g[b, extra+p, n])
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ g, float* __restrict__ dpos,
                                                        float* __restrict__ dextra, __bf16* __restrict__ dyp,
                                                        float* __restrict__ dbias_rows, int B, int seq, int extra, int D, int bchunk) {
  const int t = blockIdx.x;  // token position
  const int b_lo = blockIdx.y * bchunk, b_hi = min(B, b_lo + bchunk);
  const int np = seq - extra;
  if ((D & 3) == 0) {          // 16-B fp32 loads, 8-B bf16 stores (an HBM-bound pass over g: 232 MB at the headline shape)
    for (int c = threadIdx.x * 4; c < D; c += 1024) {
      f32x4 s = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
      for (int b = b_lo; b < b_hi; ++b) {
        const f32x4 v = *(const f32x4*)(g + ((size_t)b * seq + t) * D + c);
        s += v;
        if (t >= extra) {
          const unsigned lo = pack_bf16x2(v[0], v[1]), hi = pack_bf16x2(v[2], v[3]);
          *(u32x2*)(dyp + ((size_t)b * np + (t - extra)) * D + c) = u32x2{lo, hi};
          sb += f32x4{bf16lo(lo), bf16hi(lo), bf16lo(hi), bf16hi(hi)};
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (t < extra) atomicAdd(dextra + (size_t)t * D + c + j, s[j]);
        else {
          atomicAdd(dpos + (size_t)(t - extra) * D + c + j, s[j]);
          atomicAdd(dbias_rows + (size_t)(t - extra) * D + c + j, sb[j]);
        }
      }
    }
    return;
  }
  for (int c = threadIdx.x; c < D; c += 256) {
    float s = 0.f, sb = 0.f;
    for (int b = b_lo; b < b_hi; ++b) {
      const float v = g[((size_t)b * seq + t) * D + c];
      s += v;
      if (t >= extra) {
        const __bf16 vb = f2bf(v);
        dyp[((size_t)b * np + (t - extra)) * D + c] = vb;
        sb += bf2f(vb);
      }
    }
    if (t < extra) atomicAdd(dextra + (size_t)t * D + c, s);   // outputs are zeroed by the caller
    else {
      atomicAdd(dpos + (size_t)(t - extra) * D + c, s);
      // per-position partial of the conv-bias gradient: few adders per address (B/bchunk); the
      // positions are summed by embed_bias_reduce_kernel (a single hot [D] vector would serialise
      // seq * B/bchunk workgroups on the same 768 addresses)
      atomicAdd(dbias_rows + (size_t)(t - extra) * D + c, sb);
    }
  }
}

// dbias[c] += sum_r rows[r][c]: 64 columns per workgroup, four row groups summed in parallel and folded through LDS (fixed order)
__global__ __launch_bounds__(256) void embed_bias_reduce_kernel(const float* __restrict__ rows, float* __restrict__ dbias, int np, int D) {
  __shared__ float part[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  float s = 0.f;
  if (c < D) {
#pragma unroll 8
    for (int r = ty; r < np; r += 4) s += rows[(size_t)r * D + c];
  }
  part[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && c < D) dbias[c] += (part[0][tx] + part[1][tx]) + (part[2][tx] + part[3][tx]);
}

// idx[m] = argmin_k || x[m,:] - e[k,:] ||^2 (first minimum), fp32; the nearest-code search of the
// VQ quantisers (reference train_titok.py:53 `torch.cdist(x, embedding).argmin(dim=-1)`).
// One thread per row, the codebook streamed through LDS in chunks of 256 codes.
template <int DMAX>
__global__ __launch_bounds__(256) void vq_nearest_kernel(const float* __restrict__ x, const float* __restrict__ e,
                                                         long long* __restrict__ idx, int M, int K, int d) {
  __shared__ float ce[256 * DMAX];
  const int m = blockIdx.x * 256 + threadIdx.x;
  float xr[DMAX];
#pragma unroll
  for (int c = 0; c < DMAX; ++c) xr[c] = (m < M && c < d) ? x[(size_t)m * d + c] : 0.f;
  float best = 3.0e38f;
  int besti = 0;
  for (int k0 = 0; k0 < K; k0 += 256) {
    const int nk = min(256, K - k0);
    __syncthreads();
    for (int i = threadIdx.x; i < nk * d; i += 256) ce[(i / d) * DMAX + (i % d)] = e[(size_t)k0 * d + i];
    __syncthreads();
    for (int k = 0; k < nk; ++k) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < DMAX; ++c) {
        if (c < d) { const float t = xr[c] - ce[k * DMAX + c]; s += t * t; }
      }
      if (s < best) { best = s; besti = k0 + k; }
    }
  }
  if (m < M) idx[m] = besti;
}

// The same search cut over the codebook too (round 4): grid = (row blocks) x (256-code chunks); every thread scans ONE chunk for its row with the
// arithmetic of vq_nearest_kernel and folds (distance bits << 32 | index) into its row's slot of the index buffer with a 64-bit atomicMin - a
// non-negative float orders like its bit pattern, so the minimum of the packed words is the smallest distance and, among equal distances, the
// smallest index: exactly the first-minimum tie-break of the sequential scan, whatever order the chunks arrive in.  vq_unpack_kernel then keeps the low
// half.  TiTok-S (8 192 tokens x 2 048 codes x 12) ran on 32 workgroups for 1.08 ms with one thread per row; 256 workgroups now.
template <int DMAX>
__global__ __launch_bounds__(256) void vq_nearest_chunk_kernel(const float* __restrict__ x, const float* __restrict__ e,
                                                               unsigned long long* __restrict__ packed, int M, int K, int d) {
  __shared__ float ce[256 * DMAX];
  const int m = blockIdx.x * 256 + threadIdx.x;
  const int k0 = blockIdx.y * 256, nk = min(256, K - k0);
  float xr[DMAX];
#pragma unroll
  for (int c = 0; c < DMAX; ++c) xr[c] = (m < M && c < d) ? x[(size_t)m * d + c] : 0.f;
  for (int i = threadIdx.x; i < nk * d; i += 256) ce[(i / d) * DMAX + (i % d)] = e[(size_t)k0 * d + i];
  __syncthreads();
  float best = 3.0e38f;
  int besti = k0;
  for (int k = 0; k < nk; ++k) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < DMAX; ++c) {
      if (c < d) { const float t = xr[c] - ce[k * DMAX + c]; s += t * t; }
    }
    if (s < best) { best = s; besti = k0 + k; }
  }
  if (m < M) atomicMin(packed + m, ((unsigned long long)__builtin_bit_cast(unsigned, best) << 32) | (unsigned)besti);
}
__global__ __launch_bounds__(256) void vq_unpack_kernel(unsigned long long* __restrict__ packed, int M) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m < M) packed[m] &= 0xffffffffull;
}

// Wide codes (64 < d <= 1024, e.g. the reference blocks.VectorQuantizer default token_size 256, blocks.py:408):
// a block owns 8 rows (staged in LDS); every thread walks the codes k = tid, tid+256, ... computing the 8
// distances of its code in one pass over the code's floats (the codebook is a few MB and stays in L2), then
// the block reduces (distance, index) per row with the same first-minimum tie-break as the narrow kernel.
__global__ __launch_bounds__(256) void vq_nearest_wide_kernel(const float* __restrict__ x, const float* __restrict__ e,
                                                              long long* __restrict__ idx, int M, int K, int d) {
  constexpr int RB = 8;
  extern __shared__ float xs[];            // [RB][d]
  __shared__ float rbest[4][RB];
  __shared__ int ribest[4][RB];
  const int m0 = blockIdx.x * RB;
  for (int i = threadIdx.x; i < RB * d; i += 256) {
    const int r = i / d;
    xs[i] = (m0 + r < M) ? x[(size_t)(m0 + r) * d + (i - r * d)] : 0.f;
  }
  __syncthreads();
  float best[RB];
  int besti[RB];
#pragma unroll
  for (int r = 0; r < RB; ++r) { best[r] = 3.0e38f; besti[r] = 0x7fffffff; }
  for (int k = threadIdx.x; k < K; k += 256) {
    const float* ek = e + (size_t)k * d;
    float s[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) s[r] = 0.f;
    for (int c = 0; c < d; ++c) {
      const float ev = ek[c];
#pragma unroll
      for (int r = 0; r < RB; ++r) { const float t = xs[r * d + c] - ev; s[r] += t * t; }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) if (s[r] < best[r]) { best[r] = s[r]; besti[r] = k; }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    float b = best[r]; int bi = besti[r];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ob = __shfl_xor(b, off, 64);
      const int oi = __shfl_xor(bi, off, 64);
      if (ob < b || (ob == b && oi < bi)) { b = ob; bi = oi; }
    }
    if (lane == 0) { rbest[wave][r] = b; ribest[wave][r] = bi; }
  }
  __syncthreads();
  if (threadIdx.x < RB && m0 + threadIdx.x < M) {
    float b = rbest[0][threadIdx.x]; int bi = ribest[0][threadIdx.x];
    for (int w = 1; w < 4; ++w) {
      const float ob = rbest[w][threadIdx.x]; const int oi = ribest[w][threadIdx.x];
      if (ob < b || (ob == b && oi < bi)) { b = ob; bi = oi; }
    }
    idx[m0 + threadIdx.x] = bi;
  }
}

}  // namespace

extern "C" int vitamd_vq_nearest(const float* x, const float* codebook, long long* idx, int M, int K, int d, void* stream) {
  if (M <= 0 || K <= 0 || d <= 0 || d > 1024) return VITAMD_ERR_SHAPE;
  if (!x || !codebook || !idx) return VITAMD_ERR_ARG;
  if (d > 64) {
    hipLaunchKernelGGL(vq_nearest_wide_kernel, dim3((M + 7) / 8), dim3(256), (size_t)8 * d * sizeof(float), (hipStream_t)stream, x, codebook, idx, M, K, d);
    return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
  }
  const int grid = (M + 255) / 256;
  if (K > 256) {       // cut over the codebook as well: (row blocks) x (code chunks) workgroups + a 64-bit atomicMin per (row, chunk); see vq_nearest_chunk_kernel
    if (hipMemsetAsync(idx, 0xff, (size_t)M * sizeof(long long), (hipStream_t)stream) != hipSuccess) return VITAMD_ERR_LAUNCH;
    const dim3 g2(grid, (K + 255) / 256);
    if (d <= 16) hipLaunchKernelGGL(vq_nearest_chunk_kernel<16>, g2, dim3(256), 0, (hipStream_t)stream, x, codebook, (unsigned long long*)idx, M, K, d);
    else hipLaunchKernelGGL(vq_nearest_chunk_kernel<64>, g2, dim3(256), 0, (hipStream_t)stream, x, codebook, (unsigned long long*)idx, M, K, d);
    hipLaunchKernelGGL(vq_unpack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (unsigned long long*)idx, M);
    return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
  }
  if (d <= 16) hipLaunchKernelGGL(vq_nearest_kernel<16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, codebook, idx, M, K, d);
  else hipLaunchKernelGGL(vq_nearest_kernel<64>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, codebook, idx, M, K, d);
  return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
}

extern "C" int vitamd_cast_f32_bf16(const float* in, void* out_bf16, long n, void* stream) {
  if (n <= 0) return n == 0 ? VITAMD_OK : VITAMD_ERR_SHAPE;
  if (!in || !out_bf16) return VITAMD_ERR_ARG;
  const size_t n8 = (size_t)n / 8;
  int grid = (int)((n8 + 255) / 256); grid = grid < 1 ? 1 : (grid > 4096 ? 4096 : grid);
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, (__bf16*)out_bf16, n8, (size_t)n);
  return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
}

extern "C" int vitamd_cast_f32_bf16_dropout(const float* in, void* out_bf16, long n, float dropout_p, unsigned long long seed, void* stream) {
  if (n <= 0) return n == 0 ? VITAMD_OK : VITAMD_ERR_SHAPE;
  if (!in || !out_bf16 || !(dropout_p >= 0.f) || dropout_p >= 1.f) return VITAMD_ERR_ARG;
  unsigned thresh = dropout_p > 0.f ? (unsigned)((double)dropout_p * 4294967296.0) : 0u;
  if (dropout_p > 0.f && thresh == 0u) thresh = 1u;
  int grid = (int)(((size_t)n + 255) / 256); grid = grid > 8192 ? 8192 : grid;
  hipLaunchKernelGGL(cast_dropout_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, (__bf16*)out_bf16, (size_t)n, thresh,
                     1.0f / (1.0f - dropout_p), (unsigned)seed, (unsigned)(seed >> 32));
  return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
}

template <typename T>
static int launch_dropout(const void* in, void* out, long n, long group, float dropout_p, unsigned long long seed, void* stream) {
  if (n <= 0 || group <= 0) return n == 0 ? VITAMD_OK : VITAMD_ERR_SHAPE;
  if (!in || !out || !(dropout_p >= 0.f) || dropout_p >= 1.f) return VITAMD_ERR_ARG;
  unsigned thresh = dropout_p > 0.f ? (unsigned)((double)dropout_p * 4294967296.0) : 0u;
  if (dropout_p > 0.f && thresh == 0u) thresh = 1u;
  int grid = (int)(((size_t)n + 255) / 256); grid = grid > 8192 ? 8192 : grid;
  hipLaunchKernelGGL(dropout_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)in, (T*)out, (size_t)n, (size_t)group, thresh,
                     1.0f / (1.0f - dropout_p), (unsigned)seed, (unsigned)(seed >> 32));
  return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
}
extern "C" int vitamd_dropout_bf16(const void* in, void* out, long n, long group, float dropout_p, unsigned long long seed, void* stream) {
  return launch_dropout<__bf16>(in, out, n, group, dropout_p, seed, stream);
}
extern "C" int vitamd_dropout_f32(const float* in, float* out, long n, long group, float dropout_p, unsigned long long seed, void* stream) {
  return launch_dropout<float>(in, out, n, group, dropout_p, seed, stream);
}

extern "C" int vitamd_cast_transpose_weight(const float* w, void* wb, void* wbt, int N, int K, void* stream) {
  if (N <= 0 || K <= 0) return VITAMD_ERR_SHAPE;
  if (!w || (!wb && !wbt)) return VITAMD_ERR_ARG;
  hipLaunchKernelGGL(cast_transpose_kernel, dim3((K + 63) / 64, (N + 63) / 64), dim3(256), 0, (hipStream_t)stream, w, (__bf16*)wb, (__bf16*)wbt, N, K);
  return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
}

extern "C" int vitamd_cast_transpose_batched(const void* desc_dev, int n, int total_tiles, void* stream) {
  if (n <= 0 || total_tiles <= 0) return VITAMD_ERR_SHAPE;
  if (!desc_dev) return VITAMD_ERR_ARG;
  hipLaunchKernelGGL(cast_transpose_batched_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, (const CastDesc*)desc_dev, n);
  return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
}

extern "C" int vitamd_im2col_bf16(const float* img, void* out_bf16, int B, int C, int H, int W, int p, void* stream) {
  if (B <= 0 || C <= 0 || p <= 0 || H < p || W < p) return VITAMD_ERR_SHAPE;
  if (!img || !out_bf16) return VITAMD_ERR_ARG;
  const int gh = H / p, gw = W / p;
  if (p % 4 == 0 && W % 4 == 0) {
    const size_t total = (size_t)B * gh * gw * (C * p * p / 4);
    int grid = (int)((total + 255) / 256); grid = grid > 8192 ? 8192 : grid;
    hipLaunchKernelGGL(im2col_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, img, (__bf16*)out_bf16, B, C, H, W, p);
  } else {
    const size_t total = (size_t)B * gh * gw * C * p * p;
    int grid = (int)((total + 255) / 256); grid = grid > 8192 ? 8192 : grid;
    hipLaunchKernelGGL(im2col_generic_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, img, (__bf16*)out_bf16, B, C, H, W, p);
  }
  return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
}

extern "C" int vitamd_colsum_bf16(const void* x_bf16, float* out, int M, int N, int ld, void* stream) {
  if (M <= 0 || N <= 0 || ld < N) return VITAMD_ERR_SHAPE;
  if (!x_bf16 || !out) return VITAMD_ERR_ARG;
  const int gx = (N + 255) / 256;
  int gy = 2048 / gx; if (gy < 1) gy = 1;
  int rpb = (M + gy - 1) / gy; if (rpb < 64) rpb = 64;
  gy = (M + rpb - 1) / rpb;
  if (N % 8 == 0 && ld % 8 == 0)
    hipLaunchKernelGGL(colsum_bf16_kernel, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x_bf16, out, M, N, ld, rpb);
  else
    hipLaunchKernelGGL(colsum_bf16_generic_kernel, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x_bf16, out, M, N, ld, rpb);
  return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
}

extern "C" int vitamd_embed_bwd(const float* g, float* dpos, float* dextra, void* dyp_bf16, float* dbias, float* dbias_rows, int B,
                                int seq, int extra, int D, void* stream) {
  if (B <= 0 || seq <= 0 || extra < 0 || extra > seq || D <= 0) return VITAMD_ERR_SHAPE;
  if (!g || (seq > extra && (!dpos || !dyp_bf16 || !dbias || !dbias_rows)) || (extra > 0 && !dextra)) return VITAMD_ERR_ARG;
  const int bchunk = 32;
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(seq, (B + bchunk - 1) / bchunk), dim3(256), 0, (hipStream_t)stream, g, dpos, dextra, (__bf16*)dyp_bf16, dbias_rows, B, seq, extra, D, bchunk);
  if (seq > extra)
    hipLaunchKernelGGL(embed_bias_reduce_kernel, dim3((D + 63) / 64), dim3(256), 0, (hipStream_t)stream, dbias_rows, dbias, seq - extra, D);
  return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
}
