// Scaled-dot-product attention forward / backward for the packed fused-QKV layout of the
// reference (transformer.py:26-29): qkv [B, N, 3, H, 64] bf16 straight out of the QKV GEMM,
// output o [B, N, H*64] bf16 — the einops split/merge copies of transformer.py:27,29 never exist.
// softmax(q k^T / sqrt(dh) [+ causal -inf mask, transformer.py:22-25]) v ; dh = 64 ; N <= 512 in one LDS-resident
// chunk per head, longer sequences (to 16 384) through the two-sided tiling of the *_long_kernel forms.
//
// gfx950 design.  One workgroup (4 waves) per (batch, head); the whole K/V (or Q/dO) of the head
// sits in LDS as [N][64] bf16 tiles (128-B rows, LDS-DMA staged, chunk index XOR-swizzled so BOTH
// the row reads (ds_read_b128) and the transposed reads (ds_read_b64_tr_b16) are conflict-free:
// tools/lds_banks.py).  Each wave owns 32-row blocks of the "lane side" matrix and keeps its
// fragments in registers.  mfma_f32_32x32x16_bf16 with the reduction-side index on the accumulator
// ROWS, so an accumulator tile converts in place into the B operand of the next MFMA
// (cdna guide section 3, "accumulator tile as the next MFMA's operand") — softmax never leaves registers.
//
//   forward : lane = query.  S^T = K.Q^T ; online softmax ; O^T += V^T.P^T
//   bwd dQ  : lane = query.  S^T = K.Q^T ; dP^T = V.dO^T ; dS^T = P^T o (dP^T - delta) ; dQ^T += K^T.dS^T
//   bwd dKV : lane = key.    S = Q.K^T ; dP = dO.V^T ; dV^T += dO^T.P ; dK^T += Q^T.dS
// Scores are recomputed from Q, K and the forward's log-sum-exp (flash-attention backward).
#include "common.h"

namespace {

constexpr int DH = 64;
constexpr int MAX_N = 512;        // single-chunk kernels; longer sequences (up to MAX_N_LONG) take the *_long_kernel forms
constexpr int MAX_N_LONG = 16384;
constexpr float NEG_BIG = -1.0e30f;

typedef LDS_AS bf16x4* lds_bf16x4_ptr;

__device__ __forceinline__ int swz(int row) { return (((row >> 1) & 1) << 2) | ((row >> 3) & 3); }

// Stage rows [0, npad) of a [N][64]-per-head matrix (row stride ld elements) into an LDS tile.
// Rows >= N re-read row N-1 (finite data; masked by the callers).
__device__ __forceinline__ void stage_tile(const __bf16* __restrict__ g, int ld, int N, int npad, char* lds, int wave, int lane, int nwaves = 4) {
  const int pieces = npad / 8;
  for (int pc = wave; pc < pieces; pc += nwaves) {
    const int row = pc * 8 + (lane >> 3);
    const int logical = (lane & 7) ^ swz(row);
    const int grow = min(row, N - 1);
    glds16(g + (size_t)grow * ld + logical * 8, lds + pc * 1024);
  }
}

// A-operand fragment by ROW read: element j = Y[32T + (lane&31)][16kk + 8(lane>>5) + j]
__device__ __forceinline__ bf16x8 row_frag(const char* tile, int T, int kk, int lane) {
  const int rr = lane & 31;
  const int chunk = (2 * kk + (lane >> 5)) ^ swz(rr);
  return *(const bf16x8*)(tile + T * 4096 + rr * 128 + (chunk << 4));
}

// A-operand fragment by TRANSPOSED read, in the k order of an accumulator-derived B operand:
// element j = Y[32T + 16s + 8(j>>2) + 4(lane>>5) + (j&3)][32dt + (lane&31)]
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int T, int s, int dt, int lane) {
  const int h = lane >> 5, colhalf = (lane >> 4) & 1, qq = (lane >> 2) & 3, pp = lane & 3;
  const int chunk = 4 * dt + 2 * colhalf + (pp >> 1);
  bf16x4 part[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int rloc = 16 * s + 8 * u + 4 * h + qq;  // row inside the 32-row tile
    const int f = (((qq >> 1) & 1) << 2) | (2 * s + u);
    part[u] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(tile + T * 4096 + rloc * 128 + ((chunk ^ f) << 4) + (pp & 1) * 8));
  }
  return __builtin_shufflevector(part[0], part[1], 0, 1, 2, 3, 4, 5, 6, 7);
}

// registers 8s..8s+7 of a 32x32 accumulator -> bf16 B-operand fragment of k-step s
__device__ __forceinline__ bf16x8 acc_to_frag(const f32x16& x, int s) {
  bf16x8 f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (__bf16)x[8 * s + j];
  return f;
}

// lane-side fragments of a 32-row block: element j = X[r0 + (lane&31)][16kk + 8(lane>>5) + j]
__device__ __forceinline__ void load_lane_frags(const __bf16* __restrict__ g, int ld, int N, int r0, int lane, bf16x8 (&f)[4]) {
  const int row = min(r0 + (lane & 31), N - 1);
  const __bf16* p = g + (size_t)row * ld + 8 * (lane >> 5);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) f[kk] = *(const bf16x8*)(p + 16 * kk);
}

__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }  // bare v_exp_f32 (inputs <= 0 or masked)
__device__ __forceinline__ float max16(const f32x16& v) {
  float a = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), b = fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7]));
  float c = fmaxf(fmaxf(v[8], v[9]), fmaxf(v[10], v[11])), d = fmaxf(fmaxf(v[12], v[13]), fmaxf(v[14], v[15]));
  return fmaxf(fmaxf(a, b), fmaxf(c, d));
}

// store a transposed 64x32 accumulator pair  acc[dt][reg] = X^T[d][row]  as X[row][d] bf16, scaled
__device__ __forceinline__ void store_rows_T(__bf16* __restrict__ g, int ld, int N, int r0, int lane, const f32x16 (&acc)[2], float scale) {
  const int row = r0 + (lane & 31);
  if (row >= N) return;
  __bf16* p = g + (size_t)row * ld + 4 * (lane >> 5);
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      u32x2 o = {pack_bf16x2(acc[dt][4 * u] * scale, acc[dt][4 * u + 1] * scale),
                 pack_bf16x2(acc[dt][4 * u + 2] * scale, acc[dt][4 * u + 3] * scale)};
      *(u32x2*)(p + 32 * dt + 8 * u) = o;
    }
}

// Same, but transposed through a wave-private 4-KiB LDS image first so that every lane stores 16 B
// and a wave-instruction covers 8 whole 128-B head rows (4 store instructions instead of 16
// scattered 8-B ones: the output tail was store-issue bound).
__device__ __forceinline__ void store_rows_T_lds(__bf16* __restrict__ g, int ld, int N, int r0, int lane, const f32x16 (&acc)[2],
                                                 float scale, char* img, float* colacc = nullptr) {
  const int rr = lane & 31, h = lane >> 5;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      u32x2 o = {pack_bf16x2(acc[dt][4 * u] * scale, acc[dt][4 * u + 1] * scale),
                 pack_bf16x2(acc[dt][4 * u + 2] * scale, acc[dt][4 * u + 3] * scale)};
      // element columns 32dt + 8u + 4h .. +3  -> 16-B chunk 4dt + u, half h ; chunk XOR (row & 7)
      *(u32x2*)(img + rr * 128 + (((4 * dt + u) ^ (rr & 7)) << 4) + h * 8) = o;
    }
  // wave-private image: the same wave reads it back (the compiler orders LDS accesses of one wave)
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = (lane >> 3) + 8 * it, pc = lane & 7;
    const u32x4 v = *(const u32x4*)(img + row * 128 + pc * 16);
    if (r0 + row < N) *(u32x4*)(g + (size_t)(r0 + row) * ld + 8 * (pc ^ (row & 7))) = v;
  }
  if (colacc) {
    // column sums of the stored bf16 rows (the bias gradient of the QKV Linear): lane = column
    const int chunk = lane >> 3, within = (lane & 7) * 2;
    const int nrows = min(32, N - r0);
    float sacc = 0.f;
    for (int row = 0; row < nrows; ++row)
      sacc += bf2f(*(const __bf16*)(img + row * 128 + ((chunk ^ (row & 7)) << 4) + within));
    *colacc += sacc;
  }
}

// store_rows_T_lds + the residual add of the fp32 stream: xo[row][d] = xi[row][d] + bf16(acc) for the same 32 rows x 64 columns
__device__ __forceinline__ void store_rows_T_lds_resid(__bf16* __restrict__ g, int ld, int N, int r0, int lane, const f32x16 (&acc)[2],
                                                       float scale, char* img, const float* __restrict__ xi, float* __restrict__ xo, int ldx) {
  const int rr = lane & 31, h = lane >> 5;
  // residual loads first: their latency overlaps the LDS round trip
  f32x4 r[4][2];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = (lane >> 3) + 8 * it, pc = lane & 7;
    const float* px = xi + (size_t)min(r0 + row, N - 1) * ldx + 8 * (pc ^ (row & 7));
    r[it][0] = *(const f32x4*)px;
    r[it][1] = *(const f32x4*)(px + 4);
  }
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      u32x2 o = {pack_bf16x2(acc[dt][4 * u] * scale, acc[dt][4 * u + 1] * scale),
                 pack_bf16x2(acc[dt][4 * u + 2] * scale, acc[dt][4 * u + 3] * scale)};
      *(u32x2*)(img + rr * 128 + (((4 * dt + u) ^ (rr & 7)) << 4) + h * 8) = o;
    }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = (lane >> 3) + 8 * it, pc = lane & 7;
    const u32x4 v = *(const u32x4*)(img + row * 128 + pc * 16);
    if (r0 + row < N) {
      const int col = 8 * (pc ^ (row & 7));
      *(u32x4*)(g + (size_t)(r0 + row) * ld + col) = v;
      f32x4 o0 = {r[it][0][0] + bf16lo(v[0]), r[it][0][1] + bf16hi(v[0]), r[it][0][2] + bf16lo(v[1]), r[it][0][3] + bf16hi(v[1])};
      f32x4 o1 = {r[it][1][0] + bf16lo(v[2]), r[it][1][1] + bf16hi(v[2]), r[it][1][2] + bf16lo(v[3]), r[it][1][3] + bf16hi(v[3])};
      float* po = xo + (size_t)(r0 + row) * ldx + col;
      *(f32x4*)po = o0;
      *(f32x4*)(po + 4) = o1;
    }
  }
}

// Same through a 2-KiB image (16 rows per pass, two passes): the 8-wave kernels keep the LDS of a workgroup at 2 x npad x 128 + 8 x 2 KiB,
// the footprint of the 4-wave ones, so that two workgroups - now 16 waves - still share a CU.
template <bool RES = false>
__device__ __forceinline__ void store_rows_T_lds2k(__bf16* __restrict__ g, int ld, int N, int r0, int lane, const f32x16 (&acc)[2], float scale, char* img,
                                                   const float* __restrict__ xi = nullptr, float* __restrict__ xo = nullptr, int ldx = 0) {
  const int rr = lane & 31, h = lane >> 5;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    f32x4 r[2][2];
    if constexpr (RES) {                       // residual loads first: their latency overlaps the LDS round trip
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int row = (lane >> 3) + 8 * it, pc = lane & 7;
        const float* px = xi + (size_t)min(r0 + 16 * pass + row, N - 1) * ldx + 8 * (pc ^ (row & 7));
        r[it][0] = *(const f32x4*)px;
        r[it][1] = *(const f32x4*)(px + 4);
      }
    }
    if ((rr >> 4) == pass) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          u32x2 o = {pack_bf16x2(acc[dt][4 * u] * scale, acc[dt][4 * u + 1] * scale),
                     pack_bf16x2(acc[dt][4 * u + 2] * scale, acc[dt][4 * u + 3] * scale)};
          *(u32x2*)(img + (rr & 15) * 128 + (((4 * dt + u) ^ (rr & 7)) << 4) + h * 8) = o;
        }
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int row = (lane >> 3) + 8 * it, pc = lane & 7;                  // image row 0..15 = query row 16 pass + row
      const u32x4 v = *(const u32x4*)(img + row * 128 + pc * 16);
      const int grow = r0 + 16 * pass + row;
      if (grow < N) {
        const int col = 8 * (pc ^ (row & 7));
        *(u32x4*)(g + (size_t)grow * ld + col) = v;
        if constexpr (RES) {
          f32x4 o0 = {r[it][0][0] + bf16lo(v[0]), r[it][0][1] + bf16hi(v[0]), r[it][0][2] + bf16lo(v[1]), r[it][0][3] + bf16hi(v[1])};
          f32x4 o1 = {r[it][1][0] + bf16lo(v[2]), r[it][1][1] + bf16hi(v[2]), r[it][1][2] + bf16lo(v[3]), r[it][1][3] + bf16hi(v[3])};
          float* po = xo + (size_t)grow * ldx + col;
          *(f32x4*)po = o0;
          *(f32x4*)(po + 4) = o1;
        }
      }
    }
  }
}

struct AttnArgs {
  const __bf16* qkv;   // [B, N, 3, H, 64]
  __bf16* o;           // fwd out / bwd in  [B, N, H*64]
  float* lse2;         // [B, H, N] log2-domain log-sum-exp of the scaled scores
  const __bf16* d_o;   // bwd: [B, N, H*64]
  __bf16* dqkv;        // bwd: [B, N, 3, H, 64]
  float* delta;        // bwd: [B, H, N] rowsum(dO o O)
  float* dbias;        // bwd: optional [3*H*64] fp32, column sums of dqkv are ADDED (QKV bias gradient)
  int B, N, H;
  int causal;
  float scale_log2e;   // (1/sqrt(dh)) * log2(e)
  float scale;         // 1/sqrt(dh)
  // dropout on the softmax probabilities (reference transformer.py:28 dropout_p); thresh = p * 2^32, 0 = off
  unsigned drop_thresh;
  float drop_scale;
  unsigned seed_lo, seed_hi;
  // forward only, optional: fp32 residual stream [B*N, H*64]; with both set the kernel also writes resid_out = resid_in + bf16(o)
  // (transformer.py:44 `x = x + attn(...)`), so the LayerNorm that follows reads x once instead of x and o and writing x
  const float* resid_in;
  float* resid_out;
  int dbg;             // experimental builds: A/B bits (0 in production)
};

// Phase probe of experimental builds (dbg bit 15; tools/attn_phases.py): every wave sums the shader-clock cycles it spends in each
// phase (a tick drains its own loads and waits for the accumulator named) and adds them to g_attn_probe at exit.
#ifdef VITAMD_EXPERIMENTAL
constexpr int PROBE_WAVES = 16384;
__device__ unsigned long long g_attn_probe[2 * PROBE_WAVES * 8];   // [kernel slot][wave][phase]: one private row per wave (no atomics)
#define PROBE_DECL                                                  \
  const bool pr_on = VITAMD_DBG(a) & 0x8000;                        \
  const unsigned long long pr_wall0 = pr_on ? wall_clock64() : 0ull;  \
  unsigned long long pr_last = pr_on ? clock64() : 0ull, pr_acc[6] = {0, 0, 0, 0, 0, 0};
#define PROBE_TICK_(i, dep, WAITS)                                                         \
  if (pr_on) {                                                                             \
    const int d_ = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, (float)(dep)));  \
    asm volatile(WAITS ::"s"(d_) : "memory");                                              \
    const unsigned long long t_ = clock64();                                               \
    pr_acc[i] += t_ - pr_last;                                                             \
    pr_last = t_;                                                                          \
  }
#define PROBE_TICK(i, dep) PROBE_TICK_(i, dep, "s_waitcnt vmcnt(0) lgkmcnt(0)")
#define PROBE_TICK_NOVM(i, dep) PROBE_TICK_(i, dep, "s_waitcnt lgkmcnt(0)")
#define PROBE_END(slot)                                                                               \
  if (pr_on && lane == 0 && blockIdx.x * 4 + wave < PROBE_WAVES) {                                    \
    unsigned long long* row_ = g_attn_probe + ((size_t)(slot) * PROBE_WAVES + blockIdx.x * 4 + wave) * 8; \
    for (int i_ = 0; i_ < 6; ++i_) row_[i_] = pr_acc[i_];                                            \
    row_[6] = 1ull;                                                                                   \
  }
#else
#define PROBE_DECL
#define PROBE_TICK(i, dep)
#define PROBE_TICK_NOVM(i, dep)
#define PROBE_END(slot)
#endif

// keep-scale of probability (b, head, query, key): 1/(1-p) or 0
__device__ __forceinline__ float attn_keep(const AttnArgs& a, int bh, int query, int key) {
  const unsigned long long idx = ((unsigned long long)bh * a.N + query) * a.N + key;
  return dropout_keep(idx, a.seed_lo, a.seed_hi, a.drop_thresh, a.drop_scale);
}

// ------------------------------------------------------------------------------------------ forward
template <bool DROP>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x / a.H, hh = blockIdx.x % a.H;
  const int N = a.N, D3 = 3 * a.H * DH, D = a.H * DH;
  const int nt = (N + 31) / 32, npad = nt * 32;
  char* ktile = smem;
  char* vtile = smem + npad * 128;
  const __bf16* qbase = a.qkv + (size_t)b * N * D3 + hh * DH;
  stage_tile(qbase + D, D3, N, npad, ktile, wave, lane);
  stage_tile(qbase + 2 * D, D3, N, npad, vtile, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const float c = a.scale_log2e;
  for (int qb = wave; qb < nt; qb += 4) {
    const int q0 = qb * 32;
    const int qrow = q0 + (lane & 31);
    bf16x8 qf[4];
    load_lane_frags(qbase, D3, N, q0, lane, qf);
    f32x16 oacc[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float m = NEG_BIG, l = 0.f;
    const int t_end = a.causal ? min(nt, qb + 1) : nt;  // key tiles above the diagonal contribute nothing
    for (int T = 0; T < t_end; ++T) {
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(ktile, T, kk, lane), qf[kk], s, 0, 0, 0);
      if (32 * T + 32 > N || (a.causal && T == qb)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = 32 * T + acc_row(r, lane);
          if (!(key < N && (!a.causal || key <= qrow))) s[r] = NEG_BIG;
        }
      }
      float tmax = max16(s);
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64)) * c;
      const float mnew = fmaxf(m, tmax);
      const float alpha = fast_exp2(m - mnew);
      m = mnew;
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = fast_exp2(__builtin_fmaf(s[r], c, -mnew));
        psum += s[r];
        if constexpr (DROP) s[r] *= attn_keep(a, blockIdx.x, min(qrow, N - 1), min(32 * T + acc_row(r, lane), N - 1));
      }
      l = l * alpha + psum;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
#pragma unroll
      for (int sidx = 0; sidx < 2; ++sidx) {
        const bf16x8 pf = acc_to_frag(s, sidx);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(vtile, T, sidx, dt, lane), pf, oacc[dt], 0, 0, 0);
      }
    }
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    store_rows_T(a.o + (size_t)b * N * D + hh * DH, D, N, q0, lane, oacc, inv);
    if (lane < 32 && qrow < N) a.lse2[((size_t)b * a.H + hh) * N + qrow] = m + log2f(l);
  }
}

// ------------------------------------------------------------------------------------------ forward, N <= 256
// The whole score row of a query fits in registers (NKT tiles x 16 fp32), so there is no online
// rescaling: S for every key tile, one row maximum, exp2, then P.V.  Per element the VALU work is
// fma + v_exp + add + cvt (the kernel is VALU-bound at dh = 64, not MFMA-bound).
template <int NKT, bool DROP, bool CAUSAL, bool RES = false>
__global__ __launch_bounds__(256, 2) void attn_fwd_small_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  PROBE_DECL
  const int head = blockIdx.x;     // one workgroup per (batch, head); a persistent two-per-CU grid walking the heads is slower (backward 276 against 256 us)
  const int b = head / a.H, hh = head % a.H;
  const int N = a.N, D3 = 3 * a.H * DH, D = a.H * DH;
  constexpr int nt = NKT, npad = NKT * 32;
  char* ktile = smem;
  char* vtile = smem + npad * 128;
  char* oimg = smem + 2 * npad * 128 + wave * 4096;
  const __bf16* qbase = a.qkv + (size_t)b * N * D3 + hh * DH;
  stage_tile(qbase + D, D3, N, npad, ktile, wave, lane);
  stage_tile(qbase + 2 * D, D3, N, npad, vtile, wave, lane);
  const int wrot = (wave + head) & 3;   // rotate which wave gets the short list of query blocks
  // the Q rows of both of this wave's query blocks ride the same wait as the K/V staging (one memory round trip per head, not three)
  bf16x8 qf[4], qf2[4];
  load_lane_frags(qbase, D3, N, wrot * 32, lane, qf);
  if (wrot + 4 < nt) load_lane_frags(qbase, D3, N, wrot * 32 + 128, lane, qf2);
  else {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf2[kk] = qf[kk];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) asm volatile("" : "+v"(qf[kk]), "+v"(qf2[kk]));   // pins the loads above the wait (hipcc sinks plain loads to their first use)
  __syncthreads();
  PROBE_TICK(0, 0.f)

  const float c = a.scale_log2e;
  for (int qb = wrot; qb < nt; qb += 4) {
    const int q0 = qb * 32;
    const int qrow = q0 + (lane & 31);
    if (qb != wrot) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) qf[kk] = qf2[kk];
    }
    const int t_end = CAUSAL ? qb + 1 : nt;
    f32x16 s[NKT];
    float mx = NEG_BIG;
#pragma unroll
    for (int T = 0; T < NKT; ++T) {
      if (T < t_end) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[T][r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) s[T] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(ktile, T, kk, lane), qf[kk], s[T], 0, 0, 0);
        if (32 * T + 32 > N || (CAUSAL && T == qb)) {   // only boundary tiles need masking (wave-uniform)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = 32 * T + acc_row(r, lane);
            if (!(key < N && (!CAUSAL || key <= qrow))) s[T][r] = NEG_BIG;
          }
        }
        mx = fmaxf(mx, max16(s[T]));
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    PROBE_TICK(2, mx)
    const float mc = mx * c;
    float l = 0.f;
    f32x16 oacc[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
#pragma unroll
    for (int T = 0; T < NKT; ++T) {
      if (T < t_end) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float pexp = fast_exp2(__builtin_fmaf(s[T][r], c, -mc));
          l += pexp;
          if constexpr (DROP) pexp *= attn_keep(a, head, min(qrow, N - 1), min(32 * T + acc_row(r, lane), N - 1));
          s[T][r] = pexp;
        }
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx) {
          const bf16x8 pf = acc_to_frag(s[T], sidx);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
            oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(vtile, T, sidx, dt, lane), pf, oacc[dt], 0, 0, 0);
        }
      }
    }
    l += __shfl_xor(l, 32, 64);
    PROBE_TICK(3, l + oacc[0][15] + oacc[1][15])
    const float inv = 1.0f / l;
    if constexpr (RES)
      store_rows_T_lds_resid(a.o + (size_t)b * N * D + hh * DH, D, N, q0, lane, oacc, inv, oimg, a.resid_in + (size_t)b * N * D + hh * DH,
                             a.resid_out + (size_t)b * N * D + hh * DH, D);
    else
      store_rows_T_lds(a.o + (size_t)b * N * D + hh * DH, D, N, q0, lane, oacc, inv, oimg);
    if (lane < 32 && qrow < N) a.lse2[((size_t)b * a.H + hh) * N + qrow] = mc + log2f(l);
    PROBE_TICK_NOVM(4, 0.f)
  }
#ifdef VITAMD_EXPERIMENTAL
  if (pr_on) pr_acc[5] = wall_clock64() - pr_wall0;      // 100-MHz wall clock over the wave's life: calibrates the cycle counter
#endif
  PROBE_END(0)
}

// ------------------------------------------------------------------------------------------ forward, 129 <= N <= 256, EIGHT waves (round 3)
// attn_fwd_small_kernel gives each of its 4 waves up to two 32-row query blocks, needs 229 registers for the register-resident score row and is
// bound by memory round trips at 2 workgroups x 4 waves per CU (DESIGN.md sections 4.2, 4.4).  Here a workgroup has EIGHT waves - one query block
// each - that share the staged K / V, and the softmax is ONLINE over a rolled key-tile loop, so a wave needs 102 registers: two workgroups
// = 16 waves share a CU at the LDS footprint of the 4-wave kernel (2 x npad x 128 + 8 x 2 KiB).  93 against 103 us at B = 256, N = 197, H = 12.
// Non-causal, no dropout; RES: also writes resid_out = resid_in + bf16(o) (the 4-wave kernel's fused residual form).
// ------------------------------------------------------------------------------------------
template <int NKT, bool RES>
__global__ __launch_bounds__(512, 4) void attn_fwd_small8_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int head = blockIdx.x;
  const int b = head / a.H, hh = head % a.H;
  const int N = a.N, D3 = 3 * a.H * DH, D = a.H * DH;
  constexpr int nt = NKT, npad = NKT * 32;
  char* ktile = smem;
  char* vtile = smem + npad * 128;
  char* oimg = smem + 2 * npad * 128 + wave * 2048;
  const __bf16* qbase = a.qkv + (size_t)b * N * D3 + hh * DH;
  stage_tile(qbase + D, D3, N, npad, ktile, wave, lane, 8);
  stage_tile(qbase + 2 * D, D3, N, npad, vtile, wave, lane, 8);
  const int qb = (wave + head) & 7;            // rotate which waves sit out when the head has fewer than 8 query blocks
  const bool active = qb < nt;
  bf16x8 qf[4];
  load_lane_frags(qbase, D3, N, (active ? qb : 0) * 32, lane, qf);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) asm volatile("" : "+v"(qf[kk]));
  __syncthreads();
  if (!active) return;
  const float c = a.scale_log2e;
  const int q0 = qb * 32;
  const int qrow = q0 + (lane & 31);
  // ONLINE softmax (one key tile of scores in registers at a time): the register-resident score row of attn_fwd_small_kernel (NKT x 16 registers)
  // does not fit the 128 registers that 16 waves per CU leave a wave
  float m = NEG_BIG, l = 0.f;
  f32x16 oacc[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
#pragma unroll 1
  for (int T = 0; T < NKT; ++T) {
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(ktile, T, kk, lane), qf[kk], st, 0, 0, 0);
    if (32 * T + 32 > N) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (!(32 * T + acc_row(r, lane) < N)) st[r] = NEG_BIG;
    }
    float mt = max16(st);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float mn = fmaxf(m, mt);
    const float alpha = fast_exp2((m - mn) * c);        // first tile: exp2(-huge) = 0 on zero accumulators
    m = mn;
    const float mc = mn * c;
    float lt = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pexp = fast_exp2(__builtin_fmaf(st[r], c, -mc));
      lt += pexp;
      st[r] = pexp;
    }
    l = l * alpha + lt;
    if (T > 0) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
    }
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx) {
      const bf16x8 pf = acc_to_frag(st, sidx);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
        oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(vtile, T, sidx, dt, lane), pf, oacc[dt], 0, 0, 0);
    }
  }
  l += __shfl_xor(l, 32, 64);
  const float mc = m * c;
  const float inv = 1.0f / l;
  if constexpr (RES)
    store_rows_T_lds2k<true>(a.o + (size_t)b * N * D + hh * DH, D, N, q0, lane, oacc, inv, oimg, a.resid_in + (size_t)b * N * D + hh * DH,
                             a.resid_out + (size_t)b * N * D + hh * DH, D);
  else
    store_rows_T_lds2k(a.o + (size_t)b * N * D + hh * DH, D, N, q0, lane, oacc, inv, oimg);
  if (lane < 32 && qrow < N) a.lse2[((size_t)b * a.H + hh) * N + qrow] = mc + log2f(l);
}

// ------------------------------------------------------------------------------------------ backward, dQ
template <bool DROP, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x / a.H, hh = blockIdx.x % a.H;
  const int N = a.N, D3 = 3 * a.H * DH, D = a.H * DH;
  const int nt = (N + 31) / 32, npad = nt * 32;
  char* ktile = smem;
  char* vtile = smem + npad * 128;
  char* oimg = smem + 2 * npad * 128 + wave * 4096;
  const __bf16* qbase = a.qkv + (size_t)b * N * D3 + hh * DH;
  PROBE_DECL
  stage_tile(qbase + D, D3, N, npad, ktile, wave, lane);
  stage_tile(qbase + 2 * D, D3, N, npad, vtile, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  PROBE_TICK(0, 0.f)

  const float c = a.scale_log2e;
  const __bf16* obase = a.o + (size_t)b * N * D + hh * DH;
  const __bf16* dobase = a.d_o + (size_t)b * N * D + hh * DH;
  float csum_q = 0.f;
  for (int qb = (wave + blockIdx.x) & 3; qb < nt; qb += 4) {
    const int q0 = qb * 32;
    const int qrow = q0 + (lane & 31);
    bf16x8 qf[4], dof[4], of[4];
    load_lane_frags(qbase, D3, N, q0, lane, qf);
    load_lane_frags(dobase, D, N, q0, lane, dof);
    load_lane_frags(obase, D, N, q0, lane, of);
    float delta = 0.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int j = 0; j < 8; ++j) delta += (float)dof[kk][j] * (float)of[kk][j];
    delta += __shfl_xor(delta, 32, 64);
    const size_t stat = ((size_t)b * a.H + hh) * N + min(qrow, N - 1);
    const float lse2 = a.lse2[stat];
    if (lane < 32 && qrow < N) a.delta[stat] = delta;
    PROBE_TICK(1, delta + lse2)

    f32x16 dq[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
    const int t_end = CAUSAL ? min(nt, qb + 1) : nt;
    for (int T = 0; T < t_end; ++T) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(ktile, T, kk, lane), qf[kk], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(vtile, T, kk, lane), dof[kk], dp, 0, 0, 0);
      }
      PROBE_TICK(2, s[15] + dp[15])
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pexp = fast_exp2(__builtin_fmaf(s[r], c, -lse2));
        float dpr = dp[r];
        if constexpr (DROP) dpr *= attn_keep(a, blockIdx.x, min(qrow, N - 1), min(32 * T + acc_row(r, lane), N - 1));
        s[r] = pexp * (dpr - delta);  // dS^T (the 1/sqrt(dh) factor is applied once at the end)
      }
      if (32 * T + 32 > N || (CAUSAL && T == qb)) {   // boundary tiles: zero the masked keys
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = 32 * T + acc_row(r, lane);
          if (!(key < N && (!CAUSAL || key <= qrow))) s[r] = 0.f;
        }
      }
      PROBE_TICK(3, s[0] + s[15])
#pragma unroll
      for (int sidx = 0; sidx < 2; ++sidx) {
        const bf16x8 dsf = acc_to_frag(s, sidx);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(ktile, T, sidx, dt, lane), dsf, dq[dt], 0, 0, 0);
      }
      PROBE_TICK(4, dq[0][15] + dq[1][15])
    }
    store_rows_T_lds(a.dqkv + (size_t)b * N * D3 + hh * DH, D3, N, q0, lane, dq, a.scale, oimg, a.dbias ? &csum_q : nullptr);
    PROBE_TICK_NOVM(5, 0.f)
  }
  if (a.dbias) atomicAdd(a.dbias + hh * DH + lane, csum_q);   // 256 contiguous bytes per wave
  PROBE_END(0)
}

// ------------------------------------------------------------------------------------------ backward, dQ: software-pipelined form
// The loop above leaves the schedule to hipcc, which reads every LDS fragment right in front of the MFMA that consumes it and
// keeps the three stages of a key tile (S/dP products, exp + dS on the VALU, dQ products) strictly one after the other: the phase
// probe (tools/attn_phases.py) shows 1 660 cycles per tile against ~400 of matrix-pipe and ~300 of VALU work.  Here the tile count
// is a template parameter, the loop is unrolled and the stages of neighbouring tiles overlap by construction:
//   iteration T:  request K/V row fragments of tile T+2 and the transposed K fragments of tile T
//                 S, dP products of tile T+1 (fragments requested one iteration earlier)   } one scheduling region: hipcc interleaves
//                 exp, dS of tile T on the VALU                                            } the independent MFMA and VALU streams
//                 dQ += K^T.dS of tile T
// Same arithmetic in the same order as attn_bwd_dq_kernel (bit-identical results); non-causal, no dropout.
template <int NT>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_pipe_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  PROBE_DECL
  const int head = blockIdx.x;     // one workgroup per (batch, head); a persistent two-per-CU grid walking the heads is slower (backward 276 against 256 us)
  const int b = head / a.H, hh = head % a.H;
  const int N = a.N, D3 = 3 * a.H * DH, D = a.H * DH;
  constexpr int nt = NT, npad = NT * 32;
  char* ktile = smem;
  char* vtile = smem + npad * 128;
  char* oimg = smem + 2 * npad * 128 + wave * 4096;
  const __bf16* qbase = a.qkv + (size_t)b * N * D3 + hh * DH;
  stage_tile(qbase + D, D3, N, npad, ktile, wave, lane);
  stage_tile(qbase + 2 * D, D3, N, npad, vtile, wave, lane);
  const __bf16* obase = a.o + (size_t)b * N * D + hh * DH;
  const __bf16* dobase = a.d_o + (size_t)b * N * D + hh * DH;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  PROBE_TICK(0, 0.f)

  const float c = a.scale_log2e;
  float csum_q = 0.f;
  // (requesting the first block's Q, dO, O rows together with the K/V staging makes this kernel slower, 272 against 259 us per
  // backward: more requests in flight only queue longer - the forward, with a third of the lane-side traffic, gains 5 % from it)
  for (int qb = (wave + head) & 3; qb < nt; qb += 4) {
    const int q0 = qb * 32;
    const int qrow = q0 + (lane & 31);
    bf16x8 qf[4], dof[4];
    float delta = 0.f;
    {
      bf16x8 of[4];
      load_lane_frags(qbase, D3, N, q0, lane, qf);
      load_lane_frags(dobase, D, N, q0, lane, dof);
      load_lane_frags(obase, D, N, q0, lane, of);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int j = 0; j < 8; ++j) delta += (float)dof[kk][j] * (float)of[kk][j];
    }
    delta += __shfl_xor(delta, 32, 64);
    const size_t stat = ((size_t)b * a.H + hh) * N + min(qrow, N - 1);
    const float lse2 = a.lse2[stat];
    if (lane < 32 && qrow < N) a.delta[stat] = delta;
    PROBE_TICK(1, delta + lse2)

    f32x16 dq[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
    bf16x8 kr[2][4], vr[2][4];      // row fragments of K and V, two tiles in flight
    f32x16 sb[2], dpb[2];           // S^T and dP^T of the tile on the VALU and of the next one in the matrix pipe
    auto products = [&](int buf) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { sb[buf][r] = 0.f; dpb[buf][r] = 0.f; }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        sb[buf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr[buf][kk], qf[kk], sb[buf], 0, 0, 0);
        dpb[buf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vr[buf][kk], dof[kk], dpb[buf], 0, 0, 0);
      }
    };
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { kr[0][kk] = row_frag(ktile, 0, kk, lane); vr[0][kk] = row_frag(vtile, 0, kk, lane); }
    if (NT > 1) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) { kr[1][kk] = row_frag(ktile, 1, kk, lane); vr[1][kk] = row_frag(vtile, 1, kk, lane); }
    }
    products(0);
#pragma unroll
    for (int T = 0; T < NT; ++T) {
      const int cur = T & 1, nxt = cur ^ 1;
      bf16x8 ktr[2][2];
#pragma unroll
      for (int sidx = 0; sidx < 2; ++sidx)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) ktr[sidx][dt] = tr_frag(ktile, T, sidx, dt, lane);
      __builtin_amdgcn_sched_barrier(0);
      if (T + 1 < NT) products(nxt);
      if (T + 2 < NT) {             // tile T's row fragments were consumed by the products issued one iteration ago
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { kr[cur][kk] = row_frag(ktile, T + 2, kk, lane); vr[cur][kk] = row_frag(vtile, T + 2, kk, lane); }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pexp = fast_exp2(__builtin_fmaf(sb[cur][r], c, -lse2));
        sb[cur][r] = pexp * (dpb[cur][r] - delta);  // dS^T (the 1/sqrt(dh) factor is applied once at the end)
      }
      if (T == NT - 1 && 32 * T + 32 > N) {         // boundary tile: zero the masked keys
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (!(32 * T + acc_row(r, lane) < N)) sb[cur][r] = 0.f;
      }
#pragma unroll
      for (int sidx = 0; sidx < 2; ++sidx) {
        const bf16x8 dsf = acc_to_frag(sb[cur], sidx);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktr[sidx][dt], dsf, dq[dt], 0, 0, 0);
      }
    }
    PROBE_TICK(2, dq[0][15] + dq[1][15])
    store_rows_T_lds(a.dqkv + (size_t)b * N * D3 + hh * DH, D3, N, q0, lane, dq, a.scale, oimg, a.dbias ? &csum_q : nullptr);
    PROBE_TICK_NOVM(5, 0.f)
  }
  if (a.dbias) atomicAdd(a.dbias + hh * DH + lane, csum_q);   // 256 contiguous bytes per wave
  PROBE_END(0)
}

// ------------------------------------------------------------------------------------------ backward, dK and dV
template <bool DROP, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x / a.H, hh = blockIdx.x % a.H;
  const int N = a.N, D3 = 3 * a.H * DH, D = a.H * DH;
  const int nt = (N + 31) / 32, npad = nt * 32;
  char* qtile = smem;
  char* dotile = smem + npad * 128;
  float* lse_s = (float*)(smem + 2 * npad * 128);
  float* delta_s = lse_s + npad;
  char* oimg = smem + 2 * npad * 128 + 2 * npad * 4 + wave * 4096;
  const __bf16* qbase = a.qkv + (size_t)b * N * D3 + hh * DH;
  const __bf16* dobase = a.d_o + (size_t)b * N * D + hh * DH;
  PROBE_DECL
  stage_tile(qbase, D3, N, npad, qtile, wave, lane);
  stage_tile(dobase, D, N, npad, dotile, wave, lane);
  for (int i = threadIdx.x; i < npad; i += 256) {
    const size_t stat = ((size_t)b * a.H + hh) * N + min(i, N - 1);
    lse_s[i] = a.lse2[stat];
    delta_s[i] = a.delta[stat];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  PROBE_TICK(0, 0.f)

  const float c = a.scale_log2e;
  float csum_k = 0.f, csum_v = 0.f;
  for (int kb = (wave + blockIdx.x) & 3; kb < nt; kb += 4) {
    const int k0 = kb * 32;
    const int krow = k0 + (lane & 31);
    bf16x8 kf[4], vf[4];
    load_lane_frags(qbase + D, D3, N, k0, lane, kf);
    load_lane_frags(qbase + 2 * D, D3, N, k0, lane, vf);
    PROBE_TICK(1, (float)kf[3][7] + (float)vf[3][7])
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
    const int t_beg = CAUSAL ? kb : 0;  // queries before the key block never attend to it
    for (int T = t_beg; T < nt; ++T) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(qtile, T, kk, lane), kf[kk], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(dotile, T, kk, lane), vf[kk], dp, 0, 0, 0);
      }
      PROBE_TICK(2, s[15] + dp[15])
      f32x16 pmat;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int qr = 32 * T + 8 * u + 4 * (lane >> 5);
        const f32x4 lse4 = *(const f32x4*)(lse_s + qr);
        const f32x4 del4 = *(const f32x4*)(delta_s + qr);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * u + i;
          const float pexp = fast_exp2(__builtin_fmaf(s[r], c, -lse4[i]));
          float keep = 1.0f;
          if constexpr (DROP) keep = attn_keep(a, blockIdx.x, min(qr + i, N - 1), min(krow, N - 1));
          pmat[r] = pexp * keep;
          s[r] = pexp * (dp[r] * keep - del4[i]);
        }
        if (32 * T + 32 > N || k0 + 32 > N || (CAUSAL && T == kb)) {   // boundary tiles only (wave-uniform)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = 4 * u + i, query = qr + i;
            if (!(query < N && krow < N && (!CAUSAL || krow <= query))) { pmat[r] = 0.f; s[r] = 0.f; }
          }
        }
      }
      PROBE_TICK(3, s[0] + pmat[15])
#pragma unroll
      for (int sidx = 0; sidx < 2; ++sidx) {
        const bf16x8 pf = acc_to_frag(pmat, sidx);
        const bf16x8 dsf = acc_to_frag(s, sidx);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(dotile, T, sidx, dt, lane), pf, dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(qtile, T, sidx, dt, lane), dsf, dk[dt], 0, 0, 0);
        }
      }
      PROBE_TICK(4, dk[1][15] + dv[1][15])
    }
    __bf16* dbase = a.dqkv + (size_t)b * N * D3 + hh * DH;
    store_rows_T_lds(dbase + D, D3, N, k0, lane, dk, a.scale, oimg, a.dbias ? &csum_k : nullptr);
    store_rows_T_lds(dbase + 2 * D, D3, N, k0, lane, dv, 1.0f, oimg, a.dbias ? &csum_v : nullptr);
    PROBE_TICK_NOVM(5, 0.f)
  }
  if (a.dbias) {
    atomicAdd(a.dbias + D + hh * DH + lane, csum_k);
    atomicAdd(a.dbias + 2 * D + hh * DH + lane, csum_v);
  }
  PROBE_END(1)
}

// ------------------------------------------------------------------------------------------ backward, dK and dV: software-pipelined form
// Same restructuring as attn_bwd_dq_pipe_kernel (tile count a template parameter, unrolled, LDS fragments requested a stage
// ahead, the S/dP products of query tile T+1 issued before the VALU work of tile T).  Padded query rows need no masking here:
// their lse is staged as +1e30, so P and dS are exactly 0 for them; padded key lanes are never stored.  Bit-identical to
// attn_bwd_dkv_kernel; non-causal, no dropout.
template <int NT>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_pipe_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  PROBE_DECL
  const int head = blockIdx.x;     // one workgroup per (batch, head); a persistent two-per-CU grid walking the heads is slower (backward 276 against 256 us)
  const int b = head / a.H, hh = head % a.H;
  const int N = a.N, D3 = 3 * a.H * DH, D = a.H * DH;
  constexpr int nt = NT, npad = NT * 32;
  char* qtile = smem;
  char* dotile = smem + npad * 128;
  float* lse_s = (float*)(smem + 2 * npad * 128);
  float* delta_s = lse_s + npad;
  char* oimg = smem + 2 * npad * 128 + 2 * npad * 4 + wave * 4096;
  const __bf16* qbase = a.qkv + (size_t)b * N * D3 + hh * DH;
  const __bf16* dobase = a.d_o + (size_t)b * N * D + hh * DH;
  stage_tile(qbase, D3, N, npad, qtile, wave, lane);
  stage_tile(dobase, D, N, npad, dotile, wave, lane);
  for (int i = threadIdx.x; i < npad; i += 256) {
    const size_t stat = ((size_t)b * a.H + hh) * N + min(i, N - 1);
    lse_s[i] = i < N ? a.lse2[stat] : 1.0e30f;        // padded queries: exp2(s - 1e30) = 0 exactly
    delta_s[i] = a.delta[stat];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  PROBE_TICK(0, 0.f)

  const float c = a.scale_log2e;
  float csum_k = 0.f, csum_v = 0.f;
  for (int kb = (wave + head) & 3; kb < nt; kb += 4) {
    const int k0 = kb * 32;
    bf16x8 kf[4], vf[4];
    load_lane_frags(qbase + D, D3, N, k0, lane, kf);
    load_lane_frags(qbase + 2 * D, D3, N, k0, lane, vf);
    PROBE_TICK(1, (float)kf[3][7] + (float)vf[3][7])
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
    bf16x8 qr[4], dor[4];           // row fragments of Q and dO of the tile whose products are issued next
    f32x16 sb, dpb;                 // (one set: 256 registers do not hold a second one next to dK, dV and the fragments)
    auto products = [&]() {
#pragma unroll
      for (int r = 0; r < 16; ++r) { sb[r] = 0.f; dpb[r] = 0.f; }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        sb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qr[kk], kf[kk], sb, 0, 0, 0);
        dpb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dor[kk], vf[kk], dpb, 0, 0, 0);
      }
    };
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { qr[kk] = row_frag(qtile, 0, kk, lane); dor[kk] = row_frag(dotile, 0, kk, lane); }
    products();
    if (NT > 1) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) { qr[kk] = row_frag(qtile, 1, kk, lane); dor[kk] = row_frag(dotile, 1, kk, lane); }
    }
#pragma unroll
    for (int T = 0; T < NT; ++T) {
      bf16x8 dotr[2][2], qtr[2][2];
      f32x4 lse4[4], del4[4];
#pragma unroll
      for (int sidx = 0; sidx < 2; ++sidx)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) { dotr[sidx][dt] = tr_frag(dotile, T, sidx, dt, lane); qtr[sidx][dt] = tr_frag(qtile, T, sidx, dt, lane); }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int qr0 = 32 * T + 8 * u + 4 * (lane >> 5);
        lse4[u] = *(const f32x4*)(lse_s + qr0);
        del4[u] = *(const f32x4*)(delta_s + qr0);
      }
      f32x16 pmat;
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * u + i;
          const float pexp = fast_exp2(__builtin_fmaf(sb[r], c, -lse4[u][i]));
          pmat[r] = pexp;
          sb[r] = pexp * (dpb[r] - del4[u][i]);
        }
#pragma unroll
      for (int sidx = 0; sidx < 2; ++sidx) {
        const bf16x8 pf = acc_to_frag(pmat, sidx);
        const bf16x8 dsf = acc_to_frag(sb, sidx);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dotr[sidx][dt], pf, dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtr[sidx][dt], dsf, dk[dt], 0, 0, 0);
        }
      }
      if (T + 1 < NT) products();     // S, dP of the next tile queue up behind this tile's dV, dK products
      if (T + 2 < NT) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { qr[kk] = row_frag(qtile, T + 2, kk, lane); dor[kk] = row_frag(dotile, T + 2, kk, lane); }
      }
    }
    PROBE_TICK(2, dk[1][15] + dv[1][15])
    __bf16* dbase = a.dqkv + (size_t)b * N * D3 + hh * DH;
    store_rows_T_lds(dbase + D, D3, N, k0, lane, dk, a.scale, oimg, a.dbias ? &csum_k : nullptr);
    store_rows_T_lds(dbase + 2 * D, D3, N, k0, lane, dv, 1.0f, oimg, a.dbias ? &csum_v : nullptr);
    PROBE_TICK_NOVM(5, 0.f)
  }
  if (a.dbias) {
    atomicAdd(a.dbias + D + hh * DH + lane, csum_k);
    atomicAdd(a.dbias + 2 * D + hh * DH + lane, csum_v);
  }
  PROBE_END(1)
}

#ifdef VITAMD_EXPERIMENTAL
#include "experimental/attention_split.inc"
#endif

// ------------------------------------------------------------------------------------------ long sequences (N > 512)
// Same three algorithms with both sides tiled: the grid gets a second dimension over blocks of 128 "lane-side" rows
// (one 32-row block per wave: queries in forward / dQ, keys in dK-dV) and the kernel loops over LDS-staged chunks of
// CH = 512 rows of the other side (K,V or Q,dO + lse/delta), re-staging between two barriers.  The per-wave state
// (online-softmax m, l and the O accumulator; dQ; dK and dV) lives in registers across chunks.  K/V (Q/dO) are read
// ceil(N/128) times, from L2 after the first.
constexpr int CH = 512;

template <bool DROP>
__global__ __launch_bounds__(256) void attn_fwd_long_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x / a.H, hh = blockIdx.x % a.H;
  const int N = a.N, D3 = 3 * a.H * DH, D = a.H * DH;
  const int nt = (N + 31) / 32;
  char* ktile = smem;
  char* vtile = smem + CH * 128;
  const __bf16* qbase = a.qkv + (size_t)b * N * D3 + hh * DH;
  const int qb = blockIdx.y * 4 + wave;
  const bool active = qb < nt;                       // idle waves still stage and synchronise
  const int q0 = qb * 32, qrow = q0 + (lane & 31);
  const float c = a.scale_log2e;
  bf16x8 qf[4];
  if (active) load_lane_frags(qbase, D3, N, q0, lane, qf);
  f32x16 oacc[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
  float m = NEG_BIG, l = 0.f;
  const int t_end = a.causal ? min(nt, qb + 1) : nt;  // global key-tile bound
  for (int k0 = 0; k0 < N; k0 += CH) {
    const int rows = min(CH, N - k0), ntc = (rows + 31) / 32;
    __syncthreads();                                   // everyone is done with the previous chunk
    stage_tile(qbase + D + (size_t)k0 * D3, D3, rows, ntc * 32, ktile, wave, lane);
    stage_tile(qbase + 2 * D + (size_t)k0 * D3, D3, rows, ntc * 32, vtile, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!active) continue;
    for (int T = 0; T < ntc && k0 / 32 + T < t_end; ++T) {
      const int key0 = k0 + 32 * T;
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(ktile, T, kk, lane), qf[kk], s, 0, 0, 0);
      if (key0 + 32 > N || (a.causal && key0 / 32 == qb)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key0 + acc_row(r, lane);
          if (!(key < N && (!a.causal || key <= qrow))) s[r] = NEG_BIG;
        }
      }
      float tmax = max16(s);
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64)) * c;
      const float mnew = fmaxf(m, tmax);
      const float alpha = fast_exp2(m - mnew);
      m = mnew;
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = fast_exp2(__builtin_fmaf(s[r], c, -mnew));
        psum += s[r];
        if constexpr (DROP) s[r] *= attn_keep(a, blockIdx.x, min(qrow, N - 1), min(key0 + acc_row(r, lane), N - 1));
      }
      l = l * alpha + psum;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
#pragma unroll
      for (int sidx = 0; sidx < 2; ++sidx) {
        const bf16x8 pf = acc_to_frag(s, sidx);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(vtile, T, sidx, dt, lane), pf, oacc[dt], 0, 0, 0);
      }
    }
  }
  if (!active) return;
  l += __shfl_xor(l, 32, 64);
  const float inv = 1.0f / l;
  store_rows_T(a.o + (size_t)b * N * D + hh * DH, D, N, q0, lane, oacc, inv);
  if (lane < 32 && qrow < N) a.lse2[((size_t)b * a.H + hh) * N + qrow] = m + log2f(l);
}

template <bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_dq_long_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x / a.H, hh = blockIdx.x % a.H;
  const int N = a.N, D3 = 3 * a.H * DH, D = a.H * DH;
  const int nt = (N + 31) / 32;
  char* ktile = smem;
  char* vtile = smem + CH * 128;
  char* oimg = smem + 2 * CH * 128 + wave * 4096;
  const __bf16* qbase = a.qkv + (size_t)b * N * D3 + hh * DH;
  const __bf16* obase = a.o + (size_t)b * N * D + hh * DH;
  const __bf16* dobase = a.d_o + (size_t)b * N * D + hh * DH;
  const int qb = blockIdx.y * 4 + wave;
  const bool active = qb < nt;
  const int q0 = qb * 32, qrow = q0 + (lane & 31);
  const float c = a.scale_log2e;
  bf16x8 qf[4], dof[4];
  float delta = 0.f, lse2 = 0.f;
  if (active) {
    bf16x8 of[4];
    load_lane_frags(qbase, D3, N, q0, lane, qf);
    load_lane_frags(dobase, D, N, q0, lane, dof);
    load_lane_frags(obase, D, N, q0, lane, of);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int j = 0; j < 8; ++j) delta += (float)dof[kk][j] * (float)of[kk][j];
    delta += __shfl_xor(delta, 32, 64);
    const size_t stat = ((size_t)b * a.H + hh) * N + min(qrow, N - 1);
    lse2 = a.lse2[stat];
    if (lane < 32 && qrow < N) a.delta[stat] = delta;
  }
  f32x16 dq[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
  const int t_end = a.causal ? min(nt, qb + 1) : nt;
  for (int k0 = 0; k0 < N; k0 += CH) {
    const int rows = min(CH, N - k0), ntc = (rows + 31) / 32;
    __syncthreads();
    stage_tile(qbase + D + (size_t)k0 * D3, D3, rows, ntc * 32, ktile, wave, lane);
    stage_tile(qbase + 2 * D + (size_t)k0 * D3, D3, rows, ntc * 32, vtile, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!active) continue;
    for (int T = 0; T < ntc && k0 / 32 + T < t_end; ++T) {
      const int key0 = k0 + 32 * T;
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(ktile, T, kk, lane), qf[kk], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(vtile, T, kk, lane), dof[kk], dp, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pexp = fast_exp2(__builtin_fmaf(s[r], c, -lse2));
        float dpr = dp[r];
        if constexpr (DROP) dpr *= attn_keep(a, blockIdx.x, min(qrow, N - 1), min(key0 + acc_row(r, lane), N - 1));
        s[r] = pexp * (dpr - delta);
      }
      if (key0 + 32 > N || (a.causal && key0 / 32 == qb)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key0 + acc_row(r, lane);
          if (!(key < N && (!a.causal || key <= qrow))) s[r] = 0.f;
        }
      }
#pragma unroll
      for (int sidx = 0; sidx < 2; ++sidx) {
        const bf16x8 dsf = acc_to_frag(s, sidx);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(ktile, T, sidx, dt, lane), dsf, dq[dt], 0, 0, 0);
      }
    }
  }
  float csum_q = 0.f;
  if (active) store_rows_T_lds(a.dqkv + (size_t)b * N * D3 + hh * DH, D3, N, q0, lane, dq, a.scale, oimg, a.dbias ? &csum_q : nullptr);
  if (a.dbias && active) atomicAdd(a.dbias + hh * DH + lane, csum_q);
}

template <bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_dkv_long_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x / a.H, hh = blockIdx.x % a.H;
  const int N = a.N, D3 = 3 * a.H * DH, D = a.H * DH;
  const int nt = (N + 31) / 32;
  char* qtile = smem;
  char* dotile = smem + CH * 128;
  float* lse_s = (float*)(smem + 2 * CH * 128);
  float* delta_s = lse_s + CH;
  char* oimg = smem + 2 * CH * 128 + 2 * CH * 4 + wave * 4096;
  const __bf16* qbase = a.qkv + (size_t)b * N * D3 + hh * DH;
  const __bf16* dobase = a.d_o + (size_t)b * N * D + hh * DH;
  const int kb = blockIdx.y * 4 + wave;
  const bool active = kb < nt;
  const int k0 = kb * 32, krow = k0 + (lane & 31);
  const float c = a.scale_log2e;
  bf16x8 kf[4], vf[4];
  if (active) {
    load_lane_frags(qbase + D, D3, N, k0, lane, kf);
    load_lane_frags(qbase + 2 * D, D3, N, k0, lane, vf);
  }
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
  const int t_beg = a.causal ? kb : 0;                 // global query-tile bound: earlier queries never see this key block
  for (int r0 = 0; r0 < N; r0 += CH) {
    const int rows = min(CH, N - r0), ntc = (rows + 31) / 32;
    __syncthreads();
    stage_tile(qbase + (size_t)r0 * D3, D3, rows, ntc * 32, qtile, wave, lane);
    stage_tile(dobase + (size_t)r0 * D, D, rows, ntc * 32, dotile, wave, lane);
    for (int i = threadIdx.x; i < ntc * 32; i += 256) {
      const size_t stat = ((size_t)b * a.H + hh) * N + min(r0 + i, N - 1);
      lse_s[i] = a.lse2[stat];
      delta_s[i] = a.delta[stat];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!active) continue;
    for (int T = 0; T < ntc; ++T) {
      const int Tg = r0 / 32 + T;
      if (Tg < t_beg) continue;
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(qtile, T, kk, lane), kf[kk], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(dotile, T, kk, lane), vf[kk], dp, 0, 0, 0);
      }
      f32x16 pmat;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ql = 32 * T + 8 * u + 4 * (lane >> 5);        // row inside the staged chunk
        const f32x4 lse4 = *(const f32x4*)(lse_s + ql);
        const f32x4 del4 = *(const f32x4*)(delta_s + ql);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * u + i;
          const float pexp = fast_exp2(__builtin_fmaf(s[r], c, -lse4[i]));
          float keep = 1.0f;
          if constexpr (DROP) keep = attn_keep(a, blockIdx.x, min(r0 + ql + i, N - 1), min(krow, N - 1));
          pmat[r] = pexp * keep;
          s[r] = pexp * (dp[r] * keep - del4[i]);
        }
        if (r0 + 32 * T + 32 > N || k0 + 32 > N || (a.causal && Tg == kb)) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = 4 * u + i, query = r0 + ql + i;
            if (!(query < N && krow < N && (!a.causal || krow <= query))) { pmat[r] = 0.f; s[r] = 0.f; }
          }
        }
      }
#pragma unroll
      for (int sidx = 0; sidx < 2; ++sidx) {
        const bf16x8 pf = acc_to_frag(pmat, sidx);
        const bf16x8 dsf = acc_to_frag(s, sidx);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(dotile, T, sidx, dt, lane), pf, dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(qtile, T, sidx, dt, lane), dsf, dk[dt], 0, 0, 0);
        }
      }
    }
  }
  if (!active) return;
  float csum_k = 0.f, csum_v = 0.f;
  __bf16* dbase = a.dqkv + (size_t)b * N * D3 + hh * DH;
  store_rows_T_lds(dbase + D, D3, N, k0, lane, dk, a.scale, oimg, a.dbias ? &csum_k : nullptr);
  store_rows_T_lds(dbase + 2 * D, D3, N, k0, lane, dv, 1.0f, oimg, a.dbias ? &csum_v : nullptr);
  if (a.dbias) {
    atomicAdd(a.dbias + D + hh * DH + lane, csum_k);
    atomicAdd(a.dbias + 2 * D + hh * DH + lane, csum_v);
  }
}

int check(const AttnArgs& a) {
  if (a.B <= 0 || a.N <= 0 || a.H <= 0 || a.N > MAX_N_LONG) return VITAMD_ERR_SHAPE;
  return VITAMD_OK;
}

}  // namespace

static bool attn_dropout(AttnArgs& a, float p, unsigned long long seed) {
  if (!(p >= 0.f) || p >= 1.f) return false;
  a.drop_thresh = p > 0.f ? (unsigned)((double)p * 4294967296.0) : 0u;
  if (p > 0.f && a.drop_thresh == 0u) a.drop_thresh = 1u;
  a.drop_scale = 1.0f / (1.0f - p);
  a.seed_lo = (unsigned)seed;
  a.seed_hi = (unsigned)(seed >> 32);
  return true;
}

template <int K, bool DROP, bool CAUSAL>
static int launch_fwd_small_c(const AttnArgs& a, int lds, hipStream_t stream) {
  if (a.resid_in) {
    if (int e = set_lds((attn_fwd_small_kernel<K, DROP, CAUSAL, true>), lds)) return e;
    hipLaunchKernelGGL((attn_fwd_small_kernel<K, DROP, CAUSAL, true>), dim3(a.B * a.H), dim3(256), lds, stream, a);
    return VITAMD_OK;
  }
  if (int e = set_lds((attn_fwd_small_kernel<K, DROP, CAUSAL>), lds)) return e;
  hipLaunchKernelGGL((attn_fwd_small_kernel<K, DROP, CAUSAL>), dim3(a.B * a.H), dim3(256), lds, stream, a);
  return VITAMD_OK;
}
template <int K, bool DROP>
static int launch_fwd_small(const AttnArgs& a, int lds, hipStream_t stream) {      // causal / not: compile-time (fewer scalar registers: no runtime mask branches per tile)
  return a.causal ? launch_fwd_small_c<K, DROP, true>(a, lds, stream) : launch_fwd_small_c<K, DROP, false>(a, lds, stream);
}

static int attention_fwd_impl(const void* qkv, void* o, float* lse2, const float* resid_in, float* resid_out, int B, int N, int H,
                              int head_dim, int causal, float dropout_p, unsigned long long seed, void* stream_);

extern "C" int vitamd_attention_fwd(const void* qkv, void* o, float* lse2, int B, int N, int H, int head_dim, int causal,
                                    float dropout_p, unsigned long long seed, void* stream_) {
  return attention_fwd_impl(qkv, o, lse2, nullptr, nullptr, B, N, H, head_dim, causal, dropout_p, seed, stream_);
}

// forward + the residual add that follows it in the layer: resid_out = resid_in + o  (fp32 [B*N, H*64]); N <= 256 only
extern "C" int vitamd_attention_fwd_resid(const void* qkv, void* o, float* lse2, const float* resid_in, float* resid_out, int B, int N,
                                          int H, int head_dim, int causal, float dropout_p, unsigned long long seed, void* stream_) {
  if (!resid_in || !resid_out) return VITAMD_ERR_ARG;
  if (N > 256) return VITAMD_ERR_SHAPE;
  return attention_fwd_impl(qkv, o, lse2, resid_in, resid_out, B, N, H, head_dim, causal, dropout_p, seed, stream_);
}

static int attention_fwd_impl(const void* qkv, void* o, float* lse2, const float* resid_in, float* resid_out, int B, int N, int H,
                              int head_dim, int causal, float dropout_p, unsigned long long seed, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (head_dim != DH) return VITAMD_ERR_SHAPE;
  AttnArgs a{(const __bf16*)qkv, (__bf16*)o, lse2, nullptr, nullptr, nullptr, nullptr, B, N, H, causal, 0.125f * 1.4426950408889634f, 0.125f,
             0u, 1.0f, 0u, 0u, resid_in, resid_out, VITAMD_GDBG};
  if (int e = check(a)) return e;
  if (!qkv || !o || !lse2 || !attn_dropout(a, dropout_p, seed)) return VITAMD_ERR_ARG;
  const bool drop = a.drop_thresh != 0u;
  const int nkt = (N + 31) / 32, npad = nkt * 32;
  if (N > MAX_N) {
    const int lds = 2 * CH * 128;
    const dim3 grid(B * H, (nkt + 3) / 4);
    if (drop) {
      if (int e = set_lds(attn_fwd_long_kernel<true>, lds)) return e;
      hipLaunchKernelGGL(attn_fwd_long_kernel<true>, grid, dim3(256), lds, stream, a);
    } else {
      if (int e = set_lds(attn_fwd_long_kernel<false>, lds)) return e;
      hipLaunchKernelGGL(attn_fwd_long_kernel<false>, grid, dim3(256), lds, stream, a);
    }
    return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
  }
  if (nkt >= 5 && nkt <= 8 && !drop && !causal && !(VITAMD_GDBG & 0x8000)) {      // eight waves, one query block each (dbg bit 15 of experimental builds: off)
    const int lds = 2 * npad * 128 + 8 * 2048;
    int e = VITAMD_OK;
#define FWD_SMALL8(K) case K: \
      if (a.resid_in) { e = set_lds(attn_fwd_small8_kernel<K, true>, lds); if (!e) hipLaunchKernelGGL((attn_fwd_small8_kernel<K, true>), dim3(B * H), dim3(512), lds, stream, a); } \
      else { e = set_lds(attn_fwd_small8_kernel<K, false>, lds); if (!e) hipLaunchKernelGGL((attn_fwd_small8_kernel<K, false>), dim3(B * H), dim3(512), lds, stream, a); } break;
    switch (nkt) { FWD_SMALL8(5) FWD_SMALL8(6) FWD_SMALL8(7) FWD_SMALL8(8) }
#undef FWD_SMALL8
    if (e) return e;
  } else if (nkt <= 8) {
    const int lds = 2 * npad * 128 + 4 * 4096 + ((VITAMD_GDBG & 0x4000) ? 32768 : 0);      // (dbg bit 14, experimental builds: occupancy probe - one workgroup per CU)
    int e = VITAMD_OK;
#define FWD_SMALL(K) case K: e = drop ? launch_fwd_small<K, true>(a, lds, stream) : launch_fwd_small<K, false>(a, lds, stream); break;
    switch (nkt) { FWD_SMALL(1) FWD_SMALL(2) FWD_SMALL(3) FWD_SMALL(4) FWD_SMALL(5) FWD_SMALL(6) FWD_SMALL(7) FWD_SMALL(8) }
#undef FWD_SMALL
    if (e) return e;
  } else {
    const int lds = 2 * npad * 128;
    if (drop) {
      if (int e = set_lds(attn_fwd_kernel<true>, lds)) return e;
      hipLaunchKernelGGL(attn_fwd_kernel<true>, dim3(B * H), dim3(256), lds, stream, a);
    } else {
      if (int e = set_lds(attn_fwd_kernel<false>, lds)) return e;
      hipLaunchKernelGGL(attn_fwd_kernel<false>, dim3(B * H), dim3(256), lds, stream, a);
    }
  }
  return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
}

extern "C" int vitamd_attention_bwd(const void* qkv, const void* o, const float* lse2, const void* d_o, void* dqkv, float* delta,
                                    float* dbias, int B, int N, int H, int head_dim, int causal, float dropout_p,
                                    unsigned long long seed, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (head_dim != DH) return VITAMD_ERR_SHAPE;
  AttnArgs a{(const __bf16*)qkv, (__bf16*)o, (float*)lse2, (const __bf16*)d_o, (__bf16*)dqkv, delta, dbias, B, N, H, causal,
             0.125f * 1.4426950408889634f, 0.125f, 0u, 1.0f, 0u, 0u, nullptr, nullptr, VITAMD_GDBG};
  if (int e = check(a)) return e;
  if (!qkv || !o || !lse2 || !d_o || !dqkv || !delta || !attn_dropout(a, dropout_p, seed)) return VITAMD_ERR_ARG;
  if (N > MAX_N) {
    const int ldsq = 2 * CH * 128 + 4 * 4096, ldsk = 2 * CH * 128 + 2 * CH * 4 + 4 * 4096;
    const dim3 grid(B * H, ((N + 31) / 32 + 3) / 4);
    if (a.drop_thresh) {
      if (int e = set_lds(attn_bwd_dq_long_kernel<true>, ldsq)) return e;
      if (int e = set_lds(attn_bwd_dkv_long_kernel<true>, ldsk)) return e;
      hipLaunchKernelGGL(attn_bwd_dq_long_kernel<true>, grid, dim3(256), ldsq, stream, a);   // also writes delta
      hipLaunchKernelGGL(attn_bwd_dkv_long_kernel<true>, grid, dim3(256), ldsk, stream, a);
    } else {
      if (int e = set_lds(attn_bwd_dq_long_kernel<false>, ldsq)) return e;
      if (int e = set_lds(attn_bwd_dkv_long_kernel<false>, ldsk)) return e;
      hipLaunchKernelGGL(attn_bwd_dq_long_kernel<false>, grid, dim3(256), ldsq, stream, a);
      hipLaunchKernelGGL(attn_bwd_dkv_long_kernel<false>, grid, dim3(256), ldsk, stream, a);
    }
    return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
  }
  const int npad = (N + 31) / 32 * 32;
  const int lds1 = 2 * npad * 128 + 4 * 4096, lds2 = 2 * npad * 128 + 2 * npad * 4 + 4 * 4096;
  const dim3 grid(B * H), block(256);
  const int nkt = npad / 32;
  if (!a.drop_thresh && !a.causal && nkt >= 2 && nkt <= 7 && !(VITAMD_GDBG & 0x20000)) {      // (round 4, tools/ab_attn_bwd_pipe.py: with 8 / 9 key tiles - 256 tokens: ViT-VQGAN, 288: TiTok - the fully unrolled kernels LOSE to the plain loops, 604 against 402 us at B 256, N 256, H 12 and 597 against 440 at N 288; 2-7 tiles: equal to 13 % faster)     // the ViT shapes: pipelined forms (dbg bit 17 of experimental builds: the plain loops)
    int e = VITAMD_OK;
#ifdef VITAMD_EXPERIMENTAL
    if (nkt <= 7 && (g_vitamd_debug2 & 16)) {            // one staging of the head, split roles (attn_bwd_split_kernel)
      const int lds3 = 4 * npad * 128 + 2 * npad * 4 + 8 * 4096;
#define SPLIT(K) case K: e = set_lds(attn_bwd_split_kernel<K>, lds3); if (!e) hipLaunchKernelGGL(attn_bwd_split_kernel<K>, grid, dim3(512), lds3, stream, a); break;
      switch (nkt) { SPLIT(2) SPLIT(3) SPLIT(4) SPLIT(5) SPLIT(6) SPLIT(7) }
#undef SPLIT
      if (e) return e;
      return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
    }
#endif
#define DQ_PIPE(K) case K: e = set_lds(attn_bwd_dq_pipe_kernel<K>, lds1); if (!e) hipLaunchKernelGGL(attn_bwd_dq_pipe_kernel<K>, grid, block, lds1, stream, a); break;
    switch (nkt) { DQ_PIPE(2) DQ_PIPE(3) DQ_PIPE(4) DQ_PIPE(5) DQ_PIPE(6) DQ_PIPE(7) }
#undef DQ_PIPE
    if (e) return e;
#define DKV_PIPE(K) case K: e = set_lds(attn_bwd_dkv_pipe_kernel<K>, lds2); if (!e) hipLaunchKernelGGL(attn_bwd_dkv_pipe_kernel<K>, grid, block, lds2, stream, a); break;
    switch (nkt) { DKV_PIPE(2) DKV_PIPE(3) DKV_PIPE(4) DKV_PIPE(5) DKV_PIPE(6) DKV_PIPE(7) }
#undef DKV_PIPE
    if (e) return e;
    return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
  }
#define BWD_SMALL(DROP, CAUSAL)                                                                      \
  do {                                                                                               \
    if (int e = set_lds((attn_bwd_dq_kernel<DROP, CAUSAL>), lds1)) return e;                         \
    if (int e = set_lds((attn_bwd_dkv_kernel<DROP, CAUSAL>), lds2)) return e;                        \
    hipLaunchKernelGGL((attn_bwd_dq_kernel<DROP, CAUSAL>), grid, block, lds1, stream, a); /* also writes delta */ \
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<DROP, CAUSAL>), grid, block, lds2, stream, a);          \
  } while (0)
  if (a.drop_thresh) { if (a.causal) BWD_SMALL(true, true); else BWD_SMALL(true, false); }
  else { if (a.causal) BWD_SMALL(false, true); else BWD_SMALL(false, false); }
#undef BWD_SMALL
  return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
}

#ifdef VITAMD_EXPERIMENTAL
// experimental library only: sum (and clear) the per-wave phase-probe rows into out[16] = two kernel slots x (6 phases, waves, -)
extern "C" int vitamd_debug_attn_probe(unsigned long long* out) {
  static unsigned long long host[2 * PROBE_WAVES * 8];
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_attn_probe), sizeof(host)) != hipSuccess) return VITAMD_ERR_LAUNCH;
  for (int k = 0; k < 2; ++k)
    for (int j = 0; j < 8; ++j) {
      unsigned long long t = 0;
      for (int w = 0; w < PROBE_WAVES; ++w) t += host[((size_t)k * PROBE_WAVES + w) * 8 + j];
      out[k * 8 + j] = t;
    }
  void* dptr = nullptr;
  if (hipGetSymbolAddress(&dptr, HIP_SYMBOL(g_attn_probe)) != hipSuccess) return VITAMD_ERR_LAUNCH;
  return hipMemset(dptr, 0, sizeof(host)) == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
}
#endif
