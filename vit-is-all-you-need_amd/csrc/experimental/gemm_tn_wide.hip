// Weight-gradient GEMM, wide tile:  ws[split][P][Q] = sum over the split's rows r of L[r,p] * Rm[r,q]
// (same contraction as gemm_tn.hip; reference transformer.py:21,37,39 backward).
//
// A MEASURED ALTERNATIVE, OFF BY DEFAULT (vitamd_set_debug bit 25 turns it on; tests/test_gpu_kernels.py runs it).
// Ablations of gemm_tn_kernel (tools/ablate_tn.py) show the 256x256 form is bound by operand delivery, not by the
// matrix pipes (removing every MFMA leaves its time unchanged).  A 256 x 384 output tile needs 1/256 + 1/384
// bytes per MAC instead of 2/256 - 17 % less traffic, and 22 % fewer LDS read bytes per MAC - so it should have
// been faster.  It is not: whole-step A/B (tools/ab_dbg.py) +0.2 ms with BR = 32, +0.5 ms with BR = 64, with
// or without row-aligned DMA pieces.  What it costs: 192 accumulator registers per wave leave no room for the
// VGPR staging path (worth 0.6 ms on this GEMM) nor for double-buffered fragments; 24 tiles x 10 splits fill
// 240 of 256 CUs.  The four-wave / 512-register form (wave tile 128 x 192 = 384 accumulator registers) does not
// compile usefully: hipcc puts every MFMA accumulator in AGPRs (256) and shuttles the rest through
// v_accvgpr moves and scratch (496 moves + 77 scratch accesses per 48 MFMAs).
//
// Operand tiles are staged row-major by LDS-DMA ([64 r][256] for L; R cut in [64 r][256] + [64 r][128] so that
// no 1-KiB piece crosses a row; 16-B chunk c of row r stored at c ^ ((r&3)<<2), closed inside aligned groups of
// 16 chunks; all row strides are multiples of the 256-B bank period, so the transposed reads stay conflict-free
// as in gemm_tn.hip), fragments come from ds_read_b64_tr_b16.  8 waves (2 x 4), wave tile 128 x 96.
#include "../common.h"
#include "../vitamd_internal.h"

namespace {

constexpr int W_BR = 64;                       // reduction rows per stage
constexpr int W_BP = 256, W_BQ = 384;
constexpr int W_NW = 8;                        // waves: 2 (p) x 4 (q)
constexpr int W_MT = 4, W_NT = 3;              // 32x32 tiles per wave: 128 x 96
// staged images per stage: L [BR][256], R cut in two so that no 1-KiB DMA piece crosses a row: RA [BR][256] + RB [BR][128]
constexpr int ROWB_L = 512, ROWB_RA = 512, ROWB_RB = 256;
constexpr int L_BYTES = W_BR * ROWB_L, RA_BYTES = W_BR * ROWB_RA, RB_BYTES = W_BR * ROWB_RB;   // 32 + 32 + 16 KiB
constexpr int STAGE_BYTES = L_BYTES + RA_BYTES + RB_BYTES;   // 80 KiB
constexpr int L_PIECES = L_BYTES / 1024 / W_NW;              // 4 one-KiB pieces per wave (2 rows each)
constexpr int RA_PIECES = RA_BYTES / 1024 / W_NW;            // 4 (2 rows each)
constexpr int RB_PIECES = RB_BYTES / 1024 / W_NW;            // 2 (4 rows each)
constexpr int W_PPW = L_PIECES + RA_PIECES + RB_PIECES;      // 10

typedef LDS_AS bf16x4* lds_bf16x4_ptr;

__device__ __forceinline__ bf16x8 tr_frag_w(const char* p, int rowb) {
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p));
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p + 4 * rowb));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__global__ __launch_bounds__(W_NW * 64) void gemm_tn_wide_kernel(const GemmTnArgs a, int tiles_q, int ntile, int splits) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave >> 2, wq = wave & 3;

  const int id = xcd_remap(blockIdx.x, ntile * splits);
  const int split = id / ntile, tile = id % ntile;
  const int p0 = (tile / tiles_q) * W_BP, q0 = (tile % tiles_q) * W_BQ;
  const int nsteps = (a.R + W_BR - 1) / W_BR;
  const int s_lo = (int)((long)nsteps * split / splits), s_hi = (int)((long)nsteps * (split + 1) / splits);

  const __amdgpu_buffer_rsrc_t rsrcL = make_rsrc(a.L, (size_t)a.R * a.ldl * 2);
  const __amdgpu_buffer_rsrc_t rsrcR = make_rsrc(a.Rm, (size_t)a.R * a.ldr * 2);

  // piece i of this wave: L pieces, then RA pieces (both: rows 2q, 2q+1, 32 lanes x 16 B per row), then RB pieces
  // (rows 4q..4q+3, 16 lanes x 16 B per row); q = (index inside the kind) * 8 + wave
  unsigned voff[W_PPW];
#pragma unroll
  for (int i = 0; i < W_PPW; ++i) {
    if (i < L_PIECES + RA_PIECES) {
      const bool isL = i < L_PIECES;
      const int row = ((isL ? i : i - L_PIECES) * W_NW + wave) * 2 + (lane >> 5);
      const int logical = (lane & 31) ^ ((row & 3) << 2);
      const int col = (isL ? p0 : q0) + logical * 8;
      voff[i] = (col < (isL ? a.P : a.Q)) ? (unsigned)(((size_t)row * (isL ? a.ldl : a.ldr) + col) * 2) : 0x80000000u;
    } else {
      const int row = ((i - L_PIECES - RA_PIECES) * W_NW + wave) * 4 + (lane >> 4);
      const int logical = (lane & 15) ^ ((row & 3) << 2);
      const int col = q0 + 256 + logical * 8;
      voff[i] = (col < a.Q) ? (unsigned)(((size_t)row * a.ldr + col) * 2) : 0x80000000u;
    }
  }
  const unsigned stepL = (unsigned)W_BR * a.ldl * 2, stepR = (unsigned)W_BR * a.ldr * 2;
#define PIECE_W(i_, s_, sb_)                                                                                     \
  do {                                                                                                           \
    if ((i_) < L_PIECES) buf_glds16(rsrcL, (sb_) + ((i_) * W_NW + wave) * 1024, voff[i_], (s_) * stepL);          \
    else buf_glds16(rsrcR, (sb_) + L_BYTES + (((i_) - L_PIECES) * W_NW + wave) * 1024, voff[i_], (s_) * stepR);    \
  } while (0)
  // (RA and RB images are contiguous after L and their pieces are numbered consecutively, so one formula serves both)

  f32x16 acc[W_MT][W_NT];
#pragma unroll
  for (int i = 0; i < W_MT; ++i)
#pragma unroll
    for (int j = 0; j < W_NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // transposed-read lane addressing (as gemm_tn.hip): lane 4*qq+pp of a 16-lane group supplies row qq, columns
  // 4pp..4pp+3 of a 4x16 block; group = (k-half h = lane>>5, column half = (lane>>4)&1)
  const int h = lane >> 5, colhalf = (lane >> 4) & 1, qq = (lane >> 2) & 3, pp = lane & 3;
  int offA[W_MT], offB[W_NT], rowbB[W_NT];
#pragma unroll
  for (int i = 0; i < W_MT; ++i) {
    const int chunk = (wp * 128 + i * 32) / 8 + 2 * colhalf + (pp >> 1);
    offA[i] = (8 * h + qq) * ROWB_L + (pp & 1) * 8 + ((chunk ^ (qq << 2)) << 4);
  }
#pragma unroll
  for (int j = 0; j < W_NT; ++j) {
    const int c0 = wq * 96 + j * 32;                     // first column of this 32-wide q-tile inside the 384
    const bool inB = c0 >= 256;                          // wave-uniform
    rowbB[j] = inB ? ROWB_RB : ROWB_RA;
    const int chunk = (inB ? c0 - 256 : c0) / 8 + 2 * colhalf + (pp >> 1);
    offB[j] = L_BYTES + (inB ? RA_BYTES : 0) + (8 * h + qq) * rowbB[j] + (pp & 1) * 8 + ((chunk ^ (qq << 2)) << 4);
  }

  if (s_lo < s_hi) {
#pragma unroll
    for (int i = 0; i < W_PPW; ++i) PIECE_W(i, s_lo, smem);
  }
  for (int s = s_lo; s < s_hi; ++s) {
    const int cur = (s - s_lo) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // stage s visible to all four waves; buffer cur^1 free
    asm volatile("" ::: "memory");
    const char* buf = smem + cur * STAGE_BYTES;
    const bool more = s + 1 < s_hi;
    char* nb = smem + (cur ^ 1) * STAGE_BYTES;
    bf16x8 af[W_MT], bfr[W_NT];
#pragma unroll
    for (int ks = 0; ks < W_BR / 16; ++ks) {
      if (more) {          // this group's share of the next stage's LDS-DMA: pieces ks, ks + G, ks + 2G, ...
#pragma unroll
        for (int i = ks; i < W_PPW; i += W_BR / 16) {
          PIECE_W(i, s + 1, nb);
        }
      }
#pragma unroll
      for (int j = 0; j < W_NT; ++j) bfr[j] = tr_frag_w(buf + offB[j] + ks * 16 * rowbB[j], rowbB[j]);
#pragma unroll
      for (int i = 0; i < W_MT; ++i) af[i] = tr_frag_w(buf + offA[i] + ks * 16 * ROWB_L, ROWB_L);
#pragma unroll
      for (int i = 0; i < W_MT; ++i)
#pragma unroll
        for (int j = 0; j < W_NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  }
#undef PIECE_W

  // D[p][q]: col q = lane&31, row p = (reg&3) + 8*(reg>>2) + 4*(lane>>5).  Plain stores into this split's [P][Q] plane
  // (a wave-instruction writes two 128-B row segments); the reduce pass sums the planes.
  float* plane = a.ws + (size_t)split * a.P * a.Q;
#pragma unroll
  for (int j = 0; j < W_NT; ++j) {
    const int q = q0 + wq * 96 + j * 32 + (lane & 31);
#pragma unroll
    for (int i = 0; i < W_MT; ++i) {
      const int pb = p0 + wp * 128 + i * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int p = pb + (r & 3) + 8 * (r >> 2);
        if (p < a.P && q < a.Q) plane[(size_t)p * a.Q + q] = acc[i][j][r];
      }
    }
  }
}

// out[p][q] (+)= sum_s ws[s][p][q]; one float4 per thread (Q % 4 == 0)
__global__ __launch_bounds__(256) void splitk_reduce_planes_kernel(const float* __restrict__ ws, float* __restrict__ out, int P, int Q,
                                                                   int ldo, int splits, int accumulate) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t n4 = (size_t)P * Q / 4;
  if (idx >= n4) return;
  const size_t e = idx * 4;
  const int p = (int)(e / Q), q = (int)(e - (size_t)p * Q);
  f32x4 sum = *(const f32x4*)(ws + e);
  for (int s = 1; s < splits; ++s) sum += *(const f32x4*)(ws + (size_t)s * P * Q + e);
  float* dst = out + (size_t)p * ldo + q;
  if ((ldo & 3) == 0) {
    if (accumulate) sum += *(const f32x4*)dst;
    *(f32x4*)dst = sum;
  } else {
    for (int c = 0; c < 4; ++c) dst[c] = accumulate ? dst[c] + sum[c] : sum[c];
  }
}

}  // namespace

bool vitamd_gemm_tn_wide_ok(int R, int P, int Q, int requested_splits) {
  if (!(g_vitamd_debug & (1 << 25))) return false;         // opt-in (A/B knob): the 256x256 kernel is faster, see the header
  return requested_splits <= 0 && P % W_BP == 0 && Q % W_BQ == 0 && R >= 4096 && (P / W_BP) * (Q / W_BQ) <= 128;
}

int vitamd_gemm_tn_wide_splits(int R, int P, int Q) {
  const int ntile = (P / W_BP) * (Q / W_BQ);
  const int nsteps = (R + W_BR - 1) / W_BR;
  int splits = 256 / ntile;
  if (splits < 1) splits = 1;
  if (splits > nsteps) splits = nsteps;
  return splits;
}

int vitamd_gemm_tn_wide_launch(const GemmTnArgs& a, hipStream_t stream) {
  const int tiles_q = a.Q / W_BQ, ntile = (a.P / W_BP) * tiles_q;
  const int splits = vitamd_gemm_tn_wide_splits(a.R, a.P, a.Q);
  if (!a.ws || a.ws_bytes < (size_t)splits * a.P * a.Q * sizeof(float)) return VITAMD_ERR_ARG;
  constexpr int lds = 2 * STAGE_BYTES;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)gemm_tn_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return VITAMD_ERR_LAUNCH;
    attr_done = true;
  }
  hipLaunchKernelGGL(gemm_tn_wide_kernel, dim3(ntile * splits), dim3(W_NW * 64), lds, stream, a, tiles_q, ntile, splits);
  const size_t n4 = (size_t)a.P * a.Q / 4;
  hipLaunchKernelGGL(splitk_reduce_planes_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, a.ws, a.out, a.P, a.Q, a.ldo, splits,
                     a.accumulate);
  return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
}
