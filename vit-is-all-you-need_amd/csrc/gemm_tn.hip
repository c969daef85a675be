// out[P,Q] (fp32) += sum_r L[r,p] * Rm[r,q]   — the weight-gradient GEMM of every Linear on the path:
//   dW[out,in] = dY[M,out]^T . X[M,in]   (reduction over the M = batch*tokens rows; the backward of
//   reference transformer.py:21,37,39 and of the patch-embed conv train_vit.py:34).
//
// Both operands are stored with the REDUCTION index as the slow (row) dimension, so neither is
// MFMA-fragment shaped in memory.  gfx950 answer: stage [64 r][256 cols] tiles row-major by LDS-DMA
// (buffer_load ... lds, whole 512-B rows, zero-fill past the last row through the buffer descriptor's
// range check) and read the fragments with the hardware transpose read ds_read_b64_tr_b16.  The 16-B
// chunk index of row r is XOR-ed with (r&3)<<2 (on the global source side) which makes every
// transposed read bank-conflict-free (tools/lds_banks.py).  mfma_f32_32x32x16_bf16 so one accumulator
// register of a wave = two 128-B row segments: the shape float atomics run at full rate with.
// Split over the reduction dimension (grid = tiles x splits) with fp32 atomic accumulation.
#include "common.h"
#include "vitamd_internal.h"

namespace {

constexpr int BR = 64;    // reduction rows per stage
constexpr int BP = 256, BQ = 256;
constexpr int WP = 2, WQ = 4, NW = 8;
constexpr int MT = BP / WP / 32;  // 4 p-tiles per wave
constexpr int NT = BQ / WQ / 32;  // 2 q-tiles per wave
constexpr int TILE_BYTES = BR * 512;        // one operand tile
constexpr int BUF_BYTES = 2 * TILE_BYTES;   // L + R
constexpr int PPW = (2 * BR / 2) / NW;      // 1-KiB pieces (2 rows) per wave per stage = 8

typedef LDS_AS bf16x4* lds_bf16x4_ptr;

__device__ __forceinline__ bf16x8 tr_frag(const char* p) {
  // two transposed 4x16 reads: k = 0..3 and k = 4..7 of this lane's half (rows +4 = +2048 B)
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p));
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p + 4 * 512));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__global__ __launch_bounds__(NW * 64) void gemm_tn_kernel(const GemmTnArgs a, int tiles_p, int tiles_q, int splits) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave / WQ, wq = wave % WQ;

  const int ntile = tiles_p * tiles_q;
  const int id = xcd_remap(blockIdx.x, ntile * splits);
  const int split = id / ntile, tile = id % ntile;
  const int p0 = (tile / tiles_q) * BP, q0 = (tile % tiles_q) * BQ;

  // reduction range of this split, in BR-row steps
  const int nsteps = (a.R + BR - 1) / BR;
  const int s_lo = (int)((long)nsteps * split / splits), s_hi = (int)((long)nsteps * (split + 1) / splits);
  if (s_lo >= s_hi) return;

  const auto rsrcL = __builtin_amdgcn_make_buffer_rsrc((void*)a.L, 0, (int)((size_t)a.R * a.ldl * 2), 0x00020000);
  const auto rsrcR = __builtin_amdgcn_make_buffer_rsrc((void*)a.Rm, 0, (int)((size_t)a.R * a.ldr * 2), 0x00020000);

  // this wave's pieces: waves 0-3 stage L rows, waves 4-7 stage R rows (16 rows each per stage)
  const bool isL = wave < NW / 2;
  const int ld = isL ? a.ldl : a.ldr;
  const int c0 = isL ? p0 : q0;
  const int ncols = isL ? a.P : a.Q;
  unsigned voff[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int row = ((wave & 3) * PPW + i) * 2 + (lane >> 5);
    const int logical = (lane & 31) ^ ((row & 3) << 2);
    int col = c0 + logical * 8;
    // columns past the matrix edge: point far out of range so the DMA writes zeros
    voff[i] = (col < ncols) ? (unsigned)(((size_t)(s_lo * BR + row) * ld + col) * 2) : 0xfffffff0u;
  }
  const unsigned step_bytes = (unsigned)BR * ld * 2;

  auto stage = [&](int buf) {
    char* base = smem + buf * BUF_BYTES + (isL ? 0 : TILE_BYTES) + (wave & 3) * PPW * 1024;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      if (isL) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcL, (LDS_AS void*)(base + i * 1024), 16, voff[i], 0, 0, 0);
      else     __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcR, (LDS_AS void*)(base + i * 1024), 16, voff[i], 0, 0, 0);
      if (voff[i] < 0xf0000000u) voff[i] += step_bytes;
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // transposed-read lane addressing (see header): within a 16-lane group lane 4*qq+pp supplies
  // row qq, columns 4pp..4pp+3 of a 4x16 block; group = (k-half h = lane>>5, column half = (lane>>4)&1)
  const int h = lane >> 5, colhalf = (lane >> 4) & 1, qq = (lane >> 2) & 3, pp = lane & 3;
  const int rowpart = (8 * h + qq) * 512 + (pp & 1) * 8;
  int offA[MT], offB[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int chunk = (wp * (BP / WP) + i * 32) / 8 + 2 * colhalf + (pp >> 1);
    offA[i] = rowpart + ((chunk ^ (qq << 2)) << 4);
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int chunk = (wq * (BQ / WQ) + j * 32) / 8 + 2 * colhalf + (pp >> 1);
    offB[j] = TILE_BYTES + rowpart + ((chunk ^ (qq << 2)) << 4);
  }

  stage(0);
  for (int s = s_lo; s < s_hi; ++s) {
    const int cur = (s - s_lo) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (s + 1 < s_hi) stage(cur ^ 1);
    const char* buf = smem + cur * BUF_BYTES;
#pragma unroll
    for (int ks = 0; ks < BR / 16; ++ks) {
      bf16x8 af[MT], bfr[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) bfr[j] = tr_frag(buf + offB[j] + ks * 16 * 512);
#pragma unroll
      for (int i = 0; i < MT; ++i) af[i] = tr_frag(buf + offA[i] + ks * 16 * 512);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  }

  // D[i = p][j = q]: col q = lane&31, row p = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int q = q0 + wq * (BQ / WQ) + j * 32 + (lane & 31);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int pbase = p0 + wp * (BP / WP) + i * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int p = pbase + (r & 3) + 8 * (r >> 2);
        if (p < a.P && q < a.Q) atomicAdd(a.out + (size_t)p * a.ldo + q, acc[i][j][r]);
      }
    }
  }
}

}  // namespace

int vitamd_gemm_tn_impl(const GemmTnArgs& a, hipStream_t stream) {
  if (a.R <= 0 || a.P <= 0 || a.Q <= 0 || a.ldl % 8 || a.ldr % 8 || a.ldl < a.P || a.ldr < a.Q || a.ldo < a.Q) return VITAMD_ERR_SHAPE;
  if ((size_t)a.R * a.ldl * 2 >= 0xf0000000ull || (size_t)a.R * a.ldr * 2 >= 0xf0000000ull) return VITAMD_ERR_SHAPE;
  if (!a.L || !a.Rm || !a.out) return VITAMD_ERR_ARG;
  const int tiles_p = (a.P + BP - 1) / BP, tiles_q = (a.Q + BQ - 1) / BQ;
  const int ntile = tiles_p * tiles_q;
  const int nsteps = (a.R + BR - 1) / BR;
  int splits = a.splits;
  if (splits <= 0) splits = ntile >= 256 ? 1 : 256 / ntile;
  if (splits > nsteps) splits = nsteps;
  static bool attr_done = false;
  constexpr int lds = 2 * BUF_BYTES;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return VITAMD_ERR_LAUNCH;
    attr_done = true;
  }
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(ntile * splits), dim3(NW * 64), lds, stream, a, tiles_p, tiles_q, splits);
  return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
}
