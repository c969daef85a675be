// out[P,Q] (fp32) += sum_r L[r,p] * Rm[r,q]   — the weight-gradient GEMM of every Linear on the path:
//   dW[out,in] = dY[M,out]^T . X[M,in]   (reduction over the M = batch*tokens rows; the backward of
//   reference transformer.py:21,37,39 and of the patch-embed conv train_vit.py:34).
//
// Both operands are stored with the REDUCTION index as the slow (row) dimension, so neither is
// MFMA-fragment shaped in memory.  gfx950 answer: stage whole 512-B rows of [r][256 cols] tiles by LDS-DMA (zero-fill past the last
// row through the buffer descriptor's range check) and read the fragments with the hardware transpose read ds_read_b64_tr_b16.
// The 16-B chunk index of row r is XOR-ed with (r&3)<<2 (on the global source side) which makes every transposed read
// bank-conflict-free (tools/lds_banks.py).  mfma_f32_32x32x16_bf16 so one accumulator register of a wave = two 128-B row segments:
// the shape float atomics run at full rate with.  Split over the reduction dimension (grid = tiles x splits): partial tiles go to a
// workspace with plain stores and a reduce pass sums them (bitwise reproducible), or fp32 atomics straight into `out`.
// The production kernel is the ping-pong kernel below; the round-1 kernels are in experimental/gemm_tn_variants.inc.
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

// ---------------------------------------------------------------------------------------------
// Ping-pong form of the same GEMM (the production kernel since round 2).  PMC on the kernel above: MFMA pipe 28 % busy, waves
// parked on s_waitcnt / s_barrier 57 % of their cycles (profiles/r02/a_baseline_pmc_mfma.json) - its one vmcnt(0) + barrier
// per 64-row stage drains the load queue every microsecond.  Here nothing ever drains:
//   * the reduction rows are the slow dimension of BOTH operands, so a 64-row stage splits into four 16-row QUARTERS of whole
//     512-B rows (16 KiB: 8 KiB of L, 8 KiB of Rm) and one 32x32x16 k-step consumes exactly one quarter: a quarter's LDS slot is
//     free again as soon as its k-step is read, long before the rest of the stage;
//   * LDS is a ring of NQ quarter slots filled by LDS-DMA (2 one-KiB pieces per wave per quarter) D quarters ahead of the reads,
//     behind a COUNTED s_waitcnt vmcnt(2(D-1)) - never 0 inside the loop;
//   * one PHASE per quarter: [transposed reads of the quarter's fragments | 2 DMA pieces | counted wait] s_barrier
//     [8 MFMAs = 256 matrix-pipe cycles] s_barrier, and the waves of the second wave row (wave >= 4: the second wave of every
//     SIMD) run ONE barrier behind the first: while one wave of a SIMD issues its MFMAs its partner reads LDS and issues DMA.
// Ordering (cdna_hip_programming.md section 5, "Read a staged buffer one phase AFTER the wait that retires it"): quarter q is
// read in phase q by both groups; every wave retires ITS pieces of quarter q by the counted wait of phase q-1, which precedes
// a barrier both groups pass before any phase-q read (RAW); the slot of quarter q is refilled with quarter q+NQ in phase
// q+NQ-D >= q+2, two barriers after the later group's reads of it have been waited for (WAR)  =>  D <= NQ-2.
// Past-the-end quarters are still "loaded" (range-checked to zeros, no traffic) so the vmcnt arithmetic is uniform.
constexpr int QSLOT = 16 * 512 * 2;   // bytes of one quarter slot

template <bool WS, int NQ, int D>
__global__ __launch_bounds__(NW * 64) void gemm_tn_pp_kernel(const GemmTnArgs a, int tiles_p, int tiles_q, int splits) {
  static_assert(D >= 2 && D <= NQ - 2, "prefetch distance: WAR rule");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave / WQ, wq = wave % WQ;

  const int ntile = tiles_p * tiles_q;
  const int id = xcd_remap(blockIdx.x, ntile * splits);
  const int split = id / ntile, tile = id % ntile;
  const int p0 = (tile / tiles_q) * BP, q0 = (tile % tiles_q) * BQ;
  const int nsteps = (a.R + BR - 1) / BR;
  const int s_lo = (int)((long)nsteps * split / splits), s_hi = (int)((long)nsteps * (split + 1) / splits);
  if (s_lo >= s_hi) return;
  const int g_lo = 4 * s_lo, g_hi = 4 * s_hi;      // quarters (16 reduction rows each)

  const bool isL = wave < NW / 2;
  const int ld = isL ? a.ldl : a.ldr;
  const int c0 = isL ? p0 : q0;
  const int ncols = isL ? a.P : a.Q;
  const srd_t srd = isL ? make_srd(a.L, (size_t)a.R * a.ldl * 2) : make_srd(a.Rm, (size_t)a.R * a.ldr * 2);
  // this wave's two pieces of a quarter: rows 4*(wave&3) + 2*i + (lane>>5) of its operand, chunk = lane&31 (swizzled)
  unsigned voff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = ((wave & 3) * 2 + i) * 2 + (lane >> 5);
    const int logical = (lane & 31) ^ ((row & 3) << 2);
    const int col = c0 + logical * 8;
    voff[i] = (col < ncols) ? (unsigned)(((size_t)row * ld + col) * 2) : 0x80000000u;
  }
  const unsigned qbytes = (unsigned)16 * ld * 2;             // global bytes per quarter
  const unsigned dst0 = lds_addr(smem) + (isL ? 0 : 8192) + (wave & 3) * 2048;  // this wave's pieces inside slot 0

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

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
    offB[j] = 8192 + rowpart + ((chunk ^ (qq << 2)) << 4);
  }

  // prologue: quarters g_lo .. g_lo+D-1
  int slot_w = 0;                                   // slot the next issued quarter goes to
  unsigned soff = (unsigned)g_lo * qbytes;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const unsigned dst = dst0 + slot_w * QSLOT;
    asm_glds16(srd, dst, voff[0], soff);
    asm_glds16(srd, dst + 1024, voff[1], soff);
    soff += qbytes;
    slot_w = slot_w + 1 == NQ ? 0 : slot_w + 1;
  }
  static_assert(D <= 8, "add a vmcnt literal");
#define VITAMD_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
  VITAMD_WAIT_VM(2 * (D - 1));                      // quarter g_lo landed (this wave's pieces)
  __builtin_amdgcn_s_barrier();                     // ... and everyone's
  asm volatile("" ::: "memory");
  if (wp == 1) __builtin_amdgcn_s_barrier();        // second wave row runs one barrier behind from here on
  int slot_r = 0;
  for (int g = g_lo; g < g_hi; ++g) {
    // ---- read section
    const char* q = smem + slot_r * QSLOT;
    bf16x8 af[MT], bfr[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bfr[j] = tr_frag(q + offB[j]);
#pragma unroll
    for (int i = 0; i < MT; ++i) af[i] = tr_frag(q + offA[i]);
    {
      const unsigned dst = dst0 + slot_w * QSLOT;
      asm_glds16(srd, dst, voff[0], soff);
      asm_glds16(srd, dst + 1024, voff[1], soff);
      soff += qbytes;
      slot_w = slot_w + 1 == NQ ? 0 : slot_w + 1;
    }
    VITAMD_WAIT_VM(2 * (D - 1));                    // my pieces of quarter g+1 have landed
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // ---- matrix section
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    slot_r = slot_r + 1 == NQ ? 0 : slot_r + 1;
  }
  if (wp == 0) __builtin_amdgcn_s_barrier();        // balance the stagger
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the past-the-end pieces (zeros) must not outlive the workgroup's LDS
#undef VITAMD_WAIT_VM

  if constexpr (WS) {
    float* wt = a.ws + ((size_t)split * ntile + tile) * (BP * BQ);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int ql = wq * (BQ / WQ) + j * 32 + (lane & 31);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int pl = wp * (BP / WP) + i * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int r = 0; r < 16; ++r) wt[(pl + (r & 3) + 8 * (r >> 2)) * BQ + ql] = acc[i][j][r];
      }
    }
  } else {
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
}

// ---------------------------------------------------------------------------------------------
// Loader-wave form of gemm_tn_pp_kernel (round 3; GemmTnArgs::form = 1): the same ring, phases and MFMA order (bit-identical results), but the
// LDS-DMA requests move out of the eight compute waves into FOUR loader waves (waves 8-11, one per SIMD).  A compute wave's read section is then
// the 12 transposed reads alone; in gemm_tn_pp_kernel it also issues 2 DMA pieces whose issue costs 60-185 cycles each beside the partner wave's
// 256 cycles of MFMAs.  Loader l issues the four pieces the compute waves l (L operand) and l + 4 (R operand) would have issued, waits for them
// with the same counted vmcnt arithmetic (4 pieces per quarter instead of 2) and takes part in every barrier on the first wave row's timeline.
// 12 waves x 168 registers = 3 x 168 per SIMD lane: fits the 512-register file only because this kernel needs <= 168 - and fills it: nothing
// can share the CU with such a workgroup, where the 8-wave form leaves 176 registers per lane for e.g. LayerNorm waves of the other stream.
// Measured (tools/bench_tn.py, profiles/r03/tn_loader_waves.log): alone 171 / 214 / 216 us against 201 / 247 / 258 for the three ViT-B weight
// gradients (-15 %); inside the step it wins where the launch runs beside kernels that fill their CUs anyway (the fc2 weight gradient, beside
// the two input-gradient GEMMs: -0.1 ... -0.36 ms per step) and loses where the 8-wave form shared CUs with LayerNorm (all launches: +0.46 ms).
// LSPLIT (round 4, from the NT loader kernel where it was worth 5-14 %): the loaders issue half of a quarter's four requests behind the phase's
// first barrier instead of all four in front of it, and run at priority 3 - a loader that is late for a barrier holds up all twelve waves.
template <int NQ, int D, int ABL = 0, bool LSPLIT = true>      // ABL (experimental builds, timing only, results are garbage): 1 = no MFMAs, 2 = no transposed LDS reads, 3 = neither
__global__ __launch_bounds__(768) void gemm_tn_ld_kernel(const GemmTnArgs a, int tiles_p, int tiles_q, int splits) {
  static_assert(D >= 2 && D <= NQ - 2, "prefetch distance: WAR rule");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave >= NW;
  const int ntile = tiles_p * tiles_q;
  const int id = xcd_remap(blockIdx.x, ntile * splits);
  const int split = id / ntile, tile = id % ntile;
  const int p0 = (tile / tiles_q) * BP, q0 = (tile % tiles_q) * BQ;
  const int nsteps = (a.R + BR - 1) / BR;
  const int s_lo = (int)((long)nsteps * split / splits), s_hi = (int)((long)nsteps * (split + 1) / splits);
  if (s_lo >= s_hi) return;
  const int g_lo = 4 * s_lo, g_hi = 4 * s_hi;
#define VITAMD_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
  if (loader) {
    const int l = wave - NW;                         // pieces of compute-wave identities (wave & 3) == l, both operands
    const srd_t srdL = make_srd(a.L, (size_t)a.R * a.ldl * 2), srdR = make_srd(a.Rm, (size_t)a.R * a.ldr * 2);
    unsigned voffL[2], voffR[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (l * 2 + i) * 2 + (lane >> 5);
      const int logical = (lane & 31) ^ ((row & 3) << 2);
      voffL[i] = (p0 + logical * 8 < a.P) ? (unsigned)(((size_t)row * a.ldl + p0 + logical * 8) * 2) : 0x80000000u;
      voffR[i] = (q0 + logical * 8 < a.Q) ? (unsigned)(((size_t)row * a.ldr + q0 + logical * 8) * 2) : 0x80000000u;
    }
    const unsigned qbL = (unsigned)16 * a.ldl * 2, qbR = (unsigned)16 * a.ldr * 2;
    const unsigned dstL = lds_addr(smem) + l * 2048, dstR = dstL + 8192;
    unsigned soL = (unsigned)g_lo * qbL, soR = (unsigned)g_lo * qbR;
    int slot_w = 0;
    auto issue_l = [&]() {
      const unsigned o = slot_w * QSLOT;
      asm_glds16(srdL, dstL + o, voffL[0], soL);
      asm_glds16(srdL, dstL + o + 1024, voffL[1], soL);
    };
    auto issue_r = [&]() {
      const unsigned o = slot_w * QSLOT;
      asm_glds16(srdR, dstR + o, voffR[0], soR);
      asm_glds16(srdR, dstR + o + 1024, voffR[1], soR);
      soL += qbL; soR += qbR;
      slot_w = slot_w + 1 == NQ ? 0 : slot_w + 1;
    };
    auto issue = [&]() { issue_l(); issue_r(); };
    if constexpr (LSPLIT) __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int d = 0; d < D; ++d) issue();
    VITAMD_WAIT_VM(4 * (D - 1));
    __builtin_amdgcn_s_barrier();
    for (int g = g_lo; g < g_hi; ++g) {
      if constexpr (LSPLIT) {
        issue_l();
        __builtin_amdgcn_s_barrier();
        issue_r();
      } else {
      issue();
      __builtin_amdgcn_s_barrier();
      }
      // quarter g + 1 is first read by the first wave row BEHIND the second barrier of this iteration, so the wait may sit here rather than in
      // front of the first barrier (one more interval for the requests to land).  Measured equal (164 / 212 / 215 us either way).
      VITAMD_WAIT_VM(4 * (D - 1));
      __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_s_barrier();                   // the first wave row's balancing barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  const int wp = wave / WQ, wq = wave % WQ;
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
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
    offB[j] = 8192 + rowpart + ((chunk ^ (qq << 2)) << 4);
  }
  __builtin_amdgcn_s_barrier();                     // quarter g_lo has landed (the loaders waited for it)
  asm volatile("" ::: "memory");
  if (wp == 1) __builtin_amdgcn_s_barrier();
  int slot_r = 0;
  for (int g = g_lo; g < g_hi; ++g) {
    const char* q = smem + slot_r * QSLOT;
    bf16x8 af[MT], bfr[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) { if (ABL & 2) { for (int e = 0; e < 8; ++e) bfr[j][e] = (__bf16)1.0f; asm volatile("" : "+v"(bfr[j])); } else bfr[j] = tr_frag(q + offB[j]); }
#pragma unroll
    for (int i = 0; i < MT; ++i) { if (ABL & 2) { for (int e = 0; e < 8; ++e) af[i][e] = (__bf16)1.0f; asm volatile("" : "+v"(af[i])); } else af[i] = tr_frag(q + offA[i]); }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) { if (ABL & 1) { acc[i][j][0] += (float)af[i][0] * (float)bfr[j][0]; } else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0); }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    slot_r = slot_r + 1 == NQ ? 0 : slot_r + 1;
  }
  if (wp == 0) __builtin_amdgcn_s_barrier();
#undef VITAMD_WAIT_VM
  float* wt = a.ws + ((size_t)split * ntile + tile) * (BP * BQ);
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int ql = wq * (BQ / WQ) + j * 32 + (lane & 31);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int pl = wp * (BP / WP) + i * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int r = 0; r < 16; ++r) wt[(pl + (r & 3) + 8 * (r >> 2)) * BQ + ql] = acc[i][j][r];
    }
  }
}

// out[p][q] (+)= sum_s ws[s][tile][p_local][q_local]; RPT float4 per thread
constexpr int RPT = 1;      // 4 is faster back to back (9.2 vs ~12 us) but slower inside the step (13.9 vs 12.0 us)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, int P, int Q, int ldo,
                                                            int tiles_q, int ntile, int splits, int accumulate) {
  const int tile = blockIdx.y;
  const int p0 = (tile / tiles_q) * BP, q0 = (tile % tiles_q) * BQ;
  f32x4 sum[RPT];
  const float* src[RPT];
  bool ok[RPT];
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    const int idx = (blockIdx.x * RPT + k) * 256 + threadIdx.x;   // float4 index inside the tile: 64 per row
    const int pl = idx >> 6, ql = (idx & 63) * 4;
    ok[k] = p0 + pl < P && q0 + ql < Q;
    src[k] = ws + (size_t)tile * (BP * BQ) + pl * BQ + ql;
    sum[k] = ok[k] ? *(const f32x4*)src[k] : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  for (int s2 = 1; s2 < splits; ++s2) {
#pragma unroll
    for (int k = 0; k < RPT; ++k)
      if (ok[k]) sum[k] += *(const f32x4*)(src[k] + (size_t)s2 * ntile * (BP * BQ));
  }
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    if (!ok[k]) continue;
    const int idx = (blockIdx.x * RPT + k) * 256 + threadIdx.x;
    const int p = p0 + (idx >> 6), q = q0 + (idx & 63) * 4;
    float* dst = out + (size_t)p * ldo + q;
    if (q + 3 < Q && (ldo & 3) == 0) {
      if (accumulate) sum[k] += *(const f32x4*)dst;
      *(f32x4*)dst = sum[k];
    } else {
      for (int c = 0; c < 4 && q + c < Q; ++c) dst[c] = accumulate ? dst[c] + sum[k][c] : sum[k][c];
    }
  }
}

}  // namespace

static int auto_splits(int R, int P, int Q, int requested) {
  const int ntile = ((P + BP - 1) / BP) * ((Q + BQ - 1) / BQ);
  const int nsteps = (R + BR - 1) / BR;
  int splits = requested;
  if (splits <= 0) splits = ntile >= 256 ? 1 : 256 / ntile;
  if (splits > nsteps) splits = nsteps;
  return splits;
}

extern "C" long vitamd_gemm_tn_ws_bytes(int R, int P, int Q, int splits) {
  if (R <= 0 || P <= 0 || Q <= 0) return 0;
  const long ntile = (long)((P + BP - 1) / BP) * ((Q + BQ - 1) / BQ);
  return (long)auto_splits(R, P, Q, splits) * ntile * BP * BQ * (long)sizeof(float);
}

int vitamd_gemm_tn_impl(const GemmTnArgs& a, hipStream_t stream) {
  if (a.R <= 0 || a.P <= 0 || a.Q <= 0 || a.ldl % 8 || a.ldr % 8 || a.ldl < a.P || a.ldr < a.Q || a.ldo < a.Q) return VITAMD_ERR_SHAPE;
  if ((size_t)(a.R + BR) * a.ldl * 2 >= 0x80000000ull || (size_t)(a.R + BR) * a.ldr * 2 >= 0x80000000ull) return VITAMD_ERR_SHAPE;
  if (!a.L || !a.Rm || !a.out) return VITAMD_ERR_ARG;
  const int tiles_p = (a.P + BP - 1) / BP, tiles_q = (a.Q + BQ - 1) / BQ;
  const int ntile = tiles_p * tiles_q;
  const int splits = auto_splits(a.R, a.P, a.Q, a.splits);
  constexpr int lds = 8 * QSLOT;                                  // the ring of eight quarter slots (= two 64-row stages)
  const bool use_ws = a.ws != nullptr && a.ws_bytes >= (size_t)splits * ntile * BP * BQ * sizeof(float);
  if (!use_ws && !a.accumulate) return VITAMD_ERR_ARG;      // overwrite mode needs the workspace: the atomic form can only add to `out`
  const dim3 grid(ntile * splits), block(NW * 64);
  if (use_ws) {
    bool loader = a.form == 1;
#ifdef VITAMD_EXPERIMENTAL
    // A/B: bit 23 = the loader-wave form for every launch; bits 20 / 21 / 22 = where P <= 768 (fc2), P = 2304 (QKV), P = 3072 (fc1); bit 19 = never
    loader = !(g_vitamd_debug & 0x80000) && (loader || (g_vitamd_debug & 0x800000) || ((g_vitamd_debug & 0x100000) && a.P <= 768) ||
                                             ((g_vitamd_debug & 0x200000) && a.P == 2304) || ((g_vitamd_debug & 0x400000) && a.P == 3072));
#endif
    if (loader) {
#ifdef VITAMD_EXPERIMENTAL
      const int dv = (g_vitamd_debug >> 16) & 3;     // bits 16-17: timing-only ablations of the loader form (1 no MFMAs, 2 no transposed reads, 3 neither; results garbage)
      if (dv) {
        auto kern = dv == 1 ? gemm_tn_ld_kernel<8, 4, 1> : dv == 2 ? gemm_tn_ld_kernel<8, 4, 2> : gemm_tn_ld_kernel<8, 4, 3>;
        if (int e = set_lds(kern, lds)) return e;
        hipLaunchKernelGGL(kern, grid, dim3(768), lds, stream, a, tiles_p, tiles_q, splits);
      } else
#endif
      {
#ifdef VITAMD_EXPERIMENTAL
      if (g_vitamd_debug2 & 64) {                    // A/B (vitamd_set_debug2 bit 6): the round-3 loaders (all four requests in front of the first barrier, priority 0)
        if (int e = set_lds((gemm_tn_ld_kernel<8, 4, 0, false>), lds)) return e;
        hipLaunchKernelGGL((gemm_tn_ld_kernel<8, 4, 0, false>), grid, dim3(768), lds, stream, a, tiles_p, tiles_q, splits);
      } else
#endif
      {
      if (int e = set_lds(gemm_tn_ld_kernel<8, 4>, lds)) return e;
      hipLaunchKernelGGL((gemm_tn_ld_kernel<8, 4>), grid, dim3(768), lds, stream, a, tiles_p, tiles_q, splits);
      }
      }
    } else {
    if (int e = set_lds(gemm_tn_pp_kernel<true, 8, 4>, lds)) return e;
    hipLaunchKernelGGL((gemm_tn_pp_kernel<true, 8, 4>), grid, block, lds, stream, a, tiles_p, tiles_q, splits);
    }
#ifdef VITAMD_EXPERIMENTAL
    if (!(g_vitamd_debug2 & 128))                    // (vitamd_set_debug2 bit 7, timing only, gradients garbage: NO reduce pass - the bound on what folding / batching it could gain)
#endif
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(BP * BQ / 4 / 256 / RPT, ntile), dim3(256), 0, stream, a.ws, a.out, a.P, a.Q, a.ldo, tiles_q, ntile, splits,
                       a.accumulate);
  } else {
    if (int e = set_lds(gemm_tn_pp_kernel<false, 8, 4>, lds)) return e;
    hipLaunchKernelGGL((gemm_tn_pp_kernel<false, 8, 4>), grid, block, lds, stream, a, tiles_p, tiles_q, splits);
  }
  return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
}
