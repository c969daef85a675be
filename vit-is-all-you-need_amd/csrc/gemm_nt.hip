// C[M,N] = A[M,K] . B[N,K]^T  (both operands K-contiguous, bf16, fp32 accumulate) with fused
// epilogues.  This is the one kernel behind every forward Linear and every dgrad of the hot path:
//   forward  y = x W^T + b      : A = x [M,K],  B = W  [N,K]           (reference transformer.py:21,37,39)
//   dgrad    dx = dy W          : A = dy [M,N'], B = W^T [K',N'] (the host keeps a bf16 transposed copy)
//
// Kernels in the production library:
//   gemm_nt_pp_kernel   (32 MT) x 256 x 64 ping-pong kernel, MT = 8 (256 rows) or 10 (320 rows): LDS-DMA that never drains, one wave of
//                       every SIMD in its matrix section while its partner reads LDS and issues DMA.  Every large GEMM of the step.
//   gemm_nt_kernel      128 x 128 x 64 plain double-buffered kernel for small problems (classifier head, tiny models).
// Epilogues (gemm_nt_epilogue.h): gemm_epilogue_rows (LDS-transposed, row-major 16-B accesses), gemm_epilogue (direct; small tiles, fp32).
// The round-1 kernels (pipe / persistent / ring / deep) are measured alternatives in experimental/gemm_nt_variants.inc, built only with
// `make EXPERIMENTAL=1` (libvitamd_exp.so, for the A/B tools); DESIGN.md section 4 holds their numbers.
#include <type_traits>
#include <mutex>
#include <atomic>
#include <cmath>
#include <cstring>
#include "gemm_nt_epilogue.h"

namespace {

template <int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(WM * WN * 64) void gemm_nt_kernel(const GemmNtArgs p) {
  constexpr int NW = WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN;   // wave tile
  constexpr int MT = WTM / 16, NT = WTN / 16;   // 16x16 accumulator tiles per wave
  constexpr int PIECES = (BM + BN) / 8;         // 1-KiB LDS-DMA pieces (8 rows x 128 B) per K-tile
  constexpr int PPW = PIECES / NW;
  static_assert(PIECES % NW == 0, "pieces must divide over waves");
  constexpr int BUF_BYTES = (BM + BN) * 128;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (tile / tiles_n) * BM;
  const int n0 = (tile % tiles_n) * BN;

  const __bf16* __restrict__ A = (const __bf16*)p.A;
  const __bf16* __restrict__ B = (const __bf16*)p.B;
  const int K = p.K;

  // per-lane source pointers of this wave's pieces (row clamped: out-of-range rows re-read the
  // last valid row; their results are never stored)
  const __bf16* src[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int piece = wave * PPW + i;
    const int row = piece * 8 + (lane >> 3);
    const int logical = (lane & 7) ^ (row & 7);
    if (piece < BM / 8) {
      const int g = min(m0 + row, p.M - 1);
      src[i] = A + (size_t)g * K + logical * 8;
    } else {
      const int g = min(n0 + row - BM, p.N - 1);
      src[i] = B + (size_t)g * K + logical * 8;
    }
  }

  auto stage = [&](int kt, int buf) {
    char* base = smem + buf * BUF_BYTES + wave * PPW * 1024;
#pragma unroll
    for (int i = 0; i < PPW; ++i) glds16(src[i] + kt * BK, base + i * 1024);
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment read offsets inside a buffer (ks = 1 flips chunk bit 2 -> byte ^ 64)
  const int frag_off = (lane & 15) * 128 + ((((lane >> 4) ^ (lane & 7)) & 7) << 4);
  const int a_off = wm * WTM * 128 + frag_off;
  const int b_off = BM * 128 + wn * WTN * 128 + frag_off;

  const int nkt = K / BK;
  stage(0, 0);
  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // tile kt landed for everyone; everyone finished reading buffer cur^1
    if (kt + 1 < nkt) stage(kt + 1, cur ^ 1);
    const char* buf = smem + cur * BUF_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[MT], bfr[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) bfr[j] = *(const bf16x8*)(buf + ((b_off + j * 16 * 128) ^ (ks * 64)));
#pragma unroll
      for (int i = 0; i < MT; ++i) af[i] = *(const bf16x8*)(buf + ((a_off + i * 16 * 128) ^ (ks * 64)));
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
  }
  if constexpr (BM == 256 && BN == 256 && WM == 2 && WN == 4 && EPI != EPI_F32) {
    if (p.N % 8 == 0 && p.ldo % 8 == 0) {
      gemm_epilogue_rows<EPI>(p, acc, m0, n0, wm, wn, lane, tid, wave, smem);
      return;
    }
  }
  gemm_epilogue<BN, WM, WN, WTM, WTN, MT, NT, EPI>(p, acc, m0, n0, wm, wn, lane, tid, smem);
}

// ---------------------------------------------------------------------------------------------
// Ping-pong kernel (round 2): (32 MT) x 256 x 64 tile (MT = 8: 256 rows, MT = 10: 320 rows), 8 waves (2 x 4, wave tile 16 MT x 64),
// two K-tile buffers, LDS-DMA that never drains.  PMC on the pipe kernel above (profiles/r02/a_baseline_pmc_mfma.json): MFMA
// pipe 32-50 % busy, waves parked on s_waitcnt / s_barrier 35-42 % of their cycles - its vmcnt(0) + barrier per K-tile empties
// the load queue every microsecond.  Here:
//   * a K-tile is cut into NP = MT/2 A-PARTS (the 32 rows [32 j, 32 j + 32) of BOTH wave rows: 64 rows x 128 B = 8 KiB, one
//     1-KiB DMA piece per wave) and the B block (256 rows, 4 pieces per wave).  PHASE j of a K-tile multiplies A-part j with the
//     whole B block (2 x 4 tiles x 2 k-substeps = 16 MFMAs); B's fragments are read once, in phase 0, and stay in registers.  So
//     every LDS region is read in ONE known phase and is free long before the K-tile ends.
//   * every phase requests one A piece LA phases ahead of its use and (most phases) one B piece LB phases ahead, into the region
//     the same part of two K-tiles earlier left.  Waits are COUNTED (never 0 in the loop): the count per phase is computed at
//     compile time from the request schedule (PpSchedule below).
//   * phase = [ds_reads | DMA requests | counted wait] s_barrier [16 MFMAs, s_setprio 1] s_barrier; the second wave row (waves 4-7,
//     the second wave of every SIMD) runs ONE barrier behind the first, so on every SIMD one wave feeds the matrix pipe while its
//     partner reads LDS and issues DMA (cdna_hip_programming.md section 5, the 8-phase template).
// Ordering: a region first read in phase n is retired by every wave (its own pieces) in phase n-1, before a barrier that both
// groups pass ahead of any phase-n read (RAW); a region is refilled >= 2 phases after its only read (WAR)  =>  LA, LB <= 2 NP - 2.
// Requests for K-tiles that do not exist (before the first, past the last) are issued out of range (zero fill, no traffic) so the
// counts are the same in every phase.  The DMA is issued from inline asm (common.h::asm_glds16): hipcc would otherwise put
// vmcnt(0) in front of every ds_read.
template <int NP, int LA, int LB>
struct PpSchedule {
  // phase p of K-tile t issues: A-part (p + LA) % NP of K-tile t + (p + LA) / NP ; and B piece q = (p + LB) % NP (if q < 4) of K-tile
  // t + (p + LB - q) / NP.  Program order inside a phase: A request, then B request.
  static constexpr int a_part(int p) { return (p + LA) % NP; }
  static constexpr int a_tile(int p) { return (p + LA) / NP; }
  static constexpr int b_piece(int p) { return (p + LB) % NP < 4 ? (p + LB) % NP : -1; }
  static constexpr int b_tile(int p) { return (p + LB - (p + LB) % NP) / NP; }
  // outstanding requests allowed after phase p's requests so that everything first read in phase p+1 has landed
  static constexpr int wait(int p) {
    int allowed = 0;
    for (int d = 0; d < 4 * NP; ++d) {          // walk back over the phases p, p-1, ... (program order reversed: B then A)
      const int ph = ((p - d) % NP + NP) % NP;
      if (b_piece(ph) >= 0) {
        if (d + 1 >= LB - b_piece(ph)) return allowed;      // needed in phase (p-d) + LB - q <= p+1
        ++allowed;
      }
      if (d + 1 >= LA) return allowed;                         // A request of phase p-d is needed in phase p-d+LA <= p+1
      ++allowed;
    }
    return allowed;
  }
  static constexpr int lookback = (LA > LB ? LA : LB);          // phases before the first whose requests the prologue replays
};

template <int EPI, int MT, int LA, int LB, bool PERS = false>     // PERS: one workgroup per CU walks a strided list of tiles (the automatic choice for large problems)
__global__ __launch_bounds__(512) void gemm_nt_pp_kernel(const GemmNtArgs p) {
  constexpr int NP = MT / 2;
  static_assert(MT % 2 == 0 && LA >= 2 && LA <= 2 * NP - 2 && LB >= 5 && LB <= 2 * NP - 2, "request leads");
  using S = PpSchedule<NP, LA, LB>;
  constexpr int BM = 32 * MT, BN = 256, WN = 4, NT = 4;
  constexpr int PART = 64 * 128;                  // bytes of an A-part
  constexpr int BUFB = NP * PART + 256 * 128;     // one K-tile buffer: A-parts, then the B block

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
  const int ntiles = tiles_m * tiles_n;
  const int K = p.K;
  const int nkt = K / 64;
  const srd_t srdA = make_srd(p.A, (size_t)p.M * K * 2);
  const srd_t srdB = make_srd(p.B, (size_t)p.N * K * 2);
  if constexpr (PERS) {     // start-up stagger of the persistent form: dbg bits 20-23 = groups, bits 8-15 = delay per group in ~us
    const int P = (VITAMD_DBG(p) >> 20) & 0xf, unit = (VITAMD_DBG(p) >> 8) & 0xff;
    if (P > 1)
      for (int i = 0; i < unit * (int)(((VITAMD_DBG(p) & (1 << 24)) ? (blockIdx.x & 7) : (blockIdx.x >> 3)) % P); ++i) __builtin_amdgcn_s_sleep(32);   // bit 24: whole XCDs share a group
  }
  for (int ti = blockIdx.x; ti < ntiles; ti += PERS ? (int)gridDim.x : ntiles) {
  const int tile = xcd_remap(ti, ntiles);
  int tm, tn;
  tile_coords(tile, tiles_m, tiles_n, tiles_n >= 6, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  // this wave's piece of A-part j: LDS rows 8*wave + (lane>>3) of the part = rows 32 j + (lr&31) of wave row lr>>5;
  // its piece q of the B block: rows 64 q + 8*wave + (lane>>3).  16-B chunk lane&7, XOR (row&7) on the source side.
  unsigned voffA[NP], voffB[4];
  {
    const int lr = 8 * wave + (lane >> 3);
    const unsigned chunk = (unsigned)(((lane & 7) ^ (lr & 7)) * 16);
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int ga = min(((VITAMD_DBG(p) & 0x40000) ? (m0 & 0x3ff) : m0) + (lr >> 5) * (16 * MT) + j * 32 + (lr & 31), p.M - 1);     // clamp: rows past M are never stored (dbg bit 18, timing only: every tile loads one of a few L2-resident panels)
      voffA[j] = (unsigned)ga * (unsigned)(K * 2) + chunk;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int gb = min(((VITAMD_DBG(p) & 0x40000) ? 0 : n0) + 64 * q + lr, p.N - 1);
      voffB[q] = (unsigned)gb * (unsigned)(K * 2) + chunk;
    }
  }
  const unsigned lds0 = lds_addr(smem) + wave * 1024;
  constexpr unsigned OOB = 0x80000000u;
  // (experimental builds, timing only, results garbage: dbg bit 28 = every request out of range - the instruction is issued, nothing is fetched;
  //  dbg bit 24 = no request instructions at all: what the main loop costs without its feed)
  auto request_a = [&](int kt, int j) {
    const bool live = kt >= 0 && kt < nkt && !(VITAMD_DBG(p) & 0x10000000);
    if (VITAMD_DBG(p) & 0x1000000) return;
    asm_glds16(srdA, lds0 + (kt & 1) * BUFB + j * PART, live ? voffA[j] : OOB, live ? (unsigned)kt * 128u : 0u);
  };
  auto request_b = [&](int kt, int q) {
    const bool live = kt >= 0 && kt < nkt && !(VITAMD_DBG(p) & 0x10000000);
    if (VITAMD_DBG(p) & 0x1000000) return;
    asm_glds16(srdB, lds0 + (kt & 1) * BUFB + NP * PART + q * 8192, live ? voffB[q] : OOB, live ? (unsigned)kt * 128u : 0u);
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment reads: 16-row tile at LDS row rb: lane -> row rb + (lane&15), chunk ((lane>>4) + 4 ks) ^ (row&7); k-substep 1 flips
  // chunk bit 2 = XOR 64 on the swizzled offset, hence one base pointer per substep
  const int frag_off = (lane & 15) * 128 + ((((lane >> 4) ^ (lane & 7)) & 7) << 4);
  const char* const rdA[2] = {smem + wm * 32 * 128 + frag_off, smem + wm * 32 * 128 + (frag_off ^ 64)};                    // + part*PART + i*2048 (+ buffer)
  const char* const rdB[2] = {smem + NP * PART + wn * 64 * 128 + frag_off, smem + NP * PART + wn * 64 * 128 + (frag_off ^ 64)};   // + j*2048 (+ buffer)

#define VITAMD_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
  // prologue: replay the requests of the S::lookback phases before phase 0 (those for K-tiles < 0 go out of range: the queue then
  // looks exactly as in steady state and the same counted waits apply from the first phase on)
#pragma unroll
  for (int P = -S::lookback; P < 0; ++P) {
    const int ph = ((P % NP) + NP) % NP, t = (P - ph) / NP;      // P = NP * t + ph, t < 0
    request_a(t + S::a_tile(ph), S::a_part(ph));
    if (S::b_piece(ph) >= 0) request_b(t + S::b_tile(ph), S::b_piece(ph));
  }
  VITAMD_WAIT_VM(S::wait(NP - 1));
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (wm == 1) __builtin_amdgcn_s_barrier();        // second wave row: one barrier behind from here on

  bf16x8 bq[NT][2], af[2][2];
  auto ktile = [&](int kt, auto bufc) {
    constexpr int BUF = decltype(bufc)::value;
#pragma unroll
    for (int ph = 0; ph < NP; ++ph) {
      // ---- read section
      if (ph == 0) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) bq[j][ks] = *(const bf16x8*)(rdB[ks] + BUF * BUFB + j * 2048);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) af[i][ks] = *(const bf16x8*)(rdA[ks] + BUF * BUFB + ph * PART + i * 2048);
      request_a(kt + S::a_tile(ph), S::a_part(ph));
      if (S::b_piece(ph) >= 0) request_b(kt + S::b_tile(ph), S::b_piece(ph));
      VITAMD_WAIT_VM(S::wait(ph));
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      // ---- matrix section: A-part ph x the whole B block
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[2 * ph + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[j][ks], af[i][ks], acc[2 * ph + i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  };
  int kt = 0;
  for (; kt + 1 < nkt; kt += 2) {
    ktile(kt, std::integral_constant<int, 0>{});
    ktile(kt + 1, std::integral_constant<int, 1>{});
  }
  if (kt < nkt) ktile(kt, std::integral_constant<int, 0>{});
  if (wm == 0) __builtin_amdgcn_s_barrier();        // balance the stagger
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // past-the-end requests (zeros) land before the epilogue reuses LDS
#undef VITAMD_WAIT_VM
  if constexpr (EPI == EPI_F32) gemm_epilogue<BN, 2, WN, 16 * MT, 64, MT, NT, EPI>(p, acc, m0, n0, wm, wn, lane, tid, smem);
  else if (p.N % 8 == 0 && p.ldo % 8 == 0) gemm_epilogue_rows<EPI, MT>(p, acc, m0, n0, wm, wn, lane, tid, wave, smem);
  else gemm_epilogue<BN, 2, WN, 16 * MT, 64, MT, NT, EPI>(p, acc, m0, n0, wm, wn, lane, tid, smem);
  if constexpr (PERS) __syncthreads();              // every wave is done with its epilogue image before the next tile's DMA lands
  }
}

static int device_cus() {          // CU count of the current device (persistent launches: one workgroup per CU)
  static int cus[16] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
  if (!cus[dev]) {
    int n = 0;
    cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
  }
  return cus[dev];
}

template <int EPI, int MT, int LA, int LB, bool PERS = false>
int launch_pp(const GemmNtArgs& p, hipStream_t stream) {
  constexpr int BM = 32 * MT;
  constexpr int ops_b = 2 * ((MT / 2) * 8192 + 32768), epi_b = 8 * MT * 2048;     // operand buffers / epilogue images
  constexpr int lds = ops_b > epi_b ? ops_b : epi_b;
  auto kern = gemm_nt_pp_kernel<EPI, MT, LA, LB, PERS>;
  if (int e = set_lds(kern, lds)) return e;
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + 255) / 256);
  const int cus = PERS ? device_cus() - 16 * ((VITAMD_DBG(p) >> 25) & 7) : tiles;     // (dbg bits 25-27 of experimental builds: 16 k fewer persistent workgroups than CUs)
  hipLaunchKernelGGL(kern, dim3(PERS && tiles > cus ? cus : tiles), dim3(512), lds, stream, p);
  return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
}

template <int BM, int BN, int WM, int WN, int EPI>
int launch(const GemmNtArgs& p, hipStream_t stream) {
  constexpr int lds = 2 * (BM + BN) * 128;
  auto kern = gemm_nt_kernel<BM, BN, WM, WN, EPI>;
  if (int e = set_lds(kern, lds)) return e;
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(WM * WN * 64), lds, stream, p);
  return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
}

// 320-row tiles instead of 256-row ones when that does not need more (rounds of 256 CUs) x (rows per tile): at M = 50 432,
// N = 768 outputs take 591 tiles = 3 rounds of 256 rows but 474 tiles = 2 rounds of 320 rows (-0.6 ms/step); N = 3072 is a tie
// (10 x 256 = 8 x 320), N = 2304 stays on 256 (7 x 256 < 6 x 320; forcing 320 there measured equal).  A 384-row tile would need
// 192 accumulator registers (compiles to 256 VGPRs) and a two-pass epilogue, and quantises worse at this M.
// dbg bit 19 disables the tall tile (A/B knob).
static bool prefer_tall(const GemmNtArgs& p) {
  if ((VITAMD_DBG(p) & 0x80000) || p.N % 8 != 0 || p.ldo % 8 != 0 || p.K % 64 != 0) return false;
  if (p.epi != EPI_BIAS_BF16 && p.epi != EPI_RESID_F32 && p.epi != EPI_GELU && p.epi != EPI_DGELU) return false;
  const long tn = (p.N + 255) / 256;
  const long r256 = (((p.M + 255) / 256) * tn + 255) / 256, r320 = (((p.M + 319) / 320) * tn + 255) / 256;
  return r320 * 320 <= r256 * 256;     // ties go to the tall tile: 142 instead of 128 FLOP per staged byte (whole-step A/B: -0.5 ms on the N = 3072 GEMMs alone)
}

#include "gemm_nt_seam.h"
#include "gemm_nt_ld.h"

// bf16 nearest-even of a double, decided on exact distances (no float intermediate rounding)
static unsigned short bf16_rne_d(double v) {
  auto val = [](unsigned short b) { const unsigned w = (unsigned)b << 16; float g; memcpy(&g, &w, 4); return (double)g; };
  const float f = (float)v;
  unsigned u;
  memcpy(&u, &f, 4);
  unsigned short best = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
  double bd = fabs(val(best) - v);
  const unsigned short first = best;
  for (int dl = -1; dl <= 1; dl += 2) {
    const unsigned short n = (unsigned short)(first + dl);
    const double dd = fabs(val(n) - v);
    if (dd < bd || (dd == bd && !(n & 1) && (best & 1))) { best = n; bd = dd; }
  }
  return best;
}

// Device image of the erf-GELU table every GELU epilogue reads (gemm_nt_epilogue.h::gelu_lookup): entry sign x 2048 + (|bits| - 0x3900) for
// bf16 inputs 2^-13 <= |x| < 8 holds bf16(x Phi(x)) | bf16(Phi(x) + x phi(x)) << 16, from double (erfc for the tail).  One 16-KiB allocation per
// device, made by vitamd_init - the ONLY place this library allocates or synchronises; the launch functions read the pointer and nothing else.
static std::atomic<const unsigned*> g_gelu_tab[64];

static const unsigned* gelu_table() {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return nullptr; }
  return g_gelu_tab[dev].load(std::memory_order_acquire);
}

}  // namespace

int vitamd_init_impl(int device, hipStream_t stream) {
  static std::mutex mu;
  static unsigned host[4096];
  static bool built = false;
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); return VITAMD_ERR_LAUNCH; }
  if (device < 0) device = cur;
  if (device >= 64) return VITAMD_ERR_ARG;
  std::lock_guard<std::mutex> lock(mu);
  if (g_gelu_tab[device].load(std::memory_order_acquire)) return VITAMD_OK;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return VITAMD_ERR_ARG; }   // never inside a capture
  if (!built) {
    for (int sg = 0; sg < 2; ++sg)
      for (int i = 0; i < 2048; ++i) {
        const unsigned w = (unsigned)(0x3900 + i) << 16;
        float ax;
        memcpy(&ax, &w, 4);
        const double x = sg ? -(double)ax : (double)ax;
        const double cdf = 0.5 * erfc(-x * 0.70710678118654752440);
        const double pdf = 0.39894228040143267794 * exp(-0.5 * x * x);
        host[sg * 2048 + i] = (unsigned)bf16_rne_d(x * cdf) | ((unsigned)bf16_rne_d(cdf + x * pdf) << 16);
      }
    built = true;
  }
  if (device != cur && hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return VITAMD_ERR_ARG; }
  void* d = nullptr;
  bool ok = hipMalloc(&d, sizeof(host)) == hipSuccess;
  ok = ok && hipMemcpy(d, host, sizeof(host), hipMemcpyHostToDevice) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();                // a failed HIP call must not surface as a later launch's error
    if (d) (void)hipFree(d);
  }
  if (device != cur) (void)hipSetDevice(cur);
  if (!ok) return VITAMD_ERR_LAUNCH;
  g_gelu_tab[device].store((const unsigned*)d, std::memory_order_release);
  return VITAMD_OK;
}

namespace {

#ifdef VITAMD_EXPERIMENTAL
#include "experimental/gemm_nt_ld10.inc"
// tile codes 24 / 25 / 30 (experimental builds): the seam kernel (gemm_nt_seam.h) on 256- / 320-row tiles whatever the automatic rule says
template <int EPI>
int dispatch_seam_explicit(const GemmNtArgs& p, hipStream_t stream, int tile) {
  if constexpr (EPI == EPI_BIAS_BF16 || EPI == EPI_GELU || EPI == EPI_DGELU) {
    if (!seam_ok(p)) return VITAMD_ERR_SHAPE;
    if (tile == 24 || tile == 30) return launch_seam<EPI, 8, EPI == EPI_GELU>(p, stream, device_cus());      // (GELU: always the table form)
    if constexpr (EPI == EPI_BIAS_BF16) {
      if (tile == 25) return launch_seam<EPI, 10>(p, stream, device_cus());
    }
  }
  return VITAMD_ERR_ARG;
}
#endif

// ---- which kernel a launch takes (one place: the dispatcher below executes the plan, vitamd_gemm_nt_plan reports it) -------------------------
// tile selector of the C ABI: 0 = auto, 128 = the 128x128 kernel, 256 / 320 = the ping-pong kernel on 256- / 320-row tiles, one workgroup per
// tile; 512 = auto without persistent launches; 1024 = auto with persistent launches but without the seam form; 2048 = the loader-wave form.
enum NtForm { NT_FORM_SMALL = 1, NT_FORM_PP = 2, NT_FORM_PP_PERSISTENT = 3, NT_FORM_SEAM = 4, NT_FORM_LOADER = 5 };
struct NtPlan { int form, rows, err; };

// The loader-wave form (gemm_nt_ld.h) in the automatic choice.  Measured (profiles/r04/ab_nt_loader_whole_step*.log, two boxes,
// bit-identical losses): in place of the 256-row seam kernel on the three K = 768 launch classes of a layer (QKV, fc1 + GELU, dgrad-fc2 x gelu') the
// whole step gains 0.11 / 0.26 ms; on the N = 768 GEMMs (K = 2304 / 3072) it loses - 591 tiles of 256 rows are three rounds of the 256 CUs where
// gemm_nt_pp_kernel runs two rounds of 320-row tiles, and twelve waves at <= 168 registers cannot hold a 320-row accumulator tile - so those stay.
// `short_k`: the launch meets the 256-row seam rule (K <= 1536, >= 3 tiles per CU).
static bool ld_auto(const GemmNtArgs& p, bool short_k) {
  bool on = short_k;
#ifdef VITAMD_EXPERIMENTAL
  // whole-step A/B (tools/ab_ld.py; vitamd_set_debug2): bits 0 / 1 / 2 FLIP the default for plain-bias / GELU / dGELU-multiply launches with a short K
  // loop; bit 3 = also the plain-bias launches with a long K loop (the N = 768 GEMMs), bit 5 = those of them without a bias (the input-gradient GEMMs)
  const int d = g_vitamd_debug2;
  if (short_k) on = on != ((d & (p.epi == EPI_BIAS_BF16 ? 1 : p.epi == EPI_GELU ? 2 : 4)) != 0);
  else if (p.epi == EPI_BIAS_BF16 && p.K > 1536) on = (d & 8) != 0 || ((d & 32) != 0 && !p.bias);
#endif
  return on;
}

static NtPlan plan_single(const GemmNtArgs& p) {
  int tile = p.tile;
  const int epi = p.epi;
  const bool seam_epi = epi == EPI_BIAS_BF16 || epi == EPI_GELU || epi == EPI_DGELU;
  const bool tall_epi = seam_epi || epi == EPI_RESID_F32;
  if (tile == 2048) return (seam_epi && ld_ok(p)) ? NtPlan{NT_FORM_LOADER, 256, VITAMD_OK} : NtPlan{0, 0, VITAMD_ERR_SHAPE};
  const bool no_seam = tile == 1024;      // ABI code 1024: the automatic choice with persistent launches but WITHOUT the seam / loader forms (A/B and start-up probe: ops.seam_probe)
  if (no_seam) tile = 0;
  const long big_tiles = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
  const bool pp_ok = (size_t)p.M * p.K * 2 < 0xf0000000ull && (size_t)p.N * p.K * 2 < 0xf0000000ull && p.K % 64 == 0;
  const bool big = p.N >= 256 && big_tiles >= 192 && pp_ok;
  const bool tall = tall_epi && prefer_tall(p);
  // Automatic choice, more tiles than CUs: the PERSISTENT form (one workgroup per CU walking a strided tile list; same kernel, same
  // results).  Alone it is as fast as one workgroup per tile; inside the training step, next to the weight-gradient GEMMs of the
  // second stream, it is faster: -0.45 / -0.04 / -0.34 / -0.45 ms per step on four boxes (tools/ab_persistent.py).  The explicit
  // tile codes 256 / 320 keep the one-workgroup-per-tile launch.  (dbg bit 5 of experimental builds: no persistent launches)
  if (tile == 512) tile = 0;              // ABI code 512: the automatic choice WITHOUT persistent launches (one workgroup per tile)
  else if (tile == 0 && big && !(VITAMD_DBG(p) & 0x20)) {
    const int cus = device_cus();
    // Short K loops with several tiles per CU: the SEAM form of the persistent kernel (gemm_nt_seam.h: the next tile's pipeline fill is requested
    // before the epilogue, the epilogue runs beside the operand buffers).  Measured on the ViT-B launches (tools/bench_seam.py): QKV 172 -> 163 us,
    // fc1+GELU 316 -> 298 (265-276 with the GELU table, which needs the 256-row ring), dgrad-fc2 298 -> 268 (K = 768, 7-10 tiles per CU); equal or
    // 4 % slower where a CU sees only two tiles of a long K loop (dgrad-fc1 K = 3072, dgrad-QKV K = 2304), which therefore stay on the form
    // below.  Bit-identical results.  (dbg bit 17: off)
    if (seam_epi && !no_seam && seam_ok(p)) {
      if (p.K <= 1536 && !(VITAMD_DBG(p) & 0x20000)) {
        if (epi == EPI_BIAS_BF16 && tall && (long)((p.M + 319) / 320) * ((p.N + 255) / 256) >= 3L * cus) return NtPlan{NT_FORM_SEAM, 320, VITAMD_OK};
        // 256-row tiles: the loader-wave form (gemm_nt_ld.h) where it applies (an even number of K-tiles), else the seam kernel
        if (big_tiles >= 3L * cus) return NtPlan{ld_ok(p) && ld_auto(p, true) ? NT_FORM_LOADER : NT_FORM_SEAM, 256, VITAMD_OK};
      }
      if (ld_ok(p) && ld_auto(p, false)) return NtPlan{NT_FORM_LOADER, 256, VITAMD_OK};
    }
    return NtPlan{NT_FORM_PP_PERSISTENT, tall ? 320 : 256, VITAMD_OK};
  }
  if (tile == 0) tile = big ? (tall ? 320 : 256) : 128;
  if (tile == 320) return (tall_epi && pp_ok) ? NtPlan{NT_FORM_PP, 320, VITAMD_OK} : NtPlan{0, 0, VITAMD_ERR_SHAPE};
  if (tile == 256) return pp_ok ? NtPlan{NT_FORM_PP, 256, VITAMD_OK} : NtPlan{0, 0, VITAMD_ERR_SHAPE};
  if (tile != 128) return NtPlan{0, 0, VITAMD_ERR_ARG};
  return p.K % BK == 0 ? NtPlan{NT_FORM_SMALL, 128, VITAMD_OK} : NtPlan{0, 0, VITAMD_ERR_SHAPE};
}

template <int EPI>
int dispatch_tile(const GemmNtArgs& p, hipStream_t stream) {
  constexpr bool seam_epi = EPI == EPI_BIAS_BF16 || EPI == EPI_GELU || EPI == EPI_DGELU;
  constexpr bool tall_epi = seam_epi || EPI == EPI_RESID_F32;
#ifdef VITAMD_EXPERIMENTAL
  {   // experimental builds: further tile codes force a kernel form whatever the automatic rule says
    int tile = p.tile;
    if (tile >= 24 && tile <= 30) return dispatch_seam_explicit<EPI>(p, stream, tile);
    if (tile == 4096) {                     // the 320-row loader form on ten compute + two loader waves (experimental/gemm_nt_ld10.inc)
      if constexpr (EPI == EPI_BIAS_BF16) return ld10_ok(p) ? launch_ld10(p, stream, device_cus()) : VITAMD_ERR_SHAPE;
      return VITAMD_ERR_SHAPE;
    }
    if (tile == 2049) {                     // the loader-wave form with its first request schedule (burst in phase 0)
      if constexpr (seam_epi) return ld_ok(p) ? launch_ld<EPI, EPI == EPI_GELU, 0>(p, stream, device_cus()) : VITAMD_ERR_SHAPE;
      return VITAMD_ERR_SHAPE;
    }
  }
#endif
  const NtPlan pl = plan_single(p);
  if (pl.err) return pl.err;
  switch (pl.form) {
    case NT_FORM_LOADER:
      if constexpr (seam_epi) return launch_ld<EPI, EPI == EPI_GELU>(p, stream, device_cus());
      break;
    case NT_FORM_SEAM:
      if constexpr (EPI == EPI_BIAS_BF16) { if (pl.rows == 320) return launch_seam<EPI, 10>(p, stream, device_cus()); }
      if constexpr (seam_epi) return launch_seam<EPI, 8, EPI == EPI_GELU>(p, stream, device_cus());
      break;
    case NT_FORM_PP_PERSISTENT:
      if constexpr (tall_epi) { if (pl.rows == 320) return launch_pp<EPI, 10, 4, 6, true>(p, stream); }
      return launch_pp<EPI, 8, 4, 6, true>(p, stream);
    case NT_FORM_PP:
      if constexpr (tall_epi) { if (pl.rows == 320) return launch_pp<EPI, 10, 4, 6>(p, stream); }
      return launch_pp<EPI, 8, 4, 6>(p, stream);
    case NT_FORM_SMALL:
      return launch<128, 128, 2, 2, EPI>(p, stream);
  }
  return VITAMD_ERR_SHAPE;
}

}  // namespace

static int dispatch_epi(const GemmNtArgs& p, hipStream_t stream) {
  switch (p.epi) {
    case EPI_BIAS_BF16: return dispatch_tile<EPI_BIAS_BF16>(p, stream);
    case EPI_GELU: return dispatch_tile<EPI_GELU>(p, stream);
    case EPI_RESID_F32: return dispatch_tile<EPI_RESID_F32>(p, stream);
    case EPI_DGELU: return dispatch_tile<EPI_DGELU>(p, stream);
    case EPI_PATCH_F32: return dispatch_tile<EPI_PATCH_F32>(p, stream);
    case EPI_F32: return dispatch_tile<EPI_F32>(p, stream);
    default: return VITAMD_ERR_ARG;
  }
}

// One big-tile workgroup per CU means a launch runs in whole rounds of 256 tiles; a last round that is mostly empty idles most of
// the chip for a full tile time.  Two remedies: the tile height (prefer_tall) and the tail split decided here.
// Tail split: the last, mostly empty round of big tiles is re-cut into 128x128 tiles.  Only ever paid for the fused-residual fc2 FORWARD GEMM
// on 256-row tiles (591 tiles = 2.31 rounds; -0.2 ms/step), which the 320-row tile has since replaced (474 tiles = 1.85 rounds:
// no split).  Everywhere else it loses on the whole step: +0.9 ms forced on every GEMM (bit 4) because the weight-gradient GEMMs
// of the side stream already fill the backward tails, +0.2 ms on fc1+GELU at 7.4 rounds of 320-row tiles (the 128x128 kernel's
// direct-store GELU epilogue costs more than the 0.6 idle round).  Experimental builds: vitamd_set_debug bit 7 turns it off, bit 4 forces it.
static int tail_split_rows(const GemmNtArgs& p, bool& tall) {       // rows of the head part, 0 = no split
  const int CUS = device_cus();
  tall = (p.tile == 0 || p.tile == 512 || p.tile == 1024) && prefer_tall(p);
  const int bm = tall ? 320 : 256;
  const int tiles_m = (p.M + bm - 1) / bm, tiles_n = (p.N + 255) / 256;
  const long big_tiles = (long)tiles_m * tiles_n;
  const long rem = big_tiles % CUS;
  const bool split_on = (VITAMD_DBG(p) & 16) != 0 || (!(VITAMD_DBG(p) & 128) && p.epi == EPI_RESID_F32 && !tall);
  if (!((p.tile == 0 || p.tile == 512 || p.tile == 1024) && split_on && p.epi != EPI_PATCH_F32 && p.N >= 256 && p.K % 64 == 0 && big_tiles > 2 * CUS && rem != 0 && rem * 10 < CUS * 6))
    return 0;
  const int rows_a = (int)((big_tiles - rem) / tiles_n) * bm;       // M-panels whose tiles fill whole rounds
  return rows_a > 0 && rows_a < p.M ? rows_a : 0;
}

static int check_args(const GemmNtArgs& p) {
  if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.K % 32 != 0 || p.N % 4 != 0 || p.ldo % 4 != 0) return VITAMD_ERR_SHAPE;
  if (!p.A || !p.B || !p.out) return VITAMD_ERR_ARG;
  if (p.epi == EPI_GELU && !p.out2) return VITAMD_ERR_ARG;
  if ((p.epi == EPI_RESID_F32 || p.epi == EPI_DGELU) && !p.aux) return VITAMD_ERR_ARG;
  if (p.epi == EPI_PATCH_F32 && (!p.aux || p.n_patches <= 0)) return VITAMD_ERR_ARG;
  if (p.epi < EPI_BIAS_BF16 || p.epi > EPI_F32) return VITAMD_ERR_ARG;
  return VITAMD_OK;
}

int vitamd_gemm_nt_impl(const GemmNtArgs& p0, hipStream_t stream) {
  if (int e = check_args(p0)) return e;
  GemmNtArgs p = p0;
#ifdef VITAMD_EXPERIMENTAL
  p.dbg2 = g_vitamd_debug2;
#endif
  if (p.epi == EPI_GELU) {                 // ONE rounding of GELU whatever kernel the launch takes: every GELU epilogue reads the table
    p.gelu_tab = gelu_table();
    if (!p.gelu_tab) return VITAMD_ERR_INIT;
  }
  bool tall = false;
  const int rows_a = tail_split_rows(p, tall);
  if (rows_a) {
    GemmNtArgs a = p, b = p;
    a.M = rows_a;
    a.tile = tall ? p.tile : 256;                                   // (auto picks the 320-row form again for the head part)
    const size_t esz_out = (p.epi == EPI_RESID_F32 || p.epi == EPI_F32) ? 4 : 2;
    b.M = p.M - rows_a;
    b.A = (const char*)p.A + (size_t)rows_a * p.K * 2;
    b.out = (char*)p.out + (size_t)rows_a * p.ldo * esz_out;
    if (p.out2) b.out2 = (char*)p.out2 + (size_t)rows_a * p.ldo * 2;
    if (p.aux) b.aux = (const char*)p.aux + (size_t)rows_a * p.ldo * (p.epi == EPI_RESID_F32 ? 4 : 2);
    b.tile = 128;
    b.row0 = p.row0 + rows_a;
    if (int e = dispatch_epi(a, stream)) return e;
    return dispatch_epi(b, stream);
  }
  return dispatch_epi(p, stream);
}

// vitamd_gemm_nt_plan: the kernel form vitamd_gemm_nt_bf16 would launch for these arguments on the current device, without launching:
// form | rows << 8 (| 0x80: the launch is cut into a head of whole rounds in that form and a tail on 128x128 tiles), or -VITAMD_ERR_*.
int vitamd_gemm_nt_plan_impl(const GemmNtArgs& p0) {
  if (int e = check_args(p0)) return -e;
  bool tall = false;
  GemmNtArgs p = p0;
  const int rows_a = tail_split_rows(p, tall);
  if (rows_a) {
    p.M = rows_a;
    p.tile = tall ? p.tile : 256;
  }
  const NtPlan pl = plan_single(p);
  if (pl.err) return -pl.err;
  return pl.form | (pl.rows << 8) | (rows_a ? 0x80 : 0);
}
