// Loader-wave form of the persistent NT GEMM (round 4): C[M,N] = A[M,K] . B[N,K]^T, 256 x 256 x 64 tiles, TWELVE waves per workgroup.
//
// Why (profiles/r03/nt_request_ablations.log): with every LDS-DMA request taken out of gemm_nt_pp_kernel's main loop the K = 3072 input-gradient
// GEMM runs in 142 us instead of 199; with the requests issued but out of range (no data) 171.  Half of what the operand feed costs is the ISSUE
// of the requests from the compute waves: a 1-KiB vector-memory instruction holds the issuing wave for 60-185 cycles inside a read section that
// its SIMD partner's 256 cycles of MFMAs are supposed to cover.  gemm_tn_ld_kernel (gemm_tn.hip) moved them into four dedicated waves and gained
// 15 %; this is the same step for the NT GEMM:
//   * waves 0-7 compute (2 x 4, wave tile 128 x 64, mfma_f32_16x16x32_bf16 with A/B swapped, the ping-pong of gemm_nt_pp_kernel: the second wave
//     row runs one barrier behind the first) and issue NO vector-memory instruction inside the K loop;
//   * waves 8-11 (one per SIMD) issue every LDS-DMA piece, wait for them with counted vmcnt and take part in every barrier on the first wave
//     row's timeline.  Their vmcnt queues are their own: the compute waves' epilogue stores never enter the counts (the seam kernel's E arithmetic
//     disappears), and the loaders simply keep streaming across tile boundaries - the next tile's first K-tile lands during the last K-tile of
//     the current one, its second one during the epilogue.
// Twelve waves = three per SIMD = at most 168 registers per wave, so the compute waves cannot hold gemm_nt_pp_kernel's fragment set (a 128-register
// accumulator tile + B fragments of a whole K-tile + A fragments: 176 + addressing).  Here a K-tile is walked K-HALF by K-half: phase (ks, h)
// multiplies rows [64 h, 64 h + 64) of the wave tile (4 fragments, 16 registers) with the 64 columns (4 fragments, 16 registers) over the 32-deep
// K-half ks: 16 MFMAs per phase as before, the same LDS read traffic (B fragments are read once per K-half = 8 reads per K-tile, A 16), and
// per accumulator the same k order as gemm_nt_pp_kernel (ks = 0, then 1, K-tiles ascending): results are BIT-IDENTICAL.  128 + 32 + 4 base
// addresses = 164 registers.
// LDS: A(even K-tiles) | A(odd) | B(even) | B(odd), 32 KiB each (every fragment read = one of four base registers + a 16-bit immediate), then
// 8 x 2 KiB wave-private epilogue staging and (GELU table form) the 16-KiB table.  An A block is two REGIONS of 128 rows (rows [64 h, 64 h + 64)
// of both wave rows).  Region life: A0 and B of K-tile t are read in phases 0 and 2, A1 in phases 1 and 3; a region is refilled (with K-tile
// t + 2) two phases after its last read, i.e. loader iteration (t + 1, phase 0) requests B and A0 of K-tile t + 2... in stream order: iteration
// (t, 0) requests B(t + 1), A0(t + 1) into the buffer K-tile t - 1 left, iteration (t, 1) requests A1(t + 1); they are waited for in iterations
// (t, 3) [vmcnt(4): everything but A1(t + 1)] and (t + 1, 0) [vmcnt(12): everything but the 12 new requests]: 3.5 phases in flight, up to 64 KiB
// per CU.  K % 128 == 0 (an even number of K-tiles: the buffer parity is a compile-time constant of the twice-unrolled loop).
#pragma once
#include "gemm_nt_epilogue.h"
#include "gemm_nt_seam.h"

namespace {

// SCHED: how the loaders spread a K-tile's 16 requests (each) over its four phases.  0 = as early as the ring allows (12 in phase 0, 4 in phase 1:
// 3.5 phases in flight, but the burst - ~60 cycles per piece - makes the loaders late for phase 0's barrier); 1 = 6 / 6 / 4 / 0, half of each
// phase's requests behind its first barrier (2.5 phases in flight, <= 360 cycles of issue per phase): the shipped form.
template <int EPI, bool TAB = false, int SCHED = 1>
__global__ __launch_bounds__(768) void gemm_nt_ld_kernel(const GemmNtArgs p) {
  static_assert(EPI == EPI_BIAS_BF16 || EPI == EPI_GELU || EPI == EPI_DGELU, "epilogues of the loader form");
  static_assert(!TAB || EPI == EPI_GELU, "table = GELU");
  constexpr int BM = 256, BN = 256, MT = 8, NT = 4;
  constexpr int AREG = 16384, ABUF = 32768, BBUF = 32768, BBASE = 2 * ABUF, OPS = BBASE + 2 * BBUF;
  constexpr int STG = 2048, TABOFF = OPS + 8 * STG;
  constexpr unsigned OOB = 0x80000000u;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
  const int ntiles = tiles_m * tiles_n;
  const int K = p.K;
  const int nkt = K / 64;
  struct Tile { int m0, n0; };
  auto coords = [&](int ti) {
    int tm, tn;
    tile_coords(xcd_remap(ti, ntiles), tiles_m, tiles_n, tiles_n >= 6, tm, tn);
    return Tile{tm * BM, tn * BN};
  };
#define VITAMD_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

  if (wave >= 8) {
    // ------------------------------------------------------------------------------------------------ loader waves
    const int l = wave - 8;
    const srd_t srdA = make_srd(p.A, (size_t)p.M * K * 2);
    const srd_t srdB = make_srd(p.B, (size_t)p.N * K * 2);
    const srd_t srdBias = make_srd(p.bias ? (const void*)p.bias : p.out, p.bias ? (size_t)p.N * 4 : 0);     // no bias: zero records, every load returns 0
    // pieces of this loader: A region h: pieces a = 4 l + i (LDS rows 8 a .. 8 a + 7 of the region = rows 64 h + 8 (a & 7) + .. of wave row a >> 3);
    // B block: pieces b = 8 l + i (rows 8 b ..).  16-B chunk lane & 7, XOR (LDS row & 7) = lane >> 3 on the SOURCE side.
    const unsigned ldsA = lds_addr(smem) + l * 4096;
    const unsigned ldsB = lds_addr(smem) + BBASE + l * 8192;
    unsigned voffA[2][4], voffB[8];
    auto offsets = [&](const Tile& t) {
      const int r8 = lane >> 3;
      const unsigned chunk = (unsigned)(((lane & 7) ^ r8) * 16);
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int a = 4 * l + i;
          const int row = min(((VITAMD_DBG(p) & 4) ? (t.m0 & 0x3ff) : t.m0) + (a >> 3) * 128 + h * 64 + (a & 7) * 8 + r8, p.M - 1);      // clamp: rows past M are never stored (dbg bit 2, timing only: every tile loads one of four L2-resident A panels)
          voffA[h][i] = (unsigned)row * (unsigned)(K * 2) + chunk;
          if (VITAMD_DBG(p) & 16) voffA[h][i] = (unsigned)min(t.m0, p.M - 256) * (unsigned)(K * 2) + (unsigned)((h * 16 + a) * 1024 + lane * 16);   // (dbg bit 4, timing only: every piece = 1 KiB CONTIGUOUS of the same panel - what a K-tile-blocked operand layout would fetch)
        }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = min(((VITAMD_DBG(p) & 8) ? 0 : t.n0) + (8 * l + i) * 8 + r8, p.N - 1);      // (dbg bit 3, timing only: every tile loads the first B panel)
        voffB[i] = (unsigned)row * (unsigned)(K * 2) + chunk;
        if (VITAMD_DBG(p) & 32) voffB[i] = (unsigned)min(t.n0, p.N - 256) * (unsigned)(K * 2) + (unsigned)((8 * l + i) * 1024 + lane * 16);     // (dbg bit 5: the same for B)
      }
    };
    // K-tile kt of the tile whose offsets are loaded; !live: past the last tile - requested out of range (zero fill, no traffic) so that the
    // counts are the same in every iteration.  (experimental builds, timing only, results garbage: dbg bit 0 = every request out of range - the
    // instruction is issued, nothing is fetched; dbg bit 1 = no request instructions at all)
    bool live = true;
    unsigned so = 0u, soA = 0u, soB = 0u, par = 0u;
    // (a K loop that starts at a different K-tile per row panel - so that the CUs of different panels do not ask L2 for the same B lines at the same moment - was
    // measured equal: 1 308 against 1 296 us per layer, profiles/r04/nt_loader_k_rotation.log; removed)
    // (cache policies on these requests - nt, sc1, sc0 sc1 - were measured in round 4: sc1 equal, nt and sc0 sc1 3-8 % slower, and the two extra scalar
    // branches per request that selecting them at run time cost made the whole kernel 20-40 % slower: the loaders' issue loop is on every barrier's
    // critical path; profiles/r04/nt_loader_cache_policy.log)
    auto request_b = [&](int i) {
      if (VITAMD_DBG(p) & 2) return;
      asm_glds16(srdB, ldsB + par * BBUF + i * 1024, live ? voffB[i] : OOB, soB);
    };
    auto request_a = [&](int h, int i) {
      if (VITAMD_DBG(p) & 2) return;
      asm_glds16(srdA, ldsA + par * ABUF + h * AREG + i * 1024, live ? voffA[h][i] : OOB, soA);
    };
    auto target = [&](int kt, bool lv) {                // the K-tile the following requests fetch
      live = lv && !(VITAMD_DBG(p) & 1);
      so = live ? (unsigned)kt * 128u : 0u;
      soA = (VITAMD_DBG(p) & 16) ? so * 256u : so;      // contiguous-piece ablations: a K-tile of a 256-row panel = 32 KiB
      soB = (VITAMD_DBG(p) & 32) ? so * 256u : so;
      par = (unsigned)(kt & 1);
    };
    __builtin_amdgcn_s_setprio(3);                      // the loaders' few instructions go first: a late request costs every wave of the workgroup
    int ti = blockIdx.x;
    Tile cur = coords(ti);
    offsets(cur);
    int bias_n0 = cur.n0;
    target(0, true);
#pragma unroll
    for (int i = 0; i < 8; ++i) request_b(i);
#pragma unroll
    for (int i = 0; i < 4; ++i) request_a(0, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) request_a(1, i);
    if (VITAMD_DBG(p) & 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else
    VITAMD_WAIT_VM(4);                                  // B(0), A0(0) landed; A1(0) is waited for in iteration (0, 0)
    __builtin_amdgcn_s_barrier();                       // START
    for (;;) {
      const int ti_next = ti + (int)gridDim.x;
      const bool has_next = ti_next < ntiles;
      for (int kt = 0; kt < nkt; ++kt) {
        const bool last = kt + 1 == nkt;
        if (last && has_next) {
          cur = coords(ti_next);
          offsets(cur);
        }
        target(last ? 0 : kt + 1, !last || has_next);
        // The buffer these requests fill held K-tile kt - 1: its B and A0 parts were last read in phase 2 of that K-tile (free from phase 0
        // of this one), its A1 part in phase 3 (free behind the first barrier of phase 0).  First reads: B, A0 in phase 0 of the next
        // K-tile (waited for in phase 3), A1 in its phase 1 (waited for in its phase 0).
        if constexpr (SCHED == 0) {
          // ---- phase 0
#pragma unroll
          for (int i = 0; i < 8; ++i) request_b(i);
#pragma unroll
          for (int i = 0; i < 4; ++i) request_a(0, i);
          __builtin_amdgcn_s_barrier();
          VITAMD_WAIT_VM(12);                           // A1(kt) (first read in phase 1) has landed
          __builtin_amdgcn_s_barrier();
          // ---- phase 1
          if constexpr (EPI != EPI_DGELU) {
            if (kt == 0) {
              asm_glds4(srdBias, lds_addr(smem) + OPS + l * STG, (unsigned)(bias_n0 + l * 64 + lane) * 4u, 0u);
              asm_glds4(srdBias, lds_addr(smem) + OPS + (4 + l) * STG, (unsigned)(bias_n0 + l * 64 + lane) * 4u, 0u);
            }
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) request_a(1, i);
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_s_barrier();
          // ---- phase 2
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_s_barrier();
        } else {
          // ---- phase 0: B pieces 0-5
          request_b(0); request_b(1); request_b(2);
          __builtin_amdgcn_s_barrier();
          request_b(3); request_b(4); request_b(5);
          VITAMD_WAIT_VM(6);                            // A1(kt) (first read in phase 1) has landed
          __builtin_amdgcn_s_barrier();
          // ---- phase 1: B pieces 6, 7 and A0
          if constexpr (EPI != EPI_DGELU) {
            // first K-tile of a tile: every compute wave is inside the main loop, its staging image idle - the bias of ITS 64 columns goes there
            // (256 B by LDS-DMA; both wave rows), long before the epilogue reads it
            if (kt == 0) {
              asm_glds4(srdBias, lds_addr(smem) + OPS + l * STG, (unsigned)(bias_n0 + l * 64 + lane) * 4u, 0u);
              asm_glds4(srdBias, lds_addr(smem) + OPS + (4 + l) * STG, (unsigned)(bias_n0 + l * 64 + lane) * 4u, 0u);
            }
          }
          request_b(6); request_b(7); request_a(0, 0);
          __builtin_amdgcn_s_barrier();
          request_a(0, 1); request_a(0, 2); request_a(0, 3);
          __builtin_amdgcn_s_barrier();
          // ---- phase 2: A1
          request_a(1, 0); request_a(1, 1);
          __builtin_amdgcn_s_barrier();
          request_a(1, 2); request_a(1, 3);
          __builtin_amdgcn_s_barrier();
        }
        // ---- phase 3
        __builtin_amdgcn_s_barrier();
        if (VITAMD_DBG(p) & 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else
        VITAMD_WAIT_VM(4);                              // B, A0 of the next K-tile (first read in its phase 0) have landed; its A1 may be in flight
        __builtin_amdgcn_s_barrier();
      }
      __builtin_amdgcn_s_barrier();                     // the first wave row's balancing barrier
      if (!has_next) break;
      ti = ti_next;
      bias_n0 = cur.n0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the past-the-end pieces (zeros) must not outlive the workgroup's LDS
    return;
  }

  // -------------------------------------------------------------------------------------------------- compute waves
  const int wm = wave >> 2, wn = wave & 3;
  const srd_t rsO = make_srd(p.out, (size_t)p.M * p.ldo * 2);
  const srd_t rsO2 = make_srd(EPI == EPI_GELU ? p.out2 : p.out, (size_t)p.M * p.ldo * 2);
  const srd_t srdAux = make_srd(EPI == EPI_DGELU ? p.aux : p.out, (size_t)p.M * p.ldo * 2);
  if constexpr (TAB) {                                  // 16 KiB, once per (persistent) workgroup; visible to every wave after the START barrier
    const u32x4* src = (const u32x4*)p.gelu_tab + 2 * tid;
    const u32x4 t0 = src[0], t1 = src[1];
    u32x4* dst = (u32x4*)(smem + TABOFF) + 2 * tid;
    dst[0] = t0;
    dst[1] = t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();                         // START: the loaders waited for the first K-tile's B and A0
  asm volatile("" ::: "memory");
  for (int ti = blockIdx.x; ti < ntiles; ti += (int)gridDim.x) {
    const Tile cur = coords(ti);
    if (wm == 1) __builtin_amdgcn_s_barrier();          // second wave row: one barrier behind inside a tile
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
      // fragment reads: 16-row tile at LDS row rb: lane -> row rb + (lane & 15), chunk ((lane >> 4) + 4 ks) ^ (row & 7); K-half 1 flips chunk
      // bit 2 = XOR 64 on the swizzled offset, hence one base per K-half
      int l2 = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      asm volatile("" : "+v"(l2));                      // re-derived per tile, never carried across an epilogue
      const int frag_off = (l2 & 15) * 128 + ((((l2 >> 4) ^ (l2 & 7)) & 7) << 4);
      const char* const rdA[2] = {smem + wm * 8192 + frag_off, smem + wm * 8192 + (frag_off ^ 64)};                  // + buffer ABUF + h AREG + ii 2048
      const char* const rdB[2] = {smem + BBASE + wn * 8192 + frag_off, smem + BBASE + wn * 8192 + (frag_off ^ 64)};  // + buffer BBUF + j 2048
      bf16x8 bq[NT], af[4];
      auto ktile = [&](auto bufc) {
        constexpr int BUF = decltype(bufc)::value;
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
          const int ks = ph >> 1, h = ph & 1;
          if (h == 0) {
#pragma unroll
            for (int j = 0; j < NT; ++j) bq[j] = *(const bf16x8*)(rdB[ks] + BUF * BBUF + j * 2048);
          }
#pragma unroll
          for (int ii = 0; ii < 4; ++ii) af[ii] = *(const bf16x8*)(rdA[ks] + BUF * ABUF + h * AREG + ii * 2048);
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_barrier();
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int ii = 0; ii < 4; ++ii)
#pragma unroll
            for (int j = 0; j < NT; ++j)
              acc[4 * h + ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[j], af[ii], acc[4 * h + ii][j], 0, 0, 0);
          __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
        }
      };
      for (int kt = 0; kt < nkt; kt += 2) {
        ktile(std::integral_constant<int, 0>{});
        ktile(std::integral_constant<int, 1>{});
      }
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();          // balance the stagger: every wave is past its last operand read

    // ---- epilogue: wave-private (no workgroup barrier; the loaders are already filling the next tile's K-tiles).  Lane roles from a fresh lane
    // id: values computed before the main loop would be kept in - or spilled from - registers across it.
    // Accumulator layout: row mloc of a 16-row slice, columns 16 j + 4 g ..; row-major view: row rsub + 8 h, 16-B chunk pc.
    int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(ln));
    const int mloc = ln & 15, g = ln >> 4, rsub = ln >> 3, pc = ln & 7;
    char* const stg = smem + OPS + wave * STG;
    const int m0 = cur.m0, n0 = cur.n0;
    const int ncol = n0 + wn * 64 + 8 * (pc ^ rsub);
    const bool ncol_ok = ncol < p.N;
    const int mrow0 = m0 + wm * 128 + rsub;             // + 16 i + 8 h
    const unsigned obase = ncol_ok ? (unsigned)mrow0 * (unsigned)(p.ldo * 2) + (unsigned)ncol * 2u : OOB;
    const unsigned rstep = (unsigned)(p.ldo * 2) * 8u;
    u32x4 aux[EPI == EPI_DGELU ? 2 * MT : 1];
    u32x2 pk[EPI == EPI_DGELU ? MT : 1][NT];            // dGELU: the tile rounded to bf16 BEFORE the factors are loaded (64 + 64 registers)
    f32x4 bias4[NT];
    if constexpr (EPI == EPI_DGELU) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) pk[i][j] = (u32x2){pack_bf16x2(acc[i][j][0], acc[i][j][1]), pack_bf16x2(acc[i][j][2], acc[i][j][3])};
#pragma unroll
      for (int it = 0; it < 2 * MT; ++it) aux[it] = asm_bload16(srdAux, mrow0 + 8 * it < p.M ? obase : OOB, rstep * (unsigned)it);      // rows >= M: out of range -> 0
      static_assert(MT == 8, "16 pre-load registers named in one asm statement");
      // the only vector-memory operations of this wave that can be outstanding here are the previous tile's stores and these 16 loads
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(aux[0]), "+v"(aux[1]), "+v"(aux[2]), "+v"(aux[3]), "+v"(aux[4]), "+v"(aux[5]), "+v"(aux[6]), "+v"(aux[7]), "+v"(aux[8]), "+v"(aux[9]), "+v"(aux[10]), "+v"(aux[11]), "+v"(aux[12]), "+v"(aux[13]), "+v"(aux[14]), "+v"(aux[15]) :: "memory");
    } else {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) bias4[j][r] = round_bf16(((const float*)stg)[16 * j + 4 * g + r]);     // autocast casts the bias to bf16
    }
    float cs[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) cs[c] = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        u32x2 o;
        if constexpr (EPI == EPI_DGELU) o = pk[i][j];
        else {
          const f32x4 v = acc[i][j] + bias4[j];
          o = (u32x2){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        }
        *(u32x2*)(stg + mloc * 128 + (((2 * j + (g >> 1)) ^ (mloc & 7)) << 4) + (g & 1) * 8) = o;
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const u32x4 v = *(const u32x4*)(stg + (rsub + 8 * h) * 128 + pc * 16);
        const unsigned voff = mrow0 + 16 * i + 8 * h < p.M ? obase : OOB;
        const unsigned soff = rstep * (unsigned)(2 * i + h);
        if constexpr (EPI == EPI_BIAS_BF16) {
          asm_bstore16_nt(v, rsO, voff, soff);
        } else if constexpr (EPI == EPI_GELU) {
          u32x4 a, d = v;
          if constexpr (TAB) gelu_lookup8(v, smem + TABOFF, p.gelu_dg != 0, a, d);
          else
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              float dlo, dhi;
              a[c] = pack_bf16x2(gelu_fwd_grad(bf16lo(v[c]), dlo), gelu_fwd_grad(bf16hi(v[c]), dhi));
              if (p.gelu_dg) d[c] = pack_bf16x2(dlo, dhi);                       // `out` carries gelu'(pre) for the backward
            }
          asm_bstore16_nt(d, rsO, voff, soff);
          asm_bstore16_nt(a, rsO2, voff, soff);
        } else {
          const u32x4 pz = aux[2 * i + h];
          u32x4 o;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float lo = round_bf16(bf16lo(v[c]) * bf16lo(pz[c]));          // aux holds gelu'(pre) (stored-derivative form only: ld_ok)
            const float hi = round_bf16(bf16hi(v[c]) * bf16hi(pz[c]));
            if (voff != OOB) { cs[2 * c] += lo; cs[2 * c + 1] += hi; }
            o[c] = pack_bf16x2(lo, hi);
          }
          asm_bstore16_nt(o, rsO, voff, soff);
        }
      }
    }
    if constexpr (EPI == EPI_DGELU) {
      // column sums of the stored tile (bias gradient of the producing Linear): the butterfly of gemm_nt_seam_kernel, one atomic instruction per wave
#pragma unroll
      for (int bf = 1; bf < 8; bf <<= 1)
#pragma unroll
        for (int c = 0; c < 8; ++c) cs[c] += __shfl_xor(cs[c], 9 * bf, 64);
      float s = cs[0];
#pragma unroll
      for (int c = 1; c < 8; ++c) s = rsub == c ? cs[c] : s;
      atomicAdd(p.colsum + n0 + wn * 64 + 8 * (pc ^ rsub) + rsub, s);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the staging reads are done before the next tile's bias lands there (it is requested behind barriers this wave has yet to reach)
  }
#undef VITAMD_WAIT_VM
}

template <int EPI, bool TAB = false, int SCHED = 1>
int launch_ld(const GemmNtArgs& p, hipStream_t stream, int cus) {
  auto kern = gemm_nt_ld_kernel<EPI, TAB, SCHED>;
  if (TAB && !p.gelu_tab) return VITAMD_ERR_ARG;
  if (int e = set_lds(kern, 160 * 1024)) return e;
  const int tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
  hipLaunchKernelGGL(kern, dim3(tiles > cus ? cus : tiles), dim3(768), 160 * 1024, stream, p);
  return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
}

// launch conditions of the loader form: the seam kernel's, and an even number of K-tiles
static bool ld_ok(const GemmNtArgs& p) { return seam_ok(p) && p.K % 128 == 0; }

}  // namespace
