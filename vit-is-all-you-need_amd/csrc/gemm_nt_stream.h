// Generalised ping-pong NT GEMM main loop ("stream" kernel): the schedule of gemm_nt_pp_kernel (gemm_nt.hip) with
//   * ring depths per operand (A: 2 K-tiles, B: NBB = 2 or 3 K-tiles) and ring slots chosen at run time (no unrolling by buffer parity),
//   * NP = MT/2 = 3, 4 or 5 phases per K-tile (MT = 6: 192-row tiles, two B pieces requested in phase 0),
//   * one optional global STORE slot per phase (STS = 1), counted into the same vmcnt schedule as the LDS-DMA requests: vmcnt counts
//     loads, stores and LDS-DMA together, in issue order (MI355X_MICROARCH.md, cycle constants), so a store issued between two requests
//     shifts every counted wait whose window it falls into.  The store slot is what lets a tile's output leave the CU in small pieces under
//     the NEXT tile's main loop instead of as one burst at the tile's end.
// C[M,N] = A[M,K] . B[N,K]^T as in gemm_nt.hip (reference transformer.py:21,37,39 forward Linears and their input gradients).
#pragma once
#include <type_traits>
#include "gemm_nt_epilogue.h"

namespace {

// Request schedule in units of phases.  Global phase index P = NP * kt + ph.
//   A-part j of K-tile kt is READ in phase NP kt + j and REQUESTED LA phases earlier (one A request per phase).
//   B block of K-tile kt (4 pieces per wave) is READ in phase NP kt (into registers, for the whole K-tile); piece q is requested
//   blead(q) phases earlier.  NP >= 4: leads 6,5,4,3 (one piece per phase, the schedule of gemm_nt_pp_kernel with LB = 6);
//   NP = 3: leads 6,6,5,4 (pieces 0 and 1 share a phase).
// Program order inside a phase: [STS stores] [A request] [B requests, q ascending] [counted wait].
template <int MT, int LA, int NBB, int STS, bool LATE = false>     // LATE: the store slot sits in the matrix section, after the phase's wait
struct StreamSchedule {
  static constexpr int NP = MT / 2;
  static constexpr int blead(int q) { return NP >= 4 ? 6 - q : (q < 2 ? 6 : 7 - q); }
  static constexpr int max_blead = 6;
  static_assert(MT % 2 == 0 && NP >= 3 && NP <= 5, "tile height");
  static_assert(LA >= 2 && LA <= 2 * NP - 2, "A lead: the region of part j is refilled >= 2 phases after its read (two K-tile slots)");
  static_assert(max_blead <= NBB * NP - 2, "B lead: a ring slot is refilled >= 2 phases after its only read");
  static constexpr int a_part(int ph) { return (ph + LA) % NP; }
  static constexpr int a_tile(int ph) { return (ph + LA) / NP; }
  // B piece q is requested in phase ph iff (ph + blead(q)) % NP == 0, for K-tile kt + (ph + blead(q)) / NP
  static constexpr bool b_here(int ph, int q) { return (ph + blead(q)) % NP == 0; }
  static constexpr int b_tile(int ph, int q) { return (ph + blead(q)) / NP; }
  // operations allowed to stay outstanding after phase ph's requests so that everything first read in phase ph+1 has landed
  static constexpr int wait(int ph) {
    int allowed = 0;
    for (int d = 0; d < 4 * NP; ++d) {                 // walk back over phases ph, ph-1, ...; inside a phase in reverse program order
      const int f = ((ph - d) % NP + NP) % NP;
      if (LATE && d > 0) allowed += STS;               // an earlier phase's store slot: issued after that phase's requests
      for (int q = 3; q >= 0; --q)
        if (b_here(f, q)) {
          if (d + 1 >= blead(q)) return allowed;     // issued d phases ago for the phase blead(q) after it: due by the next phase
          ++allowed;
        }
      if (d + 1 >= LA) return allowed;
      ++allowed;
      if (!LATE) allowed += STS;                       // this phase's store slot(s), issued ahead of its requests: never waited for
    }
    return allowed;
  }
  static constexpr int lookback = (LA > max_blead ? LA : max_blead);
};

struct StreamTrickle {          // timing experiment (experimental builds): dummy stores of the PREVIOUS tile's output shape, spread over the main loop
  int per_tile;                 // live 16-B-per-lane store instructions per wave and tile (0 = none)
  int direct;                   // 0: 8 rows x 128 B per instruction (row-major image), 1: 16 rows x 64 B (accumulator layout)
  int mode;                     // bisect bits: 1 = default cache policy instead of nt, 2 = skip the instruction when nothing is due (counts then wrong: timing only)
  int resident;                 // 1: all stores hit one tile's rows (L2-resident): isolates the instruction stream from the HBM write traffic
};

// SPLIT (experiment): wave row 0 issues EVERY LDS-DMA request (its own piece and its row-1 sibling's) and does the counted waits; wave row 1
// issues no load and never waits in the loop - it carries the store slots (two per phase: its own output and its sibling's), so that no
// load's retirement is ever queued behind a store in a wave's in-order vmcnt counter.
// LEN (timing experiment, results WRONG when > 0): the counted waits allow LEN more operations outstanding than the schedule says, i.e. the
// oldest LEN store slots no longer gate the retirement of younger loads: how much of a trickled store's cost is its acknowledgement latency.
// MID (with STS = 1): 0 = the store slot opens the read section; k > 0 = it sits in the matrix section behind the k-th MFMA, when the
// partner row's burst of LDS-DMA requests has passed the CU's one address/data path to L1
template <int EPI, int MT, int NBB, int STS, bool SPLIT = false, int LEN = 0, int MID = 0>
__global__ __launch_bounds__(512) void gemm_nt_stream_kernel(const GemmNtArgs p, const StreamTrickle tr) {
  constexpr int LA = 4;
  static_assert(!SPLIT || STS == 0, "SPLIT brings its own store slots");
  using S = StreamSchedule<MT, LA, NBB, STS, (MID > 0)>;
  constexpr int RM = SPLIT ? 2 : 1;                 // asm operations per request
  constexpr int NP = S::NP;
  constexpr int BM = 32 * MT, BN = 256, WN = 4, NT = 4;
  constexpr int PART = 8192;                        // one A-part: 64 rows x 128 B (32 rows of each wave row), one 1-KiB piece per wave
  constexpr int ASLOT = NP * PART, BSLOT = 32768;   // one K-tile of A / of B
  constexpr int BBASE = 2 * ASLOT;

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
  const __amdgpu_buffer_rsrc_t rsO = make_rsrc(p.out, (size_t)p.M * p.ldo * (EPI == EPI_RESID_F32 ? 4 : 2));
  const __amdgpu_buffer_rsrc_t rsO2 = make_rsrc(p.out2 ? p.out2 : p.out, (size_t)p.M * p.ldo * 2);
  const int total_phases = nkt * NP;
  int prev_m0 = -1, prev_n0 = 0;

  for (int ti = blockIdx.x; ti < ntiles; ti += (int)gridDim.x) {
    const int tile = xcd_remap(ti, ntiles);
    int tm, tn;
    tile_coords(tile, tiles_m, tiles_n, tiles_n >= 6, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    unsigned voffA[NP], voffB[4], voffA2[SPLIT ? NP : 1], voffB2[SPLIT ? 4 : 1];
    {
      const int lr = 8 * wave + (lane >> 3);
      const unsigned chunk = (unsigned)(((lane & 7) ^ (lr & 7)) * 16);
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const int ga = min(m0 + (lr >> 5) * (16 * MT) + j * 32 + (lr & 31), p.M - 1);
        voffA[j] = (unsigned)ga * (unsigned)(K * 2) + chunk;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int gb = min(n0 + 64 * q + lr, p.N - 1);
        voffB[q] = (unsigned)gb * (unsigned)(K * 2) + chunk;
      }
      if constexpr (SPLIT) {
#pragma unroll
        for (int j = 0; j < NP; ++j) voffA2[j] = (unsigned)min(m0 + 16 * MT + j * 32 + (lr & 31), p.M - 1) * (unsigned)(K * 2) + chunk;
#pragma unroll
        for (int q = 0; q < 4; ++q) voffB2[q] = (unsigned)min(n0 + 64 * q + lr + 32, p.N - 1) * (unsigned)(K * 2) + chunk;
      }
    }
    const unsigned lds0 = lds_addr(smem) + wave * 1024;
    constexpr unsigned OOB = 0x80000000u;
    // ring slots: kt may be negative (prologue) or >= nkt (past the end): the slot arithmetic stays in range, the request goes out of range
    auto request_a = [&](int kt, int j) {
      const bool live = kt >= 0 && kt < nkt;
      if (SPLIT && wm != 0) return;
      asm_glds16(srdA, lds0 + ((kt + 8) & 1) * ASLOT + j * PART, live ? voffA[j] : OOB, live ? (unsigned)kt * 128u : 0u);
      if constexpr (SPLIT) asm_glds16(srdA, lds0 + 4096 + ((kt + 8) & 1) * ASLOT + j * PART, live ? voffA2[j] : OOB, live ? (unsigned)kt * 128u : 0u);
    };
    auto request_b = [&](int kt, int q) {
      const bool live = kt >= 0 && kt < nkt;
      if (SPLIT && wm != 0) return;
      asm_glds16(srdB, lds0 + BBASE + ((kt + 6 * NBB) % NBB) * BSLOT + q * 8192, live ? voffB[q] : OOB, live ? (unsigned)kt * 128u : 0u);
      if constexpr (SPLIT) asm_glds16(srdB, lds0 + 4096 + BBASE + ((kt + 6 * NBB) % NBB) * BSLOT + q * 8192, live ? voffB2[q] : OOB, live ? (unsigned)kt * 128u : 0u);
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frag_off = (lane & 15) * 128 + ((((lane >> 4) ^ (lane & 7)) & 7) << 4);
    const char* const rdA[2] = {smem + wm * 32 * 128 + frag_off, smem + wm * 32 * 128 + (frag_off ^ 64)};
    const char* const rdB[2] = {smem + BBASE + wn * 64 * 128 + frag_off, smem + BBASE + wn * 64 * 128 + (frag_off ^ 64)};

    // dummy-store state (timing experiment): store k of the previous tile goes out in the phase where the running sum crosses a multiple.
    // Rows of the previous tile are all < M here (the experiment's M is a multiple of the tile height, or the last row tile is skipped).
    int st_acc = 0, st_k = 0;
    bool st_second = false;
    constexpr bool F32OUT = EPI == EPI_RESID_F32;
    const int st_per_out = (F32OUT ? 4 : 2) * MT;
    const int rsub = lane >> 3, pc = lane & 7;
    const bool st_on = (STS > 0 || SPLIT) && tr.per_tile > 0 && prev_m0 >= 0 && prev_m0 + BM <= p.M;
    // whole row segments: instruction k covers rows rsub + 8 k (f32: two instructions per 8 rows); accumulator layout: 16 rows x 64 B.
    // tr.resident: every tile's stores go to the first tile's rows (L2-resident lines: no HBM write traffic, same instruction stream)
    const int st_m0 = tr.resident ? 0 : prev_m0;
    const unsigned st_row = (unsigned)(st_m0 + wm * (16 * MT) + (tr.direct ? (lane & 15) : rsub));
    const unsigned st_col = (unsigned)((F32OUT ? 4 : 2) * ((tr.resident ? 0 : prev_n0) + wn * 64)) + (tr.direct ? 16u * (lane >> 4) : 16u * pc);
    const unsigned st_base = st_row * (unsigned)(p.ldo * (F32OUT ? 4 : 2)) + st_col;
    // f32 and accumulator-layout forms take two instructions per row group (the two 128-B / 64-B halves of the wave's row segment)
    const int st_sh = (F32OUT || tr.direct) ? 1 : 0;
    const unsigned st_half = F32OUT ? 128u : 64u;
    const unsigned st_stride = (unsigned)(p.ldo * (F32OUT ? 4 : 2)) * ((tr.direct && !F32OUT) ? 16u : 8u);

#define VITAMD_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#pragma unroll
    for (int P = -S::lookback; P < 0; ++P) {
      const int ph = ((P % NP) + NP) % NP, t = (P - ph) / NP;
      if constexpr (STS > 0 && MID == 0) __builtin_amdgcn_raw_buffer_store_b128((u32x4){0u, 0u, 0u, 0u}, rsO, OOB, 0, 0);
      request_a(t + S::a_tile(ph), S::a_part(ph));
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (S::b_here(ph, q)) request_b(t + S::b_tile(ph, q), q);
      if constexpr (STS > 0 && MID > 0) __builtin_amdgcn_raw_buffer_store_b128((u32x4){0u, 0u, 0u, 0u}, rsO, OOB, 0, 0);
    }
    if (!SPLIT || wm == 0) VITAMD_WAIT_VM(RM * S::wait(NP - 1));
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (wm == 1) __builtin_amdgcn_s_barrier();

    // store slot = address preparation (read section: a few scalar ops, one v_add + one v_cndmask) + the store instruction itself
    unsigned st_voff = OOB;
    auto store_prep = [&]() {
      st_acc += tr.per_tile;
      const bool due = st_on && st_acc >= total_phases;
      if (st_acc >= total_phases) st_acc -= total_phases;
      const unsigned step = (unsigned)(st_k >> st_sh) * st_stride + (unsigned)(st_k & st_sh) * st_half;
      st_voff = due ? st_base + step : OOB;
      st_k += due ? 1 : 0;
      if (st_k >= st_per_out) { st_k = 0; st_second = true; }
    };
    auto store_issue = [&](u32x4 dv) {
      if ((tr.mode & 2) && st_voff == OOB) return;
      if (tr.mode & 1) __builtin_amdgcn_raw_buffer_store_b128(dv, st_second ? rsO2 : rsO, st_voff, 0, 0);
      else __builtin_amdgcn_raw_buffer_store_b128(dv, st_second ? rsO2 : rsO, st_voff, 0, 2);       // aux 2 = nt
    };
    bf16x8 bq[NT][2], af[2][2];
    for (int kt = 0; kt < nkt; ++kt) {
      const int sa = (kt & 1) * ASLOT, sb = (kt % NBB) * BSLOT;
      const char* const pa0 = rdA[0] + sa;
      const char* const pa1 = rdA[1] + sa;
      const char* const pb0 = rdB[0] + sb;
      const char* const pb1 = rdB[1] + sb;
#pragma unroll
      for (int ph = 0; ph < NP; ++ph) {
        if constexpr (STS > 0) store_prep();
        if constexpr (STS > 0 && MID == 0) store_issue((u32x4){(unsigned)kt, 1u, 2u, 3u});
        if constexpr (SPLIT) {
          if (wm == 1) {       // two store slots per phase: this wave's output piece and its row-0 sibling's
            st_acc += tr.per_tile;
            const bool due = st_on && st_acc >= total_phases;
            if (st_acc >= total_phases) st_acc -= total_phases;
            const unsigned step = (unsigned)(st_k >> st_sh) * st_stride + (unsigned)(st_k & st_sh) * st_half;
            const unsigned voff = due ? st_base + step : OOB;
            const unsigned voff2 = due ? st_base + step - (unsigned)(16 * MT) * (unsigned)(p.ldo * (F32OUT ? 4 : 2)) : OOB;
            st_k += due ? 1 : 0;
            if (st_k >= st_per_out) { st_k = 0; st_second = true; }
            u32x4 dv = {(unsigned)kt, 1u, 2u, 3u};
            if (st_second) { __builtin_amdgcn_raw_buffer_store_b128(dv, rsO2, voff, 0, 2); __builtin_amdgcn_raw_buffer_store_b128(dv, rsO2, voff2, 0, 2); }
            else { __builtin_amdgcn_raw_buffer_store_b128(dv, rsO, voff, 0, 2); __builtin_amdgcn_raw_buffer_store_b128(dv, rsO, voff2, 0, 2); }
          }
        }
        if (ph == 0) {
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            bq[j][0] = *(const bf16x8*)(pb0 + j * 2048);
            bq[j][1] = *(const bf16x8*)(pb1 + j * 2048);
          }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          af[i][0] = *(const bf16x8*)(pa0 + ph * PART + i * 2048);
          af[i][1] = *(const bf16x8*)(pa1 + ph * PART + i * 2048);
        }
        request_a(kt + S::a_tile(ph), S::a_part(ph));
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (S::b_here(ph, q)) request_b(kt + S::b_tile(ph, q), q);
        if (!SPLIT || wm == 0) VITAMD_WAIT_VM(RM * S::wait(ph) + LEN);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
              if constexpr (STS > 0 && MID > 0) {
                if (ks * 8 + i * 4 + j == MID) {
                  __builtin_amdgcn_sched_barrier(0);
                  store_issue(__builtin_bit_cast(u32x4, af[0][0]));       // (timing experiment: any live registers serve as data)
                  __builtin_amdgcn_sched_barrier(0);
                }
              }
              acc[2 * ph + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[j][ks], af[i][ks], acc[2 * ph + i][j], 0, 0, 0);
            }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef VITAMD_WAIT_VM
    if constexpr (EPI == EPI_F32) gemm_epilogue<BN, 2, WN, 16 * MT, 64, MT, NT, EPI>(p, acc, m0, n0, wm, wn, lane, tid, smem);
    else if (p.N % 8 == 0 && p.ldo % 8 == 0) gemm_epilogue_rows<EPI, MT>(p, acc, m0, n0, wm, wn, lane, tid, wave, smem);
    else gemm_epilogue<BN, 2, WN, 16 * MT, 64, MT, NT, EPI>(p, acc, m0, n0, wm, wn, lane, tid, smem);
    __syncthreads();
    prev_m0 = m0; prev_n0 = n0;
  }
}

template <int EPI, int MT, int NBB, int STS, bool SPLIT = false, int LEN = 0, int MID = 0>
int launch_stream(const GemmNtArgs& p, hipStream_t stream, StreamTrickle tr, int cus) {
  constexpr int ops_b = 2 * (MT / 2) * 8192 + NBB * 32768, epi_b = 8 * MT * 2048;
  constexpr int lds = ops_b > epi_b ? ops_b : epi_b;
  static_assert(lds <= 160 * 1024, "LDS");
  auto kern = gemm_nt_stream_kernel<EPI, MT, NBB, STS, SPLIT, LEN, MID>;
  if (int e = set_lds(kern, lds)) return e;
  const int tiles = ((p.M + 32 * MT - 1) / (32 * MT)) * ((p.N + 255) / 256);
  hipLaunchKernelGGL(kern, dim3(tiles > cus ? cus : tiles), dim3(512), lds, stream, p, tr);
  return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
}

}  // namespace
