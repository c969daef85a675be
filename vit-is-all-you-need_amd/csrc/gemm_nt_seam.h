// Persistent ping-pong NT GEMM whose tile SEAMS overlap: the next tile's pipeline fill is requested BEFORE the current tile's epilogue,
// and the epilogue runs out of a small wave-private LDS staging area beside the operand buffers instead of on top of them.
//
// Why (round 3, profiles/r03/store_trickle_README.md): per phase the K = 768 launches of the training step run 27 % slower than the
// K = 3072 ones - a 12-K-tile tile pays 2-4.6 us at its boundary: epilogue (LDS transposes, 16-32 store instructions per wave through
// the CU's one path to L1), then a COLD pipeline fill (first request -> first data: an HBM round trip under load), then the ramp.
// Spreading the stores over the next main loop was measured and loses (every extra vector-memory instruction in a read section costs
// 100-190 cycles of phase time); what can run beside the epilogue is the fill.  Here, after a tile's last phase:
//     rebalance barrier -> [auxiliary loads of the epilogue] -> next tile's prologue requests (13-15 LDS-DMA per wave)
//     -> epilogue (wave-private staging, burst of stores) -> counted wait that EXCLUDES the epilogue's stores -> barrier -> next tile
// vmcnt counts loads, stores, atomics and LDS-DMA together, in issue order, so the waits of the first K-tile behind a seam allow the
// epilogue's E operations on top of the schedule's count wherever the request they wait for was issued before the epilogue.
// What the epilogue must LOAD (bias, the dGELU factor) is requested from inline asm just ahead of the next tile's requests and waited for
// once, with a count that leaves those requests in flight (a compiler-visible load there would be waited for with vmcnt(0)).  For the counts
// every wave must issue EXACTLY E vector-memory instructions per epilogue: stores are buffer stores with out-of-range offsets for
// masked rows / columns (issued, dropped by the range check), never branches.
//
// C[M,N] = A[M,K] . B[N,K]^T with the bias / GELU / dGELU epilogues of gemm_nt_epilogue.h (reference transformer.py:21,37-39 and their
// input gradients).  Tile (32 MT) x 256 x 64, MT = 8 or 10; schedule = gemm_nt_pp_kernel's (LA = 4, B leads 6,5,4,3), ring slots at run time.
#pragma once
#include "gemm_nt_epilogue.h"

namespace {

// (A request-placement experiment of round 3 - the B request, or both requests, issued from inside the matrix section - measured slower everywhere
// and was removed in round 4: profiles/r03/request_placement_experiment.log.)  Both requests of a phase are issued in its read section, ahead of the
// phase's counted wait.
template <int MT, int LA>
struct SeamSchedule {
  static constexpr int NP = MT / 2;
  static_assert(MT % 2 == 0 && NP >= 4 && NP <= 5, "tile height");
  static_assert(LA >= 2 && LA <= 2 * NP - 2, "A lead");
  static constexpr int blead(int q) { return 6 - q; }
  static constexpr int lookback = LA > 6 ? LA : 6;
  static constexpr int prologue_requests() {       // LDS-DMA instructions per wave in one tile's prologue (the replayed lookback phases)
    int n = 0;
    for (int P = -lookback; P < 0; ++P) {
      const int ph = ((P % NP) + NP) % NP;
      ++n;
      for (int q = 0; q < 4; ++q) n += ((ph + 6 - q) % NP == 0) ? 1 : 0;
    }
    return n;
  }
  static constexpr int a_part(int ph) { return (ph + LA) % NP; }
  static constexpr int a_tile(int ph) { return (ph + LA) / NP; }
  static constexpr bool b_here(int ph, int q) { return (ph + blead(q)) % NP == 0; }
  static constexpr int b_tile(int ph, int q) { return (ph + blead(q)) / NP; }
  // operations allowed outstanding after phase ph's requests so that everything first read in phase ph + 1 has landed.  E: operations the
  // epilogue of the previous tile put between that tile's requests and this tile's phase 0 (first K-tile behind a seam only): they are
  // younger than any request issued before the seam, so a wait for such a request leaves them outstanding too.
  static constexpr int wait(int ph, int E = 0) {
    int allowed = 0;
    for (int d = 0; d < 4 * NP; ++d) {
      const int f = ((ph - d) % NP + NP) % NP;
      // reverse program order inside phase ph - d: its wait | [B requests] [A request]
      for (int q = 3; q >= 0; --q)
        if (b_here(f, q)) {
          if (d + 1 >= blead(q)) return allowed + (d > ph ? E : 0);
          ++allowed;
        }
      if (d + 1 >= LA) return allowed + (d > ph ? E : 0);
      ++allowed;
    }
    return allowed;
  }
};

template <int EPI, int MT, bool TAB = false>
__global__ __launch_bounds__(512) void gemm_nt_seam_kernel(const GemmNtArgs p) {
  static_assert(EPI == EPI_BIAS_BF16 || EPI == EPI_GELU || EPI == EPI_DGELU, "epilogues without (or with up-front) auxiliary loads");
  static_assert(!TAB || (EPI == EPI_GELU && MT == 8), "the GELU table needs the 16 KiB that only the 256-row ring leaves");
  constexpr int LA = 4;
  using S = SeamSchedule<MT, LA>;
  constexpr int NP = S::NP;
  constexpr int BM = 32 * MT, BN = 256, WN = 4, NT = 4;
  constexpr int PART = 8192, ASLOT = NP * PART, BSLOT = 32768, BBASE = 2 * ASLOT, OPS = BBASE + 2 * BSLOT;
  constexpr int STG = TAB ? 2048 : (160 * 1024 - OPS) / 8;       // wave-private staging: 4 KiB (MT = 8) / 2 KiB (MT = 10, or MT = 8 with the GELU table behind it)
  constexpr int TABOFF = OPS + 8 * STG;              // TAB: gelu_table() image, 4096 x 4 B
  constexpr int SL = STG / 2048;                     // 16-row slices staged per round
  static_assert(SL >= 1 && MT % SL == 0, "staging");
  // vector-memory instructions every wave issues per epilogue BEHIND the next tile's requests
  constexpr int E = EPI == EPI_GELU ? 4 * MT : EPI == EPI_DGELU ? 2 * MT + 1 : 2 * MT;
  static_assert(S::wait(NP - 1) + E <= 63 && S::wait(0, E) <= 63, "vmcnt is a 6-bit counter");

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
  const srd_t rsO = make_srd(p.out, (size_t)p.M * p.ldo * 2);
  const srd_t rsO2 = make_srd(EPI == EPI_GELU ? p.out2 : p.out, (size_t)p.M * p.ldo * 2);
  const srd_t srdAux = make_srd(EPI == EPI_DGELU ? p.aux : p.out, (size_t)p.M * p.ldo * 2);
  const srd_t srdBias = make_srd(p.bias ? (const void*)p.bias : p.out, p.bias ? (size_t)p.N * 4 : 0);     // no bias: zero records, every load returns 0
  constexpr unsigned OOB = 0x80000000u;
  const unsigned lds0 = lds_addr(smem) + wave * 1024;

  struct Tile { int m0, n0; };
  auto coords = [&](int ti) {
    int tm, tn;
    tile_coords(xcd_remap(ti, ntiles), tiles_m, tiles_n, tiles_n >= 6, tm, tn);
    return Tile{tm * BM, tn * BN};
  };
  unsigned voffA[NP], voffB[4];
  auto offsets = [&](const Tile& t) {
    int l2 = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(l2));                  // opaque (re-derived per tile, never carried across a main loop)
    const int lr = 8 * wave + (l2 >> 3);                              // LDS row of this lane inside an A-part / a 64-row B piece
    const unsigned chunk = (unsigned)(((l2 & 7) ^ (lr & 7)) * 16);      // 16-B chunk, XOR (row & 7) on the SOURCE side
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      unsigned o = (unsigned)min(t.m0 + (lr >> 5) * (16 * MT) + j * 32 + (lr & 31), p.M - 1) * (unsigned)(K * 2);
      asm volatile("" : "+v"(o));                   // keeps the product a 32-bit v_mul_lo (hipcc otherwise forms v_mad_u64_u32: a register PAIR per offset)
      voffA[j] = o + chunk;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned o = (unsigned)min(t.n0 + 64 * q + lr, p.N - 1) * (unsigned)(K * 2);
      asm volatile("" : "+v"(o));
      voffB[q] = o + chunk;
    }
  };
  // K-tiles < 0 (prologue replay) and >= nkt (past the end) are requested out of range: zero fill, no traffic, same counts in every phase
  bool fill = true;            // false: the tile whose fill is being requested does not exist (last seam): every request goes out of range
  auto request_a = [&](int kt, int j) {
    const bool live = fill && kt >= 0 && kt < nkt;
    asm_glds16(srdA, lds0 + ((kt + 8) & 1) * ASLOT + j * PART, live ? voffA[j] : OOB, live ? (unsigned)kt * 128u : 0u);
  };
  auto request_b = [&](int kt, int q) {
    const bool live = fill && kt >= 0 && kt < nkt;
    asm_glds16(srdB, lds0 + BBASE + ((kt + 8) & 1) * BSLOT + q * 8192, live ? voffB[q] : OOB, live ? (unsigned)kt * 128u : 0u);
  };
  auto prologue = [&]() {      // the requests of the S::lookback phases before a tile's phase 0
#pragma unroll
    for (int P = -S::lookback; P < 0; ++P) {
      const int ph = ((P % NP) + NP) % NP, t = (P - ph) / NP;
      request_a(t + S::a_tile(ph), S::a_part(ph));
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (S::b_here(ph, q)) request_b(t + S::b_tile(ph, q), q);
    }
  };

  const int frag_off = (lane & 15) * 128 + ((((lane >> 4) ^ (lane & 7)) & 7) << 4);
  const char* const rdA[2] = {smem + wm * 32 * 128 + frag_off, smem + wm * 32 * 128 + (frag_off ^ 64)};
  const char* const rdB[2] = {smem + BBASE + wn * 64 * 128 + frag_off, smem + BBASE + wn * 64 * 128 + (frag_off ^ 64)};

#define VITAMD_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
  if constexpr (TAB) {                               // 16 KiB, once per (persistent) workgroup; visible to every wave after the first barrier below
    const u32x4* src = (const u32x4*)p.gelu_tab + 2 * tid;
    const u32x4 t0 = src[0], t1 = src[1];
    u32x4* dst = (u32x4*)(smem + TABOFF) + 2 * tid;
    dst[0] = t0;
    dst[1] = t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  int ti = blockIdx.x;
  Tile cur = coords(ti);
  offsets(cur);
  prologue();
  VITAMD_WAIT_VM(S::wait(NP - 1));
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (wm == 1) __builtin_amdgcn_s_barrier();        // second wave row: one barrier behind from here on
  bool seam = false;

  for (;;) {
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    bf16x8 bq[NT][2], af[2][2];
    auto ktile = [&](int kt) {
      const bool lenient = seam && kt == 0;          // first K-tile behind a seam: the previous epilogue's E operations may still be in flight
      const int sa = (kt & 1) * ASLOT, sb = (kt & 1) * BSLOT;
      const char* const pa0 = rdA[0] + sa;
      const char* const pa1 = rdA[1] + sa;
      const char* const pb0 = rdB[0] + sb;
      const char* const pb1 = rdB[1] + sb;
#pragma unroll
      for (int ph = 0; ph < NP; ++ph) {
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
        if (S::wait(ph, E) != S::wait(ph) && lenient) VITAMD_WAIT_VM(S::wait(ph, E));
        else VITAMD_WAIT_VM(S::wait(ph));
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
              acc[2 * ph + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[j][ks], af[i][ks], acc[2 * ph + i][j], 0, 0, 0);
            }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    };
    for (int kt = 0; kt < nkt; ++kt) ktile(kt);
    if (wm == 0) __builtin_amdgcn_s_barrier();        // balance the stagger: every wave is past its last operand read
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // only past-the-end requests (zero fill) can still be in flight: cheap

    // ---- seam: auxiliary loads, then the next tile's fill, then this tile's epilogue.
    // Lane roles are re-derived here from a fresh lane id (v_mbcnt): values computed at kernel entry would be kept in (or spilled from)
    // registers across the whole main loop, and a spill reload brings a compiler-made vmcnt(0) into the counted pipeline.
    // Accumulator layout: row mloc of a 16-row slice, columns 16 j + 4 g ..; row-major view: row rsub + 8 h, 16-B chunk pc.
    int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(ln));                  // opaque: not to be merged with any earlier lane-id computation
    const int mloc = ln & 15, g = ln >> 4, rsub = ln >> 3, pc = ln & 7;
    char* const stg = smem + OPS + wave * STG;
    const int m0 = cur.m0, n0 = cur.n0;
    const int ncol = n0 + wn * 64 + 8 * (pc ^ rsub);            // this lane's 8 columns in the row-major view
    const bool ncol_ok = ncol < p.N;
    const int mrow0 = m0 + wm * (16 * MT) + rsub;               // + 16 i + 8 h
    // one per-lane byte offset for every output / auxiliary access of the tile; the row step 8 (2 i + h) ldo rides in the scalar offset
    // (sixteen per-row offsets would otherwise stay live from the auxiliary loads to the stores)
    const unsigned obase = ncol_ok ? (unsigned)mrow0 * (unsigned)(p.ldo * 2) + (unsigned)ncol * 2u : OOB;
    const unsigned rstep = (unsigned)(p.ldo * 2) * 8u;
    // pre-loads (inline asm, counted by hand): bias (fp32, 4 x 4 columns of the accumulator layout) or the dGELU factor (row-major view)
    u32x4 aux[EPI == EPI_DGELU ? 2 * MT : 1];
    u32x2 pk[EPI == EPI_DGELU ? MT : 1][NT];         // dGELU: the tile rounded to bf16 BEFORE the factor is loaded (64 + 64 registers instead of 128 + 64)
    if constexpr (EPI == EPI_DGELU) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) pk[i][j] = (u32x2){pack_bf16x2(acc[i][j][0], acc[i][j][1]), pack_bf16x2(acc[i][j][2], acc[i][j][3])};
#pragma unroll
      for (int it = 0; it < 2 * MT; ++it) {
#ifdef VITAMD_EXPERIMENTAL
        aux[it] = asm_bload16(srdAux, (mrow0 + 8 * it < p.M && !(p.dbg & 4)) ? obase : OOB, rstep * (unsigned)it);      // (dbg bit 2, timing only: no factor traffic)
#else
        aux[it] = asm_bload16(srdAux, mrow0 + 8 * it < p.M ? obase : OOB, rstep * (unsigned)it);      // rows >= M: out of range -> 0
#endif
      }
    } else {
      // the wave's 64 bias values -> the head of its (idle) staging area, by LDS-DMA: nothing lands in a register before the wait below
      // (an asm load into a 128-bit register tuple was tried first: hipcc treated three of its four dwords as undefined)
      asm_glds4(srdBias, lds_addr(stg), (unsigned)(n0 + wn * 64 + ln) * 4u, 0u);              // columns >= N, or no bias: out of range -> 0
    }
    const int ti_next = ti + (int)gridDim.x;
    const bool has_next = ti_next < ntiles;
    if (has_next) {
      cur = coords(ti_next);
      offsets(cur);
    }
    fill = has_next;
    prologue();                  // always issued (out of range behind the last tile): the counts below do not depend on has_next
    fill = true;
    constexpr int NREQ = S::prologue_requests();
    // ONE wait statement on every path: the pre-loaded registers are operands, so nothing that reads them can be scheduled above it
    // (two statements in the arms of a branch made hipcc copy the registers - before the data had landed - ahead of one of them)
    // (dGELU: one wait per ROUND - round 0 starting when 4 of the 16 factor loads are back - was measured equal, 275.6 against 276.6 us: what the
    // factor loads cost is their issue and their HBM bytes, not the latency of the last of them)
    if constexpr (EPI == EPI_DGELU) {
      static_assert(MT == 8, "16 pre-load registers named in one asm statement");
      asm volatile("s_waitcnt vmcnt(%16)" : "+v"(aux[0]), "+v"(aux[1]), "+v"(aux[2]), "+v"(aux[3]), "+v"(aux[4]), "+v"(aux[5]), "+v"(aux[6]), "+v"(aux[7]), "+v"(aux[8]), "+v"(aux[9]), "+v"(aux[10]), "+v"(aux[11]), "+v"(aux[12]), "+v"(aux[13]), "+v"(aux[14]), "+v"(aux[15]) : "n"(NREQ) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NREQ) : "memory");
    }
    // ---- epilogue: SL slices per round through the wave-private staging image ([16 rows][64 bf16] per slice, 16-B chunk XOR (row & 7))
    f32x4 bias4[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      bias4[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if constexpr (EPI != EPI_DGELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) bias4[j][r] = round_bf16(((const float*)stg)[16 * j + 4 * g + r]);     // autocast casts the bias to bf16
      }
    }
    float cs[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) cs[c] = 0.f;
#pragma unroll
    for (int rd = 0; rd < MT / SL; ++rd) {
#pragma unroll
      for (int s = 0; s < SL; ++s) {
        const int i = rd * SL + s;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          u32x2 o;
          if constexpr (EPI == EPI_DGELU) o = pk[i][j];
          else {
            const f32x4 v = acc[i][j] + bias4[j];
            o = (u32x2){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
          }
          *(u32x2*)(stg + s * 2048 + mloc * 128 + (((2 * j + (g >> 1)) ^ (mloc & 7)) << 4) + (g & 1) * 8) = o;
        }
      }
#pragma unroll
      for (int s = 0; s < SL; ++s) {
        const int i = rd * SL + s;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const u32x4 v = *(const u32x4*)(stg + s * 2048 + (rsub + 8 * h) * 128 + pc * 16);
          const unsigned voff = mrow0 + 16 * i + 8 * h < p.M ? obase : OOB;
          const unsigned soff = rstep * (unsigned)(2 * i + h);
          if constexpr (EPI == EPI_BIAS_BF16) {
            asm_bstore16_nt(v, rsO, voff, soff);
          } else if constexpr (EPI == EPI_GELU) {
            u32x4 a, d = v;
#ifdef VITAMD_EXPERIMENTAL
            if (p.dbg & 1) a = v;                                                // timing-only ablation: no erf / exp
            else
#endif
            if constexpr (TAB) gelu_lookup8(v, smem + TABOFF, p.gelu_dg != 0, a, d);
            else
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              float dlo, dhi;
              a[c] = pack_bf16x2(gelu_fwd_grad(bf16lo(v[c]), dlo), gelu_fwd_grad(bf16hi(v[c]), dhi));
              if (p.gelu_dg) d[c] = pack_bf16x2(dlo, dhi);                       // `out` carries gelu'(pre) for the backward
            }
#ifdef VITAMD_EXPERIMENTAL
            if (!(p.dbg & 2))                                                    // timing-only ablation: no second output
#endif
            asm_bstore16_nt(d, rsO, voff, soff);
            asm_bstore16_nt(a, rsO2, voff, soff);
          } else {
            const u32x4 pz = aux[2 * i + h];
            u32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const float lo = round_bf16(bf16lo(v[c]) * bf16lo(pz[c]));        // aux holds gelu'(pre) (stored-derivative form only: seam_ok)
              const float hi = round_bf16(bf16hi(v[c]) * bf16hi(pz[c]));
              if (voff != OOB) { cs[2 * c] += lo; cs[2 * c + 1] += hi; }
              o[c] = pack_bf16x2(lo, hi);
            }
            asm_bstore16_nt(o, rsO, voff, soff);
          }
        }
      }
    }
#ifdef VITAMD_EXPERIMENTAL
    if (EPI == EPI_DGELU && !(p.dbg & 8)) {         // (dbg bit 3, timing only: no column sums)
#else
    if constexpr (EPI == EPI_DGELU) {
#endif
      // column sums of the stored tile (bias gradient of the producing Linear).  Lanes (rsub, pc) with equal pc ^ rsub hold partial sums of the
      // same 8 columns (their rows differ): a butterfly over rsub (partner lane ^ 9 b keeps pc ^ rsub) totals them in registers, then lane
      // (rsub, pc) adds column rsub of its chunk - exactly one atomic instruction per wave (N % 256 == 0 is a launch condition), no barrier and
      // no LDS image (the round-3 first form went through LDS behind two workgroup barriers: +19 us per dgrad-fc2 launch).
#pragma unroll
      for (int bf = 1; bf < 8; bf <<= 1)
#pragma unroll
        for (int c = 0; c < 8; ++c) cs[c] += __shfl_xor(cs[c], 9 * bf, 64);
      float s = cs[0];
#pragma unroll
      for (int c = 1; c < 8; ++c) s = rsub == c ? cs[c] : s;
#ifdef VITAMD_EXPERIMENTAL
      if (p.dbg & 16) { if (s == 123.456f) p.colsum[0] = s; } else      // (dbg bit 4, timing only: the butterfly without the atomic)
#endif
      atomicAdd(p.colsum + n0 + wn * 64 + 8 * (pc ^ rsub) + rsub, s);
    }
    if (!has_next) break;
    ti = ti_next;
    VITAMD_WAIT_VM(S::wait(NP - 1) + E);               // the next tile's fill has landed; this tile's stores may still be in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (wm == 1) __builtin_amdgcn_s_barrier();
    seam = true;
  }
#undef VITAMD_WAIT_VM
}

template <int EPI, int MT, bool TAB = false>
int launch_seam(const GemmNtArgs& p, hipStream_t stream, int cus) {
  auto kern = gemm_nt_seam_kernel<EPI, MT, TAB>;
  if (TAB && !p.gelu_tab) return VITAMD_ERR_ARG;
  if (int e = set_lds(kern, 160 * 1024)) return e;
  const int tiles = ((p.M + 32 * MT - 1) / (32 * MT)) * ((p.N + 255) / 256);
  hipLaunchKernelGGL(kern, dim3(tiles > cus ? cus : tiles), dim3(512), 160 * 1024, stream, p);
  return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
}

// launch conditions of the seam kernel (everything else stays on gemm_nt_pp_kernel)
static bool seam_ok(const GemmNtArgs& p) {
  if (p.K % 64 != 0 || p.K < 128 || p.N % 8 != 0 || p.ldo % 8 != 0) return false;
  if ((size_t)p.M * p.K * 2 >= 0xf0000000ull || (size_t)p.N * p.K * 2 >= 0xf0000000ull || (size_t)p.M * p.ldo * 2 >= 0x80000000ull) return false;
  if (p.epi == EPI_DGELU) return p.gelu_dg && p.colsum != nullptr && p.N % 256 == 0 && p.aux != nullptr;
  if (p.epi == EPI_GELU) return p.out2 != nullptr;
  return p.epi == EPI_BIAS_BF16;
}

}  // namespace
