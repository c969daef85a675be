// C[M,N] = A[M,K] . B[N,K]^T  (both operands K-contiguous, bf16, fp32 accumulate) with fused
// epilogues.  This is the one kernel behind every forward Linear and every dgrad of the hot path:
//   forward  y = x W^T + b      : A = x [M,K],  B = W  [N,K]           (reference transformer.py:21,37,39)
//   dgrad    dx = dy W          : A = dy [M,N'], B = W^T [K',N'] (the host keeps a bf16 transposed copy)
//
// gfx950 design: BMxBNx64 tile per workgroup, operands staged by LDS-DMA (global_load_lds 16 B/lane,
// full 128-B lines per row), double-buffered, one barrier per K-tile; LDS rows are 128 B with the
// 16-B chunk index XOR-ed with (row & 7) (applied on the global SOURCE address, LDS image stays
// lane-linear) so the ds_read_b128 fragment reads are bank-conflict-free (tools/lds_banks.py).
// mfma_f32_16x16x32_bf16 with A/B swapped (D[n][m]) so each lane owns 4 consecutive output columns.
//
// Kernels in this file (DESIGN.md section 4 has the measurements):
//   gemm_nt_pipe_kernel     256x256x64 or 320x256x64 (MT = 8 / 10), 8 waves, grouped DMA/ds_read/MFMA issue   <- production (auto, tile=2)
//                           the 320-row form wherever it needs no more rounds x rows of the 256 CUs (prefer_tall): -1.1 ms/step
//   gemm_nt_kernel          plain double-buffered; 128x128 instance serves small problems (auto) / tile=256
//   gemm_nt_persist_kernel  pipe + persistent tile loop with cross-tile prefetch         (tile=6, +0.4 ms/step)
//   gemm_nt_ring_kernel     256x128, BK=32, 3-stage ring, 2 workgroups per CU            (tile=1, slower)
//   gemm_nt_deep_kernel     256x256, BK=32, 3..5-stage ring                              (tile=3..5, much slower on the whole step)
// Epilogues: gemm_epilogue_rows (LDS-transposed, row-major 16-B accesses; production),
//            epilogue_rows_halves (same in 64-KiB of LDS; persistent kernel), gemm_epilogue (direct; small tiles).
#include <type_traits>
#include "common.h"
#include "vitamd_internal.h"

namespace {

constexpr int BK = 64;  // bf16 elements per K-tile = 128 B per LDS row

// Shared epilogue: acc[i][j][r] = C[m][n] with m = m0 + wm*WTM + i*16 + (lane&15),
// n = n0 + wn*WTN + j*16 + 4*(lane>>4) + r  (A/B swapped MFMA: each lane owns 4 consecutive columns).
template <int BN, int WM, int WN, int WTM, int WTN, int MT, int NT, int EPI>
__device__ __forceinline__ void gemm_epilogue(const GemmNtArgs& p, f32x4 (&acc)[MT][NT], int m0, int n0, int wm, int wn,
                                              int lane, int tid, char* smem) {
  constexpr int NW = WM * WN;
  // acc[i][j][r] = C[m][n], m = m0 + wm*WTM + i*16 + (lane&15), n = n0 + wn*WTN + j*16 + 4*(lane>>4) + r
  const int mrow = m0 + wm * WTM + (lane & 15);
  const int ncol = n0 + wn * WTN + 4 * (lane >> 4);
  const int ldo = p.ldo;

  float cs[NT][4];
  if constexpr (EPI == EPI_DGELU) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) cs[j][r] = 0.f;
  }

#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = ncol + j * 16;
    if (n >= p.N) continue;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (EPI == EPI_BIAS_BF16 || EPI == EPI_GELU || EPI == EPI_RESID_F32 || EPI == EPI_PATCH_F32) {
      if (p.bias) {
        const f32x4 b = *(const f32x4*)(p.bias + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) bias4[r] = round_bf16(b[r]);  // autocast casts the bias to bf16
      }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = mrow + i * 16;
      if (m >= p.M) continue;
      f32x4 v = acc[i][j] + bias4;
      if constexpr (EPI == EPI_BIAS_BF16) {
        u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        *(u32x2*)((__bf16*)p.out + (size_t)m * ldo + n) = o;
      } else if constexpr (EPI == EPI_GELU) {
        f32x4 pre, act;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pre[r] = round_bf16(v[r]);
          float dg;
          act[r] = (p.dbg & 1) ? pre[r] : gelu_fwd_grad(pre[r], dg);
          if (p.gelu_dg) pre[r] = dg;                       // `out` carries gelu'(pre) for the backward
        }
        u32x2 o1 = {pack_bf16x2(pre[0], pre[1]), pack_bf16x2(pre[2], pre[3])};
        u32x2 o2 = {pack_bf16x2(act[0], act[1]), pack_bf16x2(act[2], act[3])};
        *(u32x2*)((__bf16*)p.out + (size_t)m * ldo + n) = o1;
        if (!(p.dbg & 2)) *(u32x2*)((__bf16*)p.out2 + (size_t)m * ldo + n) = o2;
      } else if constexpr (EPI == EPI_RESID_F32) {
        const f32x4 res = (p.dbg & 4) ? v : *(const f32x4*)((const float*)p.aux + (size_t)m * ldo + n);
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float y = round_bf16(v[r]);
          if (p.drop_thresh) y = round_bf16(y * dropout_keep((unsigned long long)(p.row0 + m) * p.N + n + r, p.drop_seed_lo, p.drop_seed_hi, p.drop_thresh, p.drop_scale));
          o[r] = res[r] + y;
        }
        *(f32x4*)((float*)p.out + (size_t)m * ldo + n) = o;
      } else if constexpr (EPI == EPI_DGELU) {
        const u32x2 pz = *(const u32x2*)((const __bf16*)p.aux + (size_t)m * ldo + n);
        const float pre[4] = {bf16lo(pz[0]), bf16hi(pz[0]), bf16lo(pz[1]), bf16hi(pz[1])};
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o[r] = round_bf16(round_bf16(v[r]) * (p.gelu_dg ? pre[r] : gelu_grad(pre[r])));
          cs[j][r] += o[r];
        }
        u32x2 ov = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
        *(u32x2*)((__bf16*)p.out + (size_t)m * ldo + n) = ov;
      } else if constexpr (EPI == EPI_PATCH_F32) {
        // row m = b * n_patches + pidx  ->  token row b * seq + extra + pidx ; + pos_emb[pidx]
        const int b = m / p.n_patches, pidx = m - b * p.n_patches;
        const f32x4 pos = *(const f32x4*)((const float*)p.aux + (size_t)pidx * ldo + n);
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = round_bf16(v[r]) + pos[r];
        *(f32x4*)((float*)p.out + ((size_t)b * p.seq + p.extra + pidx) * ldo + n) = o;
      } else if constexpr (EPI == EPI_F32) {
        *(f32x4*)((float*)p.out + (size_t)m * ldo + n) = v;
      }
    }
  }

  if constexpr (EPI == EPI_DGELU) {
    // column sums of the stored tile (= bias gradient of the producing Linear), one shaped
    // 256-B atomic wave-instruction per 64 columns
    if (p.colsum) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float s = cs[j][r];
          s += __shfl_xor(s, 1, 64);
          s += __shfl_xor(s, 2, 64);
          s += __shfl_xor(s, 4, 64);
          s += __shfl_xor(s, 8, 64);
          cs[j][r] = s;
        }
      __syncthreads();  // main-loop LDS reads finished everywhere
      float* red = (float*)smem;  // [WM][BN]
      if ((lane & 15) == 0) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[wm * BN + wn * WTN + j * 16 + 4 * (lane >> 4) + r] = cs[j][r];
      }
      __syncthreads();
      for (int c = tid; c < BN; c += NW * 64) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) s += red[w * BN + c];
        if (n0 + c < p.N) atomicAdd(p.colsum + n0 + c, s);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Row-major epilogue for the 128x64 wave tile (8 waves, 256x256 block).  In the accumulator layout
// a lane owns 4 columns of 16 different rows, so direct stores are 8-B pieces scattered over 16 rows:
// the epilogue was store/load-ISSUE bound (tools/ablate_epilogue.py: the second GELU output cost
// 160 us, the residual read 148 us, the erf math 7 us).  Here each wave transposes its tile through
// a private 16-KiB LDS image ([128 rows][64 bf16], 16-B chunk index XOR (row&7)), after which a lane
// owns 8 consecutive columns of one row: every global access is 16 B per lane and a wave-instruction
// covers whole 128-B (bf16) / 256-B (fp32) row segments of 8 rows.
// output stores are non-temporal (keeps the 32 MB-per-round output burst from evicting operand
// panels out of the 8 x 4 MiB L2s: -5..7 % on the K = 3072 shapes); p.dbg bit 3 turns that off (A/B knob)
#define ST16(ptr, val)                                              \
  do {                                                              \
    if (p.dbg & 8) *(ptr) = (val);                                  \
    else __builtin_nontemporal_store((val), (ptr));                                            \
  } while (0)

template <int EPI, int MT = 8>
__device__ __forceinline__ void gemm_epilogue_rows(const GemmNtArgs& p, f32x4 (&acc)[MT][4], int m0, int n0, int wm, int wn,
                                                   int lane, int tid, int wave, char* smem) {
  constexpr int BN = 256;
  __syncthreads();                       // every wave is done reading the operand buffers
  char* tile = smem + wave * (MT * 2048);   // wave-private image: [16*MT rows][64 bf16]
  const int mloc = lane & 15, g = lane >> 4;
  const int ncol_acc = n0 + wn * 64 + 4 * g;
  // ---- 1. bias (+ bf16 rounding of the Linear output) in the accumulator layout, pack, write to LDS
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (EPI != EPI_DGELU) {
      const int n = ncol_acc + j * 16;
      if (p.bias && n < p.N) {
        const f32x4 b = *(const f32x4*)(p.bias + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) bias4[r] = round_bf16(b[r]);
      }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const f32x4 v = acc[i][j] + bias4;
      const int row = 16 * i + mloc;
      const int chunk = (2 * j + (g >> 1)) ^ (row & 7);
      u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
      *(u32x2*)(tile + row * 128 + (chunk << 4) + (g & 1) * 8) = o;
    }
  }
  // ---- 2. read back row-major: lane -> row (lane>>3) + 8*it, physical chunk lane&7
  const int rsub = lane >> 3, pc = lane & 7;
  const int ldo = p.ldo;
  float cs[8];
  if constexpr (EPI == EPI_DGELU) {
#pragma unroll
    for (int c = 0; c < 8; ++c) cs[c] = 0.f;
  }
  // the logical chunk of (row, pc) is pc ^ (row&7) and row&7 == rsub for every it (rows step by 8)
  const int n = n0 + wn * 64 + 8 * (pc ^ rsub);
  const bool ncol_ok = n < p.N;           // N % 8 == 0 is required by the row epilogue
  const int nc = ncol_ok ? n : 0;
  const int mbase = m0 + wm * (16 * MT) + rsub;
  // ---- 2a. the accumulators are dead now: issue EVERY auxiliary load of the tile up front (clamped
  // rows, unconditional) so their latency overlaps the LDS round trip instead of serialising per row
  u32x4 auxb[EPI == EPI_DGELU ? 2 * MT : 1];
  f32x4 auxf[(EPI == EPI_RESID_F32 || EPI == EPI_PATCH_F32) ? 4 * MT : 1];
  if constexpr (EPI == EPI_DGELU) {
#pragma unroll
    for (int it = 0; it < 2 * MT; ++it) {
      const int m = min(mbase + 8 * it, p.M - 1);
      auxb[it] = *(const u32x4*)((const __bf16*)p.aux + (size_t)m * ldo + nc);
    }
  } else if constexpr (EPI == EPI_RESID_F32) {
#pragma unroll
    for (int it = 0; it < 2 * MT; ++it) {
      const int m = min(mbase + 8 * it, p.M - 1);
      const float* rp = (const float*)p.aux + (size_t)m * ldo + nc;
      auxf[2 * it] = *(const f32x4*)rp;
      auxf[2 * it + 1] = *(const f32x4*)(rp + 4);
    }
  } else if constexpr (EPI == EPI_PATCH_F32) {
#pragma unroll
    for (int it = 0; it < 2 * MT; ++it) {
      const int m = min(mbase + 8 * it, p.M - 1);
      const float* pp = (const float*)p.aux + (size_t)(m % p.n_patches) * ldo + nc;
      auxf[2 * it] = *(const f32x4*)pp;
      auxf[2 * it + 1] = *(const f32x4*)(pp + 4);
    }
  }
#pragma unroll
  for (int it = 0; it < 2 * MT; ++it) {
    const int rloc = rsub + 8 * it;
    const int m_true = mbase + 8 * it;
    const int m = (p.dbg & 4) ? m_true % 640 : m_true;          // dbg bit 2: timing-only, every tile stores to the same few (L2-resident) rows
    const u32x4 v = *(const u32x4*)(tile + rloc * 128 + pc * 16);
    const bool ok = m_true < p.M && ncol_ok && !(p.dbg & 0x10000);   // dbg bit 16: timing-only, no output stores
    if constexpr (EPI == EPI_BIAS_BF16) {
      if (ok) ST16((u32x4*)((__bf16*)p.out + (size_t)m * ldo + n), v);
    } else if constexpr (EPI == EPI_GELU) {
      u32x4 a;
      u32x4 d = v;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float dlo, dhi;
        a[c] = pack_bf16x2(gelu_fwd_grad(bf16lo(v[c]), dlo), gelu_fwd_grad(bf16hi(v[c]), dhi));
        if (p.gelu_dg) d[c] = pack_bf16x2(dlo, dhi);       // `out` carries gelu'(pre) for the backward
      }
      if (ok) {
        ST16((u32x4*)((__bf16*)p.out + (size_t)m * ldo + n), d);
        ST16((u32x4*)((__bf16*)p.out2 + (size_t)m * ldo + n), a);
      }
    } else if constexpr (EPI == EPI_RESID_F32 || EPI == EPI_PATCH_F32) {
      const f32x4 r0 = auxf[2 * it], r1 = auxf[2 * it + 1];
      float y8[8] = {bf16lo(v[0]), bf16hi(v[0]), bf16lo(v[1]), bf16hi(v[1]), bf16lo(v[2]), bf16hi(v[2]), bf16lo(v[3]), bf16hi(v[3])};
      if constexpr (EPI == EPI_RESID_F32) {
        if (p.drop_thresh) {
          const unsigned long long base = (unsigned long long)(p.row0 + m) * p.N + n;
#pragma unroll
          for (int c = 0; c < 8; ++c) y8[c] = round_bf16(y8[c] * dropout_keep(base + c, p.drop_seed_lo, p.drop_seed_hi, p.drop_thresh, p.drop_scale));
        }
      }
      f32x4 o0 = {r0[0] + y8[0], r0[1] + y8[1], r0[2] + y8[2], r0[3] + y8[3]};
      f32x4 o1 = {r1[0] + y8[4], r1[1] + y8[5], r1[2] + y8[6], r1[3] + y8[7]};
      size_t orow = (size_t)m;
      if constexpr (EPI == EPI_PATCH_F32) {
        const int b = m / p.n_patches, pidx = m - b * p.n_patches;
        orow = (size_t)b * p.seq + p.extra + pidx;
      }
      if (ok) {
        float* op = (float*)p.out + orow * ldo + n;
        ST16((f32x4*)op, o0);
        ST16((f32x4*)(op + 4), o1);
      }
    } else if constexpr (EPI == EPI_DGELU) {
      const u32x4 pz = auxb[it];
      u32x4 o;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bool raw = p.gelu_dg;         // aux already holds gelu'(pre)
        const float lo = round_bf16(bf16lo(v[c]) * (raw ? bf16lo(pz[c]) : gelu_grad(bf16lo(pz[c]))));
        const float hi = round_bf16(bf16hi(v[c]) * (raw ? bf16hi(pz[c]) : gelu_grad(bf16hi(pz[c]))));
        if (ok) { cs[2 * c] += lo; cs[2 * c + 1] += hi; }
        o[c] = pack_bf16x2(lo, hi);
      }
      if (ok) ST16((u32x4*)((__bf16*)p.out + (size_t)m * ldo + n), o);
    }
  }
  if constexpr (EPI == EPI_DGELU) {
    if (p.colsum) {
      // lanes with equal (pc, rsub) parity... every lane's 8 columns are fixed: reduce over the 8 lanes that
      // share pc ^ rsub?  No: column block = pc ^ rsub, so lanes (pc, rsub) and (pc', rsub') share columns
      // iff pc^rsub == pc'^rsub'.  Combine through LDS: red[wave][col] += with shaped accesses.
      __syncthreads();                   // all waves finished reading their images
      float* red = (float*)smem;         // [8 waves][8 rsub][64 cols] floats = 16 KiB
      float* mine = red + (wave * 8 + rsub) * 64 + 8 * (pc ^ rsub);
#pragma unroll
      for (int c = 0; c < 8; ++c) mine[c] = cs[c];
      __syncthreads();
      // column c of the block: waves with wn == c/64 (two of them: wm = 0,1), 8 rsub rows each
      for (int c = tid; c < BN; c += 512) {
        const int wnn = c >> 6, cc = c & 63;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 2; ++w)
#pragma unroll
          for (int r = 0; r < 8; ++r) s += red[((w * 4 + wnn) * 8 + r) * 64 + cc];
        if (n0 + c < p.N) atomicAdd(p.colsum + n0 + c, s);
      }
    }
  }
}

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
// Main kernel: 256x128 output tile, 4 waves (2x2, wave tile 128x64), BK = 32, 3-stage LDS ring
// (72 KiB) so TWO workgroups share a CU.  Why this shape on gfx950: with one 8-wave / 128-KiB
// workgroup per CU every workgroup reaches its (HBM-bound) epilogue at the same time and the MFMA
// pipes idle meanwhile; two independent 4-wave workgroups per CU de-phase naturally — one's
// epilogue stores / GELU VALU work / barrier and LDS-latency stalls hide under the other's MFMAs
// (each SIMD hosts one wave of each).  Tiles are streamed by LDS-DMA two K-steps ahead behind a
// COUNTED s_waitcnt vmcnt (never 0 in the loop) and a raw s_barrier.  LDS rows are 64 B (4 chunks);
// chunk index XOR g[(row>>2)&3], g = {0,2,3,1}, applied on the source address: conflict-free
// ds_read_b128 (tools/lds_banks.py).
template <int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(WM * WN * 64) void gemm_nt_ring_kernel(const GemmNtArgs p) {
  constexpr int NW = WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MT = WTM / 16, NT = WTN / 16;
  constexpr int KS = 32;                         // K-step
  constexpr int PIECES = (BM + BN) / 16;         // 1-KiB pieces: 16 rows x 64 B
  constexpr int PPW = PIECES / NW;
  constexpr int A_ITERS = BM / 16 / NW;          // pieces i < A_ITERS of every wave are A rows
  static_assert(PIECES % NW == 0 && (BM / 16) % NW == 0, "piece split");
  constexpr int STAGES = 3;
  constexpr int STAGE_BYTES = (BM + BN) * 64;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int K = p.K;

  const __amdgpu_buffer_rsrc_t rsrcA = make_rsrc(p.A, (size_t)p.M * K * 2);
  const __amdgpu_buffer_rsrc_t rsrcB = make_rsrc(p.B, (size_t)p.N * K * 2);

  // piece q = i*NW + wave ; rows 16q .. 16q+15 of the stacked [A rows ; B rows] stage image
  unsigned voff[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int row = (i * NW + wave) * 16 + (lane >> 2);
    const int g4 = (0x1320 >> (((row >> 2) & 3) * 4)) & 3;     // g = {0,2,3,1}
    const int logical = (lane & 3) ^ g4;
    const int grow = (i < A_ITERS) ? min(m0 + row, p.M - 1) : min(n0 + row - BM, p.N - 1);  // clamp: never stored
    voff[i] = (unsigned)grow * (unsigned)(K * 2) + logical * 16;
  }
#define STAGE(kt_, slot_)                                                                                   \
  do {                                                                                                      \
    char* base_ = smem + (slot_) * STAGE_BYTES + wave * 1024;                                               \
    const int soff_ = (kt_) * (KS * 2);                                                                     \
    _Pragma("unroll") for (int i = 0; i < PPW; ++i) {                                                       \
      buf_glds16(i < A_ITERS ? rsrcA : rsrcB, base_ + i * NW * 1024, voff[i], soff_);                        \
    }                                                                                                       \
  } while (0)

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment read: row = 16t + (lane&15), chunk = lane>>4, swizzled with g[(row>>2)&3] (tile offsets are multiples of 16 rows)
  const int fr = lane & 15;
  const int frag_off = fr * 64 + ((((lane >> 4) ^ ((0x1320 >> (((fr >> 2) & 3) * 4)) & 3)) & 3) << 4);
  const int a_off = wm * WTM * 64 + frag_off;
  const int b_off = BM * 64 + wn * WTN * 64 + frag_off;

  const int nkt = K / KS;
  STAGE(0, 0);
  if (nkt > 1) STAGE(1, 1);
  int slot = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    // tile kt has landed once at most the PPW loads of tile kt+1 are still outstanding
    if (kt + 1 < nkt) {
      static_assert(PPW == 6 || PPW == 4 || PPW == 8, "add a vmcnt literal for this piece count");
      if constexpr (PPW == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      if constexpr (PPW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      if constexpr (PPW == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();   // publishes tile kt; everyone is done reading slot (kt+2)%3 = slot of tile kt-1
    asm volatile("" ::: "memory");
    if (kt + 2 < nkt) STAGE(kt + 2, slot == 0 ? 2 : slot - 1);
    const char* buf = smem + slot * STAGE_BYTES;
    bf16x8 af[MT], bfr[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bfr[j] = *(const bf16x8*)(buf + b_off + j * 16 * 64);
#pragma unroll
    for (int i = 0; i < MT; ++i) af[i] = *(const bf16x8*)(buf + a_off + i * 16 * 64);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    slot = slot == 2 ? 0 : slot + 1;
  }
#undef STAGE
  gemm_epilogue<BN, WM, WN, WTM, WTN, MT, NT, EPI>(p, acc, m0, n0, wm, wn, lane, tid, smem);
}

template <int BM, int BN, int WM, int WN, int EPI>
int launch_ring(const GemmNtArgs& p, hipStream_t stream) {
  constexpr int lds = 3 * (BM + BN) * 64;
  auto kern = gemm_nt_ring_kernel<BM, BN, WM, WN, EPI>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return VITAMD_ERR_LAUNCH;
    attr_done = true;
  }
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(WM * WN * 64), lds, stream, p);
  return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
}

// ---------------------------------------------------------------------------------------------
// Main kernel: 256x256 tile, 8 waves (2x4, wave tile 128x64), BK = 64, two LDS buffers (128 KiB).
// The K-tile is cut into 4 groups of 16 (20) MFMAs (half the A fragments x four B fragments x one 32-deep
// k-substep; 8 groups of 8 originally).  Each group FIRST issues its share of the next tile's LDS-DMA (one 1-KiB piece) and
// the ds_reads of the NEXT group's fragments, THEN runs its 8 MFMAs, so VMEM issue, LDS latency
// and matrix work overlap inside one wave instead of arriving in bursts behind the barrier
// (PMC on the burst form: MFMA pipe 38 % busy, waves 49 % issue-stalled; profiles/r01).
// MT = 16-row MFMA tiles per wave along M: 8 -> 256x256 tile; 10 -> 320x256 (wave tile 160x64, 160 accumulator registers), used for
// N = 768 outputs where 256-row tiles need 2.31 rounds of the 256 CUs and 320-row tiles 1.85 (dispatch_tile picks by rounds x rows)
template <int EPI, int ABL = 0, int MT = 8>   // ABL: timing-only ablations (1 = no MFMA, 2 = no LDS-DMA, 3 = no ds_read); results are garbage
__global__ __launch_bounds__(512) void gemm_nt_pipe_kernel(const GemmNtArgs p) {
  constexpr int BM = 32 * MT, BN = 256, WM = 2, WN = 4, NW = 8;
  constexpr int WTM = 16 * MT, WTN = 64, NT = 4;
  constexpr int APW = MT / 2;                  // A pieces per wave per K-tile
  constexpr int PPW = APW + 4;                 // 1-KiB pieces (8 rows x 128 B) per wave per K-tile
  constexpr int BUF_BYTES = (BM + BN) * 128;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  int tm, tn;
  tile_coords(tile, tiles_m, tiles_n, tiles_n >= 6 && !(p.dbg & 32), tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int K = p.K;

  const __amdgpu_buffer_rsrc_t rsrcA = make_rsrc(p.A, (size_t)p.M * K * 2);
  const __amdgpu_buffer_rsrc_t rsrcB = make_rsrc(p.B, (size_t)p.N * K * 2);
  // piece q = i*8 + wave: i < 4 -> A rows, i >= 4 -> B rows (stacked [A;B] stage image, 128-B rows)
  unsigned voff[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int row = (i * NW + wave) * 8 + (lane >> 3);
    const int logical = (lane & 7) ^ (row & 7);
    const int grow = (i < APW) ? min(m0 + row, p.M - 1) : min(n0 + row - BM, p.N - 1);
    voff[i] = (unsigned)grow * (unsigned)(K * 2) + logical * 16;
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int frag_off = (lane & 15) * 128 + ((((lane >> 4) ^ (lane & 7)) & 7) << 4);
  const int a_off = wm * WTM * 128 + frag_off;
  const int b_off = BM * 128 + wn * WTN * 128 + frag_off;
  const int nkt = (p.dbg & 0x20000) ? 1 : K / 64;   // dbg bit 17: timing-only, one K-tile (epilogue cost in isolation)

  // Phase stagger: the odd workgroups of the FIRST round start ~8 us late, so that from then on the
  // two halves of the chip reach their HBM-bound epilogues at different times (successor workgroups
  // inherit the offset).  Measured -3..-7 % per GEMM (tools/ablate_epilogue.py).  dbg bits 8..15
  // override the delay in ~1 us units (255 = off).
  {
    const int req = (p.dbg >> 8) & 0xff;
    const int st = req == 255 ? 0 : (req ? req : 8);
    int P = (p.dbg >> 20) & 0xf;                       // number of phases (timing knob; default 2)
    if (P == 0) P = 2;
    if (st && gridDim.x > 256 && blockIdx.x < 256) {
      const int ph = (p.dbg & (1 << 24)) ? ((blockIdx.x >> 3) % P) : (blockIdx.x % P);
      for (int i = 0; i < st * ph; ++i) __builtin_amdgcn_s_sleep(32);
    }
  }
  {  // prologue: whole tile 0
    char* base = smem + wave * 1024;
#pragma unroll
    for (int i = 0; i < PPW; ++i) buf_glds16(i < APW ? rsrcA : rsrcB, base + i * NW * 1024, voff[i], 0);
  }
  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();       // tile kt visible; buffer cur^1 free
    asm volatile("" ::: "memory");
    const char* buf = smem + cur * BUF_BYTES;
    char* nbase = smem + (cur ^ 1) * BUF_BYTES + wave * 1024;
    const bool more = kt + 1 < nkt;
    const int soff = (kt + 1) * 128;

    // A fragments per group.  MT/2: four groups of 16 (20) MFMAs per K-tile, each ~256 (320) MFMA cycles long, which covers the
    // latency of the next group's ds_reads with room to spare; the original 8 (10) groups of 8 MFMAs (ABL 7) were 0.2 ms/step slower
    constexpr int GA = (ABL == 7) ? 2 : MT / 2;
    constexpr int GPK = APW * 2 / GA;               // groups per 32-deep k-substep   (APW = MT / 2)
    constexpr int NG = 2 * GPK;                     // groups per K-tile
    bf16x8 bq[2][NT], aq[2][GA];
    u32x4 stg[ABL == 5 ? PPW : 1];    // ABL 5: stage the next tile through VGPRs + ds_write instead of LDS-DMA (correct results)
#pragma unroll
    for (int j = 0; j < NT; ++j) if (ABL != 3 || kt == 0) bq[0][j] = *(const bf16x8*)(buf + b_off + j * 2048);
#pragma unroll
    for (int i = 0; i < GA; ++i) if (ABL != 3 || kt == 0) aq[0][i] = *(const bf16x8*)(buf + a_off + i * 2048);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int ks = g / GPK, pr = g % GPK;
      if (more && ABL != 2) {
#pragma unroll
        for (int q = g * PPW / NG; q < (g + 1) * PPW / NG; ++q) {     // this group's share of the next tile's pieces
          if constexpr (ABL == 5) stg[q] = buf_load16(q < APW ? rsrcA : rsrcB, voff[q], soff);
          else buf_glds16(q < APW ? rsrcA : rsrcB, nbase + q * NW * 1024, voff[q], soff);
        }
      }
      if (g < NG - 1) {
        const int ks2 = (g + 1) / GPK, pr2 = (g + 1) % GPK;
#pragma unroll
        for (int i = 0; i < GA; ++i) if (ABL != 3 || kt == 0) aq[(g + 1) & 1][i] = *(const bf16x8*)(buf + ((a_off + (GA * pr2 + i) * 2048) ^ (ks2 * 64)));
      }
      if (g == (NG > 4 ? 1 : 0)) {
#pragma unroll
        for (int j = 0; j < NT; ++j) if (ABL != 3 || kt == 0) bq[1][j] = *(const bf16x8*)(buf + ((b_off + j * 2048) ^ 64));
      }
      if constexpr (ABL == 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < GA; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          if constexpr (ABL == 1) { asm volatile("" ::"v"(bq[ks][j]), "v"(aq[g & 1][i])); }
          else acc[GA * pr + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[ks][j], aq[g & 1][i], acc[GA * pr + i][j], 0, 0, 0);
      if constexpr (ABL == 4) __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (ABL == 5) {
      if (more) {
#pragma unroll
        for (int g = 0; g < PPW; ++g) *(u32x4*)(nbase + g * NW * 1024 + lane * 16) = stg[g];
      }
    }
  }
  if constexpr (EPI == EPI_F32) gemm_epilogue<BN, WM, WN, WTM, WTN, MT, NT, EPI>(p, acc, m0, n0, wm, wn, lane, tid, smem);
  else if (p.N % 8 == 0 && p.ldo % 8 == 0) gemm_epilogue_rows<EPI, MT>(p, acc, m0, n0, wm, wn, lane, tid, wave, smem);
  else gemm_epilogue<BN, WM, WN, WTM, WTN, MT, NT, EPI>(p, acc, m0, n0, wm, wn, lane, tid, smem);
}

template <int EPI, int ABL = 0, int MT = 8>
int launch_pipe(const GemmNtArgs& p, hipStream_t stream) {
  constexpr int BM = 32 * MT;
  constexpr int lds = (2 * (BM + 256) * 128 > 8 * MT * 2048) ? 2 * (BM + 256) * 128 : 8 * MT * 2048;   // operand stages / epilogue images
  auto kern = gemm_nt_pipe_kernel<EPI, ABL, MT>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return VITAMD_ERR_LAUNCH;
    attr_done = true;
  }
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + 255) / 256);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), lds, stream, p);
  return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
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
constexpr int PP_STAGGER_GROUPS = 1;   // 1 = no stagger

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

template <int EPI, int MT, int LA, int LB>
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
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  int tm, tn;
  tile_coords(tile, tiles_m, tiles_n, tiles_n >= 6, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int K = p.K;
  const int nkt = K / 64;

  const srd_t srdA = make_srd(p.A, (size_t)p.M * K * 2);
  const srd_t srdB = make_srd(p.B, (size_t)p.N * K * 2);
  // this wave's piece of A-part j: LDS rows 8*wave + (lane>>3) of the part = rows 32 j + (lr&31) of wave row lr>>5;
  // its piece q of the B block: rows 64 q + 8*wave + (lane>>3).  16-B chunk lane&7, XOR (row&7) on the source side.
  unsigned voffA[NP], voffB[4];
  {
    const int lr = 8 * wave + (lane >> 3);
    const unsigned chunk = (unsigned)(((lane & 7) ^ (lr & 7)) * 16);
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int ga = min(((p.dbg & 0x40000) ? (m0 & 0x3ff) : m0) + (lr >> 5) * (16 * MT) + j * 32 + (lr & 31), p.M - 1);     // clamp: rows past M are never stored (dbg bit 18, timing only: every tile loads one of a few L2-resident panels)
      voffA[j] = (unsigned)ga * (unsigned)(K * 2) + chunk;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int gb = min(((p.dbg & 0x40000) ? 0 : n0) + 64 * q + lr, p.N - 1);
      voffB[q] = (unsigned)gb * (unsigned)(K * 2) + chunk;
    }
  }
  const unsigned lds0 = lds_addr(smem) + wave * 1024;
  constexpr unsigned OOB = 0x80000000u;
  auto request_a = [&](int kt, int j) {
    const bool live = kt >= 0 && kt < nkt;
    asm_glds16(srdA, lds0 + (kt & 1) * BUFB + j * PART, live ? voffA[j] : OOB, live ? (unsigned)kt * 128u : 0u);
  };
  auto request_b = [&](int kt, int q) {
    const bool live = kt >= 0 && kt < nkt;
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
  // First-round stagger: all workgroups of a launch start together and would reach their (HBM-write-bound) epilogues together,
  // with the matrix pipes idle meanwhile; delaying workgroup group g = (blockIdx / 8) % P of the FIRST round by g/P of a tile
  // period puts the groups' epilogues at different times for the rest of the launch (successors inherit the offset).
  // dbg bits 20-23 = P (0 = default), bits 8-15 = delay unit per group in ~us (0 = default, 255 = off).
  {
    const int req = (p.dbg >> 8) & 0xff;
    int P = (p.dbg >> 20) & 0xf;
    if (P == 0) P = PP_STAGGER_GROUPS;
    const int unit = req == 255 ? 0 : (req ? req : max(1, (2 * nkt + 6) / P));      // tile period ~ (2 us per K-tile + epilogue) / P
    if (unit && gridDim.x > 256 && blockIdx.x < 256) {
      const int g = (blockIdx.x >> 3) % P;
      for (int i = 0; i < unit * g; ++i) __builtin_amdgcn_s_sleep(32);
    }
  }
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
}

template <int EPI, int MT, int LA, int LB>
int launch_pp(const GemmNtArgs& p, hipStream_t stream) {
  constexpr int BM = 32 * MT;
  constexpr int ops_b = 2 * ((MT / 2) * 8192 + 32768), epi_b = 8 * MT * 2048;     // operand buffers / epilogue images
  constexpr int lds = ops_b > epi_b ? ops_b : epi_b;
  auto kern = gemm_nt_pp_kernel<EPI, MT, LA, LB>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return VITAMD_ERR_LAUNCH;
    attr_done = true;
  }
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + 255) / 256);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), lds, stream, p);
  return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
}

// ---------------------------------------------------------------------------------------------
// Deep-prefetch variant of the 256x256 kernel: BK = 32 stages of 32 KiB in an NS-deep LDS ring
// (4 -> 128 KiB).  Ablation of the 2-buffer kernel (tools/ablate_gemm.py, profiles/r01) showed the
// main loop is bound by operand-fetch LATENCY, not MFMA rate: with one K-tile of prefetch distance
// a tile's slowest 1-KiB piece (an L2 miss, ~2 us under load) gates the whole workgroup every
// K-step.  Here tile kt+NS-1 is issued while tile kt is computed (NS-1 stages = 96 KiB in flight
// per CU), behind a COUNTED s_waitcnt vmcnt and one raw s_barrier per 32-deep step.
template <int EPI, int NS>
__global__ __launch_bounds__(512) void gemm_nt_deep_kernel(const GemmNtArgs p) {
  constexpr int BM = 256, BN = 256, WM = 2, WN = 4, NW = 8;
  constexpr int WTM = 128, WTN = 64, MT = 8, NT = 4;
  constexpr int PPW = 4;                       // 1-KiB pieces (16 rows x 64 B) per wave per stage
  constexpr int STAGE_BYTES = (BM + BN) * 64;  // 32 KiB

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int K = p.K;

  const __amdgpu_buffer_rsrc_t rsrcA = make_rsrc(p.A, (size_t)p.M * K * 2);
  const __amdgpu_buffer_rsrc_t rsrcB = make_rsrc(p.B, (size_t)p.N * K * 2);
  // piece q = i*8 + wave: i < 2 -> A rows 16q.., i >= 2 -> B rows (stacked [A;B] image, 64-B rows)
  unsigned voff[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int row = (i * NW + wave) * 16 + (lane >> 2);
    const int g4 = (0x1320 >> (((row >> 2) & 3) * 4)) & 3;
    const int logical = (lane & 3) ^ g4;
    const int grow = (i < 2) ? min(m0 + row, p.M - 1) : min(n0 + row - BM, p.N - 1);
    voff[i] = (unsigned)grow * (unsigned)(K * 2) + logical * 16;
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15;
  const int frag_off = fr * 64 + ((((lane >> 4) ^ ((0x1320 >> (((fr >> 2) & 3) * 4)) & 3)) & 3) << 4);
  const int a_off = wm * WTM * 64 + frag_off;
  const int b_off = BM * 64 + wn * WTN * 64 + frag_off;
  const int nkt = K / 32;

#pragma unroll
  for (int t = 0; t < NS - 1; ++t) {           // prologue: tiles 0 .. NS-2 (always PPW loads each so
    char* base = smem + t * STAGE_BYTES + wave * 1024;   // the vmcnt arithmetic below is uniform; past-the-end tiles read clamped, unused bytes)
    const int kk = min(t, nkt - 1);
#pragma unroll
    for (int i = 0; i < PPW; ++i) buf_glds16(i < 2 ? rsrcA : rsrcB, base + i * NW * 1024, voff[i], kk * 64);
  }
  int slot = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    // tiles kt+1 .. kt+NS-2 may stay in flight: (NS-2)*PPW outstanding pieces
    if constexpr (NS == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if constexpr (NS == 5) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    if constexpr (NS == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // tile kt visible to all; slot of tile kt-1 is free
    asm volatile("" ::: "memory");
    const char* buf = smem + slot * STAGE_BYTES;
    const int nslot = slot == 0 ? NS - 1 : slot - 1;                  // slot of tile kt-1 == slot of tile kt+NS-1
    char* nbase = smem + nslot * STAGE_BYTES + wave * 1024;
    const int soff = min(kt + NS - 1, nkt - 1) * 64;                  // tail: harmless re-read of the last tile

    bf16x8 bq[NT], aq[2][2];
#pragma unroll
    for (int j = 0; j < NT; ++j) bq[j] = *(const bf16x8*)(buf + b_off + j * 1024);
#pragma unroll
    for (int i = 0; i < 2; ++i) aq[0][i] = *(const bf16x8*)(buf + a_off + i * 1024);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      buf_glds16(g < 2 ? rsrcA : rsrcB, nbase + g * NW * 1024, voff[g], soff);
      if (g < 3) {
#pragma unroll
        for (int i = 0; i < 2; ++i) aq[(g + 1) & 1][i] = *(const bf16x8*)(buf + a_off + (2 * (g + 1) + i) * 1024);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[2 * g + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[j], aq[g & 1][i], acc[2 * g + i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    slot = slot == NS - 1 ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the tail re-reads before LDS is reused / the wave ends
  __builtin_amdgcn_s_barrier();
  gemm_epilogue<BN, WM, WN, WTM, WTN, MT, NT, EPI>(p, acc, m0, n0, wm, wn, lane, tid, smem);
}

template <int EPI, int NS>
int launch_deep(const GemmNtArgs& p, hipStream_t stream) {
  constexpr int lds = NS * 512 * 64;
  auto kern = gemm_nt_deep_kernel<EPI, NS>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return VITAMD_ERR_LAUNCH;
    attr_done = true;
  }
  const int tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), lds, stream, p);
  return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
}

// ---------------------------------------------------------------------------------------------
// Persistent form of the pipe kernel: one workgroup per CU walks a strided list of tiles.  The
// LDS-DMA of the NEXT tile's first K-tile is issued during the current tile's last K-step, so the
// per-tile prologue (launch gap, descriptor setup, first DMA round trip: ~4 of ~27 us at K = 768)
// disappears, and the epilogue's global stores are still draining while the next main loop starts
// (its first wait is a COUNTED vmcnt that skips the younger stores).  The epilogue transposes through
// the ONE operand buffer that is free at that point (64 KiB: two passes of 64 rows per wave).
template <int EPI>
__device__ __forceinline__ void epilogue_rows_halves(const GemmNtArgs& p, f32x4 (&acc)[8][4], int m0, int n0, int wm, int wn,
                                                     int lane, int tid, int wave, char* scratch /* 64 KiB, free */) {
  constexpr int BN = 256;
  char* tile = scratch + wave * 8192;     // wave-private image: 64 rows x 128 B
  const int mloc = lane & 15, g = lane >> 4;
  const int ncol_acc = n0 + wn * 64 + 4 * g;
  const int rsub = lane >> 3, pc = lane & 7;
  const int ldo = p.ldo;
  const int n = n0 + wn * 64 + 8 * (pc ^ rsub);
  const bool ncol_ok = n < p.N;
  const int nc = ncol_ok ? n : 0;
  float cs[8];
  if constexpr (EPI == EPI_DGELU) {
#pragma unroll
    for (int c = 0; c < 8; ++c) cs[c] = 0.f;
  }
  f32x4 bias4[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    bias4[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (EPI != EPI_DGELU) {
      const int nb = ncol_acc + j * 16;
      if (p.bias && nb < p.N) {
        const f32x4 b = *(const f32x4*)(p.bias + nb);
#pragma unroll
        for (int r = 0; r < 4; ++r) bias4[j][r] = round_bf16(b[r]);
      }
    }
  }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int mbase = m0 + wm * 128 + 64 * half + rsub;
    // auxiliary loads of this half first (latency overlaps the LDS round trip)
    u32x4 auxb[EPI == EPI_DGELU ? 8 : 1];
    f32x4 auxf[(EPI == EPI_RESID_F32 || EPI == EPI_PATCH_F32) ? 16 : 1];
    if constexpr (EPI == EPI_DGELU) {
#pragma unroll
      for (int it = 0; it < 8; ++it) auxb[it] = *(const u32x4*)((const __bf16*)p.aux + (size_t)min(mbase + 8 * it, p.M - 1) * ldo + nc);
    } else if constexpr (EPI == EPI_RESID_F32) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const float* rp = (const float*)p.aux + (size_t)min(mbase + 8 * it, p.M - 1) * ldo + nc;
        auxf[2 * it] = *(const f32x4*)rp;
        auxf[2 * it + 1] = *(const f32x4*)(rp + 4);
      }
    } else if constexpr (EPI == EPI_PATCH_F32) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const float* pp = (const float*)p.aux + (size_t)(min(mbase + 8 * it, p.M - 1) % p.n_patches) * ldo + nc;
        auxf[2 * it] = *(const f32x4*)pp;
        auxf[2 * it + 1] = *(const f32x4*)(pp + 4);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 v = acc[4 * half + i][j] + bias4[j];
        const int row = 16 * i + mloc;
        const int chunk = (2 * j + (g >> 1)) ^ (row & 7);
        u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        *(u32x2*)(tile + row * 128 + (chunk << 4) + (g & 1) * 8) = o;
      }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int rloc = rsub + 8 * it;
      const int m = mbase + 8 * it;
      const u32x4 v = *(const u32x4*)(tile + rloc * 128 + pc * 16);
      const bool ok = m < p.M && ncol_ok;
      if constexpr (EPI == EPI_BIAS_BF16) {
        if (ok) ST16((u32x4*)((__bf16*)p.out + (size_t)m * ldo + n), v);
      } else if constexpr (EPI == EPI_GELU) {
        u32x4 a;
        u32x4 d = v;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float dlo, dhi;
          a[c] = pack_bf16x2(gelu_fwd_grad(bf16lo(v[c]), dlo), gelu_fwd_grad(bf16hi(v[c]), dhi));
          if (p.gelu_dg) d[c] = pack_bf16x2(dlo, dhi);     // `out` carries gelu'(pre) for the backward
        }
        if (ok) {
          ST16((u32x4*)((__bf16*)p.out + (size_t)m * ldo + n), d);
          ST16((u32x4*)((__bf16*)p.out2 + (size_t)m * ldo + n), a);
        }
      } else if constexpr (EPI == EPI_RESID_F32 || EPI == EPI_PATCH_F32) {
        const f32x4 r0 = auxf[2 * it], r1 = auxf[2 * it + 1];
        float y8[8] = {bf16lo(v[0]), bf16hi(v[0]), bf16lo(v[1]), bf16hi(v[1]), bf16lo(v[2]), bf16hi(v[2]), bf16lo(v[3]), bf16hi(v[3])};
        if constexpr (EPI == EPI_RESID_F32) {
          if (p.drop_thresh) {
            const unsigned long long base = (unsigned long long)(p.row0 + m) * p.N + n;
  #pragma unroll
            for (int c = 0; c < 8; ++c) y8[c] = round_bf16(y8[c] * dropout_keep(base + c, p.drop_seed_lo, p.drop_seed_hi, p.drop_thresh, p.drop_scale));
          }
        }
        f32x4 o0 = {r0[0] + y8[0], r0[1] + y8[1], r0[2] + y8[2], r0[3] + y8[3]};
        f32x4 o1 = {r1[0] + y8[4], r1[1] + y8[5], r1[2] + y8[6], r1[3] + y8[7]};
        size_t orow = (size_t)m;
        if constexpr (EPI == EPI_PATCH_F32) {
          const int b = m / p.n_patches, pidx = m - b * p.n_patches;
          orow = (size_t)b * p.seq + p.extra + pidx;
        }
        if (ok) {
          float* op = (float*)p.out + orow * ldo + n;
          ST16((f32x4*)op, o0);
          ST16((f32x4*)(op + 4), o1);
        }
      } else if constexpr (EPI == EPI_DGELU) {
        const u32x4 pz = auxb[it];
        u32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float lo = round_bf16(bf16lo(v[c]) * (p.gelu_dg ? bf16lo(pz[c]) : gelu_grad(bf16lo(pz[c]))));
          const float hi = round_bf16(bf16hi(v[c]) * (p.gelu_dg ? bf16hi(pz[c]) : gelu_grad(bf16hi(pz[c]))));
          if (ok) { cs[2 * c] += lo; cs[2 * c + 1] += hi; }
          o[c] = pack_bf16x2(lo, hi);
        }
        if (ok) ST16((u32x4*)((__bf16*)p.out + (size_t)m * ldo + n), o);
      }
    }
  }
  if constexpr (EPI == EPI_DGELU) {
    if (p.colsum) {
      __syncthreads();                   // every wave finished with its image
      float* red = (float*)scratch;      // [8 waves][8 rsub][64 cols] floats = 16 KiB
      float* mine = red + (wave * 8 + rsub) * 64 + 8 * (pc ^ rsub);
#pragma unroll
      for (int c = 0; c < 8; ++c) mine[c] = cs[c];
      __syncthreads();
      for (int c = tid; c < BN; c += 512) {
        const int wnn = c >> 6, cc = c & 63;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 2; ++w)
#pragma unroll
          for (int r = 0; r < 8; ++r) s += red[((w * 4 + wnn) * 8 + r) * 64 + cc];
        if (n0 + c < p.N) atomicAdd(p.colsum + n0 + c, s);
      }
    }
  }
}

template <int EPI>
__global__ __launch_bounds__(512) void gemm_nt_persist_kernel(const GemmNtArgs p) {
  constexpr int BM = 256, BN = 256, WN = 4, NW = 8;
  constexpr int MT = 8, NT = 4, PPW = 8;
  constexpr int BUF_BYTES = (BM + BN) * 128;
  // younger VMEM ops (epilogue stores) guaranteed per lane after the next tile's first DMA, for FULL tiles
  constexpr int EPI_STORES = (EPI == EPI_BIAS_BF16 || EPI == EPI_DGELU) ? 16 : 32;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
  const int ntiles = tiles_m * tiles_n;
  const bool grouped = tiles_n >= 6 && !(p.dbg & 32);
  const int K = p.K;
  const int nkt = K / 64;
  const __amdgpu_buffer_rsrc_t rsrcA = make_rsrc(p.A, (size_t)p.M * K * 2);
  const __amdgpu_buffer_rsrc_t rsrcB = make_rsrc(p.B, (size_t)p.N * K * 2);
  const int frag_off = (lane & 15) * 128 + ((((lane >> 4) ^ (lane & 7)) & 7) << 4);
  const int a_off = wm * 128 * 128 + frag_off;
  const int b_off = BM * 128 + wn * 64 * 128 + frag_off;

  auto tile_origin = [&](int t, int& m0, int& n0) {
    int tm, tn;
    tile_coords(xcd_remap(t, ntiles), tiles_m, tiles_n, grouped, tm, tn);
    m0 = tm * BM;
    n0 = tn * BN;
  };
  auto make_voff = [&](int m0, int n0, unsigned (&voff)[PPW]) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int row = (i * NW + wave) * 8 + (lane >> 3);
      const int logical = (lane & 7) ^ (row & 7);
      const int grow = (i < 4) ? min(m0 + row, p.M - 1) : min(n0 + row - BM, p.N - 1);
      voff[i] = (unsigned)grow * (unsigned)(K * 2) + logical * 16;
    }
  };

  int t = blockIdx.x;
  if (t >= ntiles) return;
  int m0, n0;
  tile_origin(t, m0, n0);
  unsigned voff[PPW];
  make_voff(m0, n0, voff);
  int par = 0;                                   // LDS buffer of this tile's K-tile 0
  {
    char* base = smem + wave * 1024;
#pragma unroll
    for (int i = 0; i < PPW; ++i) buf_glds16(i < 4 ? rsrcA : rsrcB, base + i * NW * 1024, voff[i], 0);
  }
  bool counted_wait = false;                     // first wait of a tile may skip the previous tile's stores
  while (true) {
    const int tnext = t + gridDim.x;
    const bool has_next = tnext < ntiles;
    int m0n = 0, n0n = 0;
    if (has_next) tile_origin(tnext, m0n, n0n);
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int kt = 0; kt < nkt; ++kt) {
      const int cur = (par + kt) & 1;
      if (kt == 0 && counted_wait) {
        if constexpr (EPI_STORES == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const char* buf = smem + cur * BUF_BYTES;
      char* nbase = smem + (cur ^ 1) * BUF_BYTES + wave * 1024;
      const bool last = kt + 1 == nkt;
      const bool more = !last || has_next;
      if (last && has_next) make_voff(m0n, n0n, voff);     // the prefetch now targets the next tile's K-tile 0
      const int soff = last ? 0 : (kt + 1) * 128;

      bf16x8 bq[2][NT], aq[2][2];
#pragma unroll
      for (int j = 0; j < NT; ++j) bq[0][j] = *(const bf16x8*)(buf + b_off + j * 2048);
#pragma unroll
      for (int i = 0; i < 2; ++i) aq[0][i] = *(const bf16x8*)(buf + a_off + i * 2048);
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const int ks = g >> 2, pr = g & 3;
        if (more) buf_glds16(g < 4 ? rsrcA : rsrcB, nbase + g * NW * 1024, voff[g], soff);
        if (g < 7) {
          const int ks2 = (g + 1) >> 2, pr2 = (g + 1) & 3;
#pragma unroll
          for (int i = 0; i < 2; ++i) aq[(g + 1) & 1][i] = *(const bf16x8*)(buf + ((a_off + (2 * pr2 + i) * 2048) ^ (ks2 * 64)));
        }
        if (g == 1) {
#pragma unroll
          for (int j = 0; j < NT; ++j) bq[1][j] = *(const bf16x8*)(buf + ((b_off + j * 2048) ^ 64));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[2 * pr + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[ks][j], aq[g & 1][i], acc[2 * pr + i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // the last K-tile was read from buffer (par + nkt - 1) & 1; the other one is receiving the next tile
    const int last_buf = (par + nkt - 1) & 1;
    __syncthreads();                              // all waves done reading last_buf (plain barrier: may drain DMA, harmless)
    epilogue_rows_halves<EPI>(p, acc, m0, n0, wm, wn, lane, tid, wave, smem + last_buf * BUF_BYTES);
    if (!has_next) break;
    // a FULL tile issued exactly EPI_STORES stores per lane after the DMA: its landing can be waited for
    // with a counted vmcnt; a partial tile issued fewer, so fall back to vmcnt(0)
    counted_wait = (m0 + BM <= p.M) && (n0 + BN <= p.N) && (EPI != EPI_DGELU || p.colsum == nullptr);
    par = last_buf ^ 1;
    t = tnext;
    m0 = m0n;
    n0 = n0n;
  }
}

template <int EPI>
int launch_persist(const GemmNtArgs& p, hipStream_t stream) {
  constexpr int lds = 2 * 512 * 128;
  auto kern = gemm_nt_persist_kernel<EPI>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return VITAMD_ERR_LAUNCH;
    attr_done = true;
  }
  const int tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
  hipLaunchKernelGGL(kern, dim3(tiles < 256 ? tiles : 256), dim3(512), lds, stream, p);
  return hipGetLastError() == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
}

template <int BM, int BN, int WM, int WN, int EPI>
int launch(const GemmNtArgs& p, hipStream_t stream) {
  constexpr int lds = 2 * (BM + BN) * 128;
  auto kern = gemm_nt_kernel<BM, BN, WM, WN, EPI>;
  static bool attr_done = false;  // per instantiation
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return VITAMD_ERR_LAUNCH;
    attr_done = true;
  }
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
  if ((p.dbg & 0x80000) || p.N % 8 != 0 || p.ldo % 8 != 0 || p.K % 64 != 0) return false;
  if (p.epi != EPI_BIAS_BF16 && p.epi != EPI_RESID_F32 && p.epi != EPI_GELU && p.epi != EPI_DGELU) return false;
  const long tn = (p.N + 255) / 256;
  const long r256 = (((p.M + 255) / 256) * tn + 255) / 256, r320 = (((p.M + 319) / 320) * tn + 255) / 256;
  return r320 * 320 <= r256 * 256;     // ties go to the tall tile: 142 instead of 128 FLOP per staged byte (whole-step A/B: -0.5 ms on the N = 3072 GEMMs alone)
}

template <int EPI>
int dispatch_tile(const GemmNtArgs& p, hipStream_t stream) {
  // tile selector: 0 = auto, 1 = 256x128 ring kernel (2 workgroups/CU), 256 = 256x256 double-buffered,
  // 128 = 128x128 double-buffered (small problems; needs K % 64 == 0)
  int tile = p.tile;
  const long big_tiles = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
  const bool ring_ok = (size_t)p.M * p.K * 2 < 0xf0000000ull && (size_t)p.N * p.K * 2 < 0xf0000000ull;
  if (tile == 0) {
    // the grouped-issue pipe kernel is the default for big problems; small problems -> 128x128
    if (p.N >= 256 && big_tiles >= 192 && p.K % 64 == 0) tile = ring_ok ? 2 : 256;
    else tile = (p.K % 64 == 0) ? 128 : 1;
  }
  if (tile == 1) return ring_ok ? launch_ring<256, 128, 2, 2, EPI>(p, stream) : VITAMD_ERR_SHAPE;
  if (tile == 2) {
    if (!(ring_ok && p.K % 64 == 0)) return VITAMD_ERR_SHAPE;
    if (p.tile == 0 && !(p.dbg & 0x40000000)) {       // auto: the ping-pong kernel (dbg bit 30 = the round-1 pipe kernel, A/B knob)
      if constexpr (EPI == EPI_BIAS_BF16 || EPI == EPI_RESID_F32 || EPI == EPI_GELU || EPI == EPI_DGELU) {
        if (prefer_tall(p)) return launch_pp<EPI, 10, 4, 6>(p, stream);
      }
      return launch_pp<EPI, 8, 4, 6>(p, stream);
    }
    if constexpr (EPI == EPI_BIAS_BF16 || EPI == EPI_RESID_F32 || EPI == EPI_GELU || EPI == EPI_DGELU) {
      if (p.tile == 0 && prefer_tall(p)) return launch_pipe<EPI, 0, 10>(p, stream);
    }
    return (p.dbg & 0x20000000) ? launch_pipe<EPI, 5>(p, stream) : launch_pipe<EPI>(p, stream);   // dbg bit 29: A/B of the staging path
  }
  if constexpr (EPI != EPI_F32) {
    if (tile == 6) return (ring_ok && p.K % 64 == 0 && p.N % 8 == 0 && p.ldo % 8 == 0) ? launch_persist<EPI>(p, stream) : VITAMD_ERR_SHAPE;
  }
  if (tile >= 7 && tile <= 9) {
    if (!(ring_ok && p.K % 64 == 0)) return VITAMD_ERR_SHAPE;
    if constexpr (EPI == EPI_BIAS_BF16 || EPI == EPI_RESID_F32 || EPI == EPI_GELU || EPI == EPI_DGELU) {
      if (tile == 9) return launch_pp<EPI, 10, 6, 6>(p, stream);
      if (tile == 8) return launch_pp<EPI, 10, 4, 6>(p, stream);
    }
    return launch_pp<EPI, 8, 4, 6>(p, stream);
  }
  if constexpr (EPI == EPI_BIAS_BF16) {   // schedule sweep of the ping-pong kernel (tools/bench_nt_bias.py)
    if (tile >= 10 && tile <= 19 && ring_ok && p.K % 64 == 0) {
      switch (tile) {
        case 10: return launch_pp<EPI, 8, 3, 5>(p, stream);
        case 11: return launch_pp<EPI, 8, 4, 5>(p, stream);
        case 12: return launch_pp<EPI, 8, 5, 6>(p, stream);
        case 13: return launch_pp<EPI, 8, 6, 6>(p, stream);
        case 14: return launch_pp<EPI, 8, 2, 5>(p, stream);
        case 15: return launch_pp<EPI, 10, 3, 5>(p, stream);
        case 16: return launch_pp<EPI, 10, 4, 5>(p, stream);
        case 17: return launch_pp<EPI, 10, 2, 5>(p, stream);
        case 18: return launch_pp<EPI, 10, 5, 6>(p, stream);
        default: return launch_pp<EPI, 10, 3, 6>(p, stream);
      }
    }
  }
  if (tile == 3) return ring_ok ? launch_deep<EPI, 3>(p, stream) : VITAMD_ERR_SHAPE;
  if (tile == 4) return ring_ok ? launch_deep<EPI, 4>(p, stream) : VITAMD_ERR_SHAPE;
  if (tile == 5) return ring_ok ? launch_deep<EPI, 5>(p, stream) : VITAMD_ERR_SHAPE;
  if constexpr (EPI == EPI_BIAS_BF16) {   // timing-only ablations of the pipe kernel (tools/ablate_gemm.py)
    if (tile == 21) return launch_pipe<EPI, 1>(p, stream);
    if (tile == 22) return launch_pipe<EPI, 2>(p, stream);
    if (tile == 23) return launch_pipe<EPI, 3>(p, stream);
    if (tile == 24) return launch_pipe<EPI, 4>(p, stream);   // s_setprio(1) around every MFMA group
  }
  if (p.K % BK != 0) return VITAMD_ERR_SHAPE;
  if (tile == 256) return launch<256, 256, 2, 4, EPI>(p, stream);
  return launch<128, 128, 2, 2, EPI>(p, stream);
}

}  // namespace

static int dispatch_epi(const GemmNtArgs& p, hipStream_t stream) {
  switch (p.epi) {
    case EPI_BIAS_BF16: return dispatch_tile<EPI_BIAS_BF16>(p, stream);
    case EPI_GELU: return p.out2 ? dispatch_tile<EPI_GELU>(p, stream) : VITAMD_ERR_ARG;
    case EPI_RESID_F32: return p.aux ? dispatch_tile<EPI_RESID_F32>(p, stream) : VITAMD_ERR_ARG;
    case EPI_DGELU: return p.aux ? dispatch_tile<EPI_DGELU>(p, stream) : VITAMD_ERR_ARG;
    case EPI_PATCH_F32: return (p.aux && p.n_patches > 0) ? dispatch_tile<EPI_PATCH_F32>(p, stream) : VITAMD_ERR_ARG;
    case EPI_F32: return dispatch_tile<EPI_F32>(p, stream);
    default: return VITAMD_ERR_ARG;
  }
}

int vitamd_gemm_nt_impl(const GemmNtArgs& p, hipStream_t stream) {
  if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.K % 32 != 0 || p.N % 4 != 0 || p.ldo % 4 != 0) return VITAMD_ERR_SHAPE;
  if (!p.A || !p.B || !p.out) return VITAMD_ERR_ARG;
  // One big-tile workgroup per CU means a launch runs in whole rounds of 256 tiles; a last round that is mostly empty idles most of
  // the chip for a full tile time.  Two remedies live here: the tile height (prefer_tall) and the tail split below.
  constexpr int CUS = 256;
  const bool tall = p.tile == 0 && prefer_tall(p);
  const int bm = tall ? 320 : 256;
  const int tiles_m = (p.M + bm - 1) / bm, tiles_n = (p.N + 255) / 256;
  const long big_tiles = (long)tiles_m * tiles_n;
  const long rem = big_tiles % CUS;
  // Tail split: the last, mostly empty round of big tiles is re-cut into 128x128 tiles.  Only ever paid for the fc2 FORWARD GEMM
  // on 256-row tiles (591 tiles = 2.31 rounds; -0.2 ms/step), which the 320-row tile has since replaced (474 tiles = 1.85 rounds:
  // no split).  Everywhere else it loses on the whole step: +0.9 ms forced on every GEMM (bit 4) because the weight-gradient GEMMs
  // of the side stream already fill the backward tails, +0.2 ms on fc1+GELU at 7.4 rounds of 320-row tiles (the 128x128 kernel's
  // direct-store GELU epilogue costs more than the 0.6 idle round).  vitamd_set_debug bit 7 turns it off, bit 4 forces it.
  const bool split_on = (p.dbg & 16) != 0 || (!(p.dbg & 128) && p.epi == EPI_RESID_F32 && !tall);
  if (p.tile == 0 && split_on && p.epi != EPI_PATCH_F32 && p.N >= 256 && p.K % 64 == 0 && big_tiles > 2 * CUS && rem != 0 && rem * 10 < CUS * 6) {
    const int panels_a = (int)((big_tiles - rem) / tiles_n);          // M-panels whose tiles fill whole rounds
    const int rows_a = panels_a * bm;
    if (panels_a > 0 && rows_a < p.M) {
      GemmNtArgs a = p, b = p;
      a.M = rows_a;
      a.tile = tall ? 0 : 2;                                          // (auto picks the 320-row form again for the head part)
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
  }
  return dispatch_epi(p, stream);
}
