// Epilogues of the NT GEMM kernels (shared by gemm_nt.hip and, in experimental builds, its measured alternatives).
#pragma once
#include "common.h"
#include "vitamd_internal.h"

namespace {

constexpr int BK = 64;  // bf16 elements per K-tile = 128 B per LDS row

// erf-GELU by table for bf16 inputs (every EPI_GELU epilogue).  The pre-activation reaches the epilogue rounded to bf16 (autocast's Linear output,
// transformer.py:37-38), so gelu(x) and gelu'(x) are functions of 16 bits: for 2^-13 <= |x| < 8 (16 exponents x 128 mantissas x 2 signs = 4096
// inputs) the table holds bf16(gelu(x)) | bf16(gelu'(x)) << 16, correctly rounded from double (gemm_nt.hip::vitamd_init_impl builds the device
// image).  `tab` is the LDS copy in the persistent kernels (seam / loader forms: two ds_read_b32 per pair) and the device image itself in the
// others (small problems: 4-byte gathers that hit the 16-KiB image in L1) - ONE rounding of GELU whatever kernel a launch takes.
// Per PAIR of elements: 9 packed-16-bit / 32-bit integer operations, two table reads and two v_perm against ~41 VALU-equivalents of the
// erf / exp / rcp formula (two quarter-rate transcendentals per element).  Inputs outside the table (|x| < 2^-13, |x| >= 8, inf, nan: ~1e-4 of
// N(0,1) data) take the formula - a wave-uniform branch, per element.  V = u32x4 (8 elements) or u32x2 (4).
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
template <typename V>
__device__ __forceinline__ void gelu_lookup(const V& v, const char* tab, bool want_dg, V& a, V& d) {
  constexpr int NPR = (int)(sizeof(V) / 4);
  constexpr unsigned T_LO = 0x3900u;                 // bf16 bits of 2^-13; the table ends below 0x4100 = 8.0
  unsigned r[NPR], any = 0u;
#pragma unroll
  for (int c = 0; c < NPR; ++c) {
    const unsigned t = v[c] & 0x7fff7fffu;
    const u16x2 rr = __builtin_bit_cast(u16x2, t) - (u16x2){(unsigned short)T_LO, (unsigned short)T_LO};      // wraps below the table
    r[c] = __builtin_bit_cast(unsigned, rr);
    any |= r[c];
    const u16x2 rc = __builtin_elementwise_min(rr, (u16x2){2047, 2047});
    // byte offset of entry sign x 2048 + index in each half: plain 32-bit arithmetic (no carry between the halves: index <= 4095).  Written this
    // way on purpose: with the sign taken by a packed 16-bit shift inside this unrolled loop hipcc 7.2 used pair 0's sign for all four pairs.
    const unsigned b = (__builtin_bit_cast(unsigned, rc) | ((v[c] >> 4) & 0x08000800u)) << 2;
    const unsigned elo = *(const unsigned*)(tab + (b & 0xffffu));
    const unsigned ehi = *(const unsigned*)(tab + (b >> 16));
    a[c] = __builtin_amdgcn_perm(ehi, elo, 0x05040100u);
    if (want_dg) d[c] = __builtin_amdgcn_perm(ehi, elo, 0x07060302u);
  }
  if (__builtin_amdgcn_ballot_w64((any & 0xf800f800u) != 0u)) {                    // an index >= 2048 in some lane of the wave
#pragma unroll
    for (int c = 0; c < NPR; ++c) {
      float dlo, dhi;
      const unsigned g = pack_bf16x2(gelu_fwd_grad(bf16lo(v[c]), dlo), gelu_fwd_grad(bf16hi(v[c]), dhi));
      const unsigned dg = pack_bf16x2(dlo, dhi);
      const unsigned m = ((r[c] & 0xf800u) ? 0xffffu : 0u) | ((r[c] & 0xf8000000u) ? 0xffff0000u : 0u);
      a[c] = (g & m) | (a[c] & ~m);
      if (want_dg) d[c] = (dg & m) | (d[c] & ~m);
    }
  }
}
__device__ __forceinline__ void gelu_lookup8(const u32x4& v, const char* tab, bool want_dg, u32x4& a, u32x4& d) { gelu_lookup<u32x4>(v, tab, want_dg, a, d); }

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
        const u32x2 pz = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};       // the Linear output, rounded to bf16
        u32x2 o1 = pz, o2;                                                           // `out` carries the pre-activation, or gelu'(pre) for the backward (gelu_dg)
        gelu_lookup<u32x2>(pz, (const char*)p.gelu_tab, p.gelu_dg != 0, o2, o1);
        *(u32x2*)((__bf16*)p.out + (size_t)m * ldo + n) = o1;
        if (!(VITAMD_DBG(p) & 2)) *(u32x2*)((__bf16*)p.out2 + (size_t)m * ldo + n) = o2;
      } else if constexpr (EPI == EPI_RESID_F32) {
        const f32x4 res = (VITAMD_DBG(p) & 4) ? v : *(const f32x4*)((const float*)p.aux + (size_t)m * ldo + n);
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
// panels out of the 8 x 4 MiB L2s: -5..7 % on the K = 3072 shapes); VITAMD_DBG(p) bit 3 turns that off (A/B knob)
#define ST16(ptr, val)                                              \
  do {                                                              \
    if (VITAMD_DBG(p) & 8) *(ptr) = (val);                                  \
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
    const int m = (VITAMD_DBG(p) & 4) ? m_true % 640 : m_true;          // dbg bit 2: timing-only, every tile stores to the same few (L2-resident) rows
    const u32x4 v = *(const u32x4*)(tile + rloc * 128 + pc * 16);
    const bool ok = m_true < p.M && ncol_ok && !(VITAMD_DBG(p) & 0x10000);   // dbg bit 16: timing-only, no output stores
    if constexpr (EPI == EPI_BIAS_BF16) {
      if (ok) ST16((u32x4*)((__bf16*)p.out + (size_t)m * ldo + n), v);
    } else if constexpr (EPI == EPI_GELU) {
      u32x4 a;
      u32x4 d = v;                                          // `out` carries the pre-activation, or gelu'(pre) for the backward (gelu_dg)
      gelu_lookup8(v, (const char*)p.gelu_tab, p.gelu_dg != 0, a, d);
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


}  // namespace
