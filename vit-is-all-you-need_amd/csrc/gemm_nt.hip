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
#include "common.h"
#include "vitamd_internal.h"

namespace {

constexpr int BK = 64;  // bf16 elements per K-tile = 128 B per LDS row

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

  // ------------------------------------------------------------------ epilogue
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
          act[r] = gelu_fwd(pre[r]);
        }
        u32x2 o1 = {pack_bf16x2(pre[0], pre[1]), pack_bf16x2(pre[2], pre[3])};
        u32x2 o2 = {pack_bf16x2(act[0], act[1]), pack_bf16x2(act[2], act[3])};
        *(u32x2*)((__bf16*)p.out + (size_t)m * ldo + n) = o1;
        *(u32x2*)((__bf16*)p.out2 + (size_t)m * ldo + n) = o2;
      } else if constexpr (EPI == EPI_RESID_F32) {
        const f32x4 res = *(const f32x4*)((const float*)p.aux + (size_t)m * ldo + n);
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = res[r] + round_bf16(v[r]);
        *(f32x4*)((float*)p.out + (size_t)m * ldo + n) = o;
      } else if constexpr (EPI == EPI_DGELU) {
        const u32x2 pz = *(const u32x2*)((const __bf16*)p.aux + (size_t)m * ldo + n);
        const float pre[4] = {bf16lo(pz[0]), bf16hi(pz[0]), bf16lo(pz[1]), bf16hi(pz[1])};
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o[r] = round_bf16(round_bf16(v[r]) * gelu_grad(pre[r]));
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

template <int EPI>
int dispatch_tile(const GemmNtArgs& p, hipStream_t stream) {
  // tile choice: 256x256 (8 waves) when it yields enough workgroups, else 128x128 (4 waves)
  const long big = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
  int tile = p.tile;
  if (tile == 0) tile = (p.N >= 256 && big >= 192) ? 256 : 128;
  if (tile == 256) return launch<256, 256, 2, 4, EPI>(p, stream);
  return launch<128, 128, 2, 2, EPI>(p, stream);
}

}  // namespace

int vitamd_gemm_nt_impl(const GemmNtArgs& p, hipStream_t stream) {
  if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.K % BK != 0 || p.N % 4 != 0 || p.ldo % 4 != 0) return VITAMD_ERR_SHAPE;
  if (!p.A || !p.B || !p.out) return VITAMD_ERR_ARG;
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
