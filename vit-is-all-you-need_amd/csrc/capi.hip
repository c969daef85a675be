// extern "C" entry points of libvitamd.so that wrap C++ argument blocks (see include/vitamd.h).
#include "common.h"
#include "vitamd_internal.h"
#include "../../include/vitamd.h"

extern "C" int vitamd_abi_version(void) { return 4; }

int g_vitamd_debug = 0;
// Diagnostics knob for the A/B tools (tools/ab_dbg.py, tools/ablate_*.py); process-global, 0 in production, NOT in the public
// header.  Bits marked (!) make results wrong (timing only).
//   NT GEMM : 0 skip gelu math(!)   1 skip 2nd GELU store(!)   2 skip residual load(!)   3 plain (temporal) output stores
//             4 tail split for every GEMM   5 plain tile order   7 no fc2-forward tail split   8-15 stagger unit in ~us (255 = off)
//             16 no output stores(!)   17 one K-tile only(!)   19 no 320-row tiles   20-23 stagger phases   24 stagger map
//             29 stage through VGPRs instead of LDS-DMA
//   TN GEMM : 6 16x16x32 form   25 256x384-tile kernel (gemm_tn_wide.hip)   26-28: 1 no MFMA(!) 2 no loads(!) 3 no LDS reads(!) 5 LDS-DMA staging
extern "C" int vitamd_set_debug(int bits) { g_vitamd_debug = bits; return 0; }

extern "C" int vitamd_gemm_nt_bf16(const void* A, const void* B, void* out, void* out2, const float* bias, const void* aux,
                                   float* colsum, int M, int N, int K, int ldo, int epi, int n_patches, int seq, int extra,
                                   int tile, void* stream) {
  // ABI codes 6 / 7 are the GELU / dGELU epilogues in stored-derivative form (out = gelu'(pre) ; multiply by aux as stored)
  const int dg = (epi == 6 || epi == 7) ? 1 : 0;
  if (epi == 6) epi = EPI_GELU;
  if (epi == 7) epi = EPI_DGELU;
  GemmNtArgs p{A, B, out, out2, bias, aux, colsum, M, N, K, ldo, epi, n_patches, seq, extra, tile, g_vitamd_debug, 0u, 1.0f, 0u, 0u, 0, dg};
  if (!(tile >= 0 && tile <= 19) && tile != 128 && tile != 256 && (tile < 21 || tile > 24)) return VITAMD_ERR_ARG;
  return vitamd_gemm_nt_impl(p, (hipStream_t)stream);
}

extern "C" int vitamd_gemm_tn_bf16(const void* L, const void* Rm, float* out, int R, int P, int Q, int ldl, int ldr, int ldo,
                                   int splits, void* stream) {
  GemmTnArgs a{L, Rm, out, R, P, Q, ldl, ldr, ldo, splits, nullptr, 0, 1};
  return vitamd_gemm_tn_impl(a, (hipStream_t)stream);
}

extern "C" int vitamd_gemm_tn_bf16_ws(const void* L, const void* Rm, float* out, int R, int P, int Q, int ldl, int ldr, int ldo,
                                      int splits, float* ws, long ws_bytes, int accumulate, void* stream) {
  GemmTnArgs a{L, Rm, out, R, P, Q, ldl, ldr, ldo, splits, ws, (size_t)(ws_bytes < 0 ? 0 : ws_bytes), accumulate};
  return vitamd_gemm_tn_impl(a, (hipStream_t)stream);
}

static bool dropout_params(float p, unsigned& thresh, float& scale) {
  if (!(p >= 0.0f) || p >= 1.0f) return false;
  thresh = p > 0.0f ? (unsigned)((double)p * 4294967296.0) : 0u;
  if (p > 0.0f && thresh == 0u) thresh = 1u;
  scale = 1.0f / (1.0f - p);
  return true;
}

// fc2 with dropout: out f32 = resid + dropout_p(bf16(A.B^T + bias)) — reference transformer.py:39-40,44
extern "C" int vitamd_linear_dropout_resid_bf16(const void* A, const void* B, float* out, const float* bias, const float* resid,
                                                int M, int N, int K, float dropout_p, unsigned long long seed, void* stream) {
  GemmNtArgs p{A, B, out, nullptr, bias, resid, nullptr, M, N, K, N, EPI_RESID_F32, 0, 0, 0, 0, g_vitamd_debug, 0u, 1.0f,
               (unsigned)seed, (unsigned)(seed >> 32), 0, 0};
  if (!dropout_params(dropout_p, p.drop_thresh, p.drop_scale)) return VITAMD_ERR_ARG;
  return vitamd_gemm_nt_impl(p, (hipStream_t)stream);
}
