// extern "C" entry points of libvitamd.so that wrap C++ argument blocks (see include/vitamd.h).
#include "common.h"
#include "vitamd_internal.h"
#include "../../include/vitamd.h"

extern "C" int vitamd_abi_version(void) { return 8; }

// Per-device set-up: the only entry point that allocates (16 KiB: the erf-GELU table) or synchronises.  Idempotent; device < 0 = the current device.
extern "C" int vitamd_init(int device, void* stream) { return vitamd_init_impl(device, (hipStream_t)stream); }

#ifdef VITAMD_EXPERIMENTAL
int g_vitamd_debug = 0;
// Diagnostics knobs of the EXPERIMENTAL library only (libvitamd_exp.so: tools/ab_*.py, tools/ablate_*.py, tools/bench_ld.py); process-global, not in the
// public header, absent from the production library.  Bits marked (!) make results wrong (timing only).  Each kernel file documents its own bits
// where it reads them (VITAMD_DBG / g_vitamd_debug / g_vitamd_debug2); the measured alternative KERNELS of rounds 1-3 that these words used to select
// were deleted in round 4 (their numbers live in DESIGN.md and under profiles/r02, profiles/r03).
extern "C" int vitamd_set_debug(int bits) { g_vitamd_debug = bits; return 0; }
int g_vitamd_debug2 = 0;     // second word (experiments of round 4 on: the first one is full): bits 0-3 = which launch classes take the loader-wave NT form (gemm_nt.hip::ld_auto)
extern "C" int vitamd_set_debug2(int bits) { g_vitamd_debug2 = bits; return 0; }
#endif

extern "C" int vitamd_gemm_nt_bf16(const void* A, const void* B, void* out, void* out2, const float* bias, const void* aux,
                                   float* colsum, int M, int N, int K, int ldo, int epi, int n_patches, int seq, int extra,
                                   int tile, void* stream) {
  // ABI codes 6 / 7 are the GELU / dGELU epilogues in stored-derivative form (out = gelu'(pre) ; multiply by aux as stored)
  const int dg = (epi == 6 || epi == 7) ? 1 : 0;
  if (epi == 6) epi = EPI_GELU;
  if (epi == 7) epi = EPI_DGELU;
  GemmNtArgs p{A, B, out, out2, bias, aux, colsum, M, N, K, ldo, epi, n_patches, seq, extra, tile, VITAMD_GDBG, 0u, 1.0f, 0u, 0u, 0, dg};
#ifdef VITAMD_EXPERIMENTAL
  if (tile != 0 && tile != 24 && tile != 25 && tile != 30 && tile != 128 && tile != 256 && tile != 320 && tile != 512 && tile != 1024 && tile != 2048 && tile != 2049 && tile != 4096) return VITAMD_ERR_ARG;
#else
  if (tile != 0 && tile != 128 && tile != 256 && tile != 320 && tile != 512 && tile != 1024 && tile != 2048) return VITAMD_ERR_ARG;
#endif
  return vitamd_gemm_nt_impl(p, (hipStream_t)stream);
}

// Which kernel the call above would launch on the current device (no launch, no pointer is read): see include/vitamd.h VITAMD_NT_FORM_*.
extern "C" int vitamd_gemm_nt_plan(int M, int N, int K, int ldo, int epi, int tile) {
  const int dg = (epi == 6 || epi == 7) ? 1 : 0;
  if (epi == 6) epi = EPI_GELU;
  if (epi == 7) epi = EPI_DGELU;
  static char dummy[16];                   // the launch rules only ask whether the optional pointers are present
  GemmNtArgs p{dummy, dummy, dummy, dummy, nullptr, dummy, (float*)dummy, M, N, K, ldo, epi, 1, 1, 0, tile, VITAMD_GDBG, 0u, 1.0f, 0u, 0u, 0, dg};
  if (tile != 0 && tile != 128 && tile != 256 && tile != 320 && tile != 512 && tile != 1024 && tile != 2048) return -VITAMD_ERR_ARG;
  return vitamd_gemm_nt_plan_impl(p);
}

extern "C" int vitamd_gemm_tn_bf16(const void* L, const void* Rm, float* out, int R, int P, int Q, int ldl, int ldr, int ldo,
                                   int splits, void* stream) {
  GemmTnArgs a{L, Rm, out, R, P, Q, ldl, ldr, ldo, splits, nullptr, 0, 1, 0};
  return vitamd_gemm_tn_impl(a, (hipStream_t)stream);
}

extern "C" int vitamd_gemm_tn_bf16_ws(const void* L, const void* Rm, float* out, int R, int P, int Q, int ldl, int ldr, int ldo,
                                      int splits, float* ws, long ws_bytes, int accumulate, int form, void* stream) {
  if (form != 0 && form != 1) return VITAMD_ERR_ARG;
  GemmTnArgs a{L, Rm, out, R, P, Q, ldl, ldr, ldo, splits, ws, (size_t)(ws_bytes < 0 ? 0 : ws_bytes), accumulate, form};
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
                                                int M, int N, int K, float dropout_p, unsigned long long seed, int tile, void* stream) {
  if (tile != 0 && tile != 128 && tile != 256 && tile != 320 && tile != 512 && tile != 1024) return VITAMD_ERR_ARG;
  GemmNtArgs p{A, B, out, nullptr, bias, resid, nullptr, M, N, K, N, EPI_RESID_F32, 0, 0, 0, tile, VITAMD_GDBG, 0u, 1.0f,
               (unsigned)seed, (unsigned)(seed >> 32), 0, 0};
  if (!dropout_params(dropout_p, p.drop_thresh, p.drop_scale)) return VITAMD_ERR_ARG;
  return vitamd_gemm_nt_impl(p, (hipStream_t)stream);
}
