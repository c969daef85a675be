// Internal (C++) argument blocks shared between the kernel files and capi.hip.
#pragma once
#include <hip/hip_runtime.h>

enum GemmEpilogue {
  EPI_BIAS_BF16 = 0,  // out bf16 = bf16(acc + bias)            (bias may be null)
  EPI_GELU = 1,       // out bf16 = pre = bf16(acc + bias); out2 bf16 = gelu(pre)
  EPI_RESID_F32 = 2,  // out f32  = aux_f32 + bf16(acc + bias)   (residual stream stays fp32)
  EPI_DGELU = 3,      // out bf16 = bf16(bf16(acc) * gelu'(aux_bf16)); colsum[n] += column sums of out
  EPI_PATCH_F32 = 4,  // out f32[b*seq + extra + p] = bf16(acc + bias) + aux_f32[p]  (patch embed + pos_emb)
  EPI_F32 = 5,        // out f32 = acc
};

struct GemmNtArgs {
  const void* A;      // [M,K] bf16
  const void* B;      // [N,K] bf16
  void* out;          // [M(or token rows), ldo]
  void* out2;         // EPI_GELU second output
  const float* bias;  // [N] fp32 or null
  const void* aux;    // residual f32 / pre-activation bf16 / pos_emb f32
  float* colsum;      // [N] fp32 atomics target or null
  int M, N, K, ldo;
  int epi;
  int n_patches, seq, extra;  // EPI_PATCH_F32 row remap
  int tile;                   // 0 = auto, 128, 256
  int dbg;                    // timing-only ablation bits (vitamd_set_debug); 0 in production
  // EPI_RESID_F32 only: dropout on the Linear output before the residual add (reference nn.Dropout, transformer.py:40)
  unsigned drop_thresh;       // p * 2^32, 0 = off
  float drop_scale;           // 1 / (1 - p)
  unsigned drop_seed_lo, drop_seed_hi;
  int row0;                   // global row of this launch's row 0 (tail split): dropout indices stay global
  // stored-derivative GELU (ABI codes VITAMD_EPI_GELU_DG / VITAMD_EPI_DMUL): EPI_GELU writes out = bf16(gelu'(pre)) instead
  // of pre, EPI_DGELU multiplies by aux as stored instead of evaluating gelu'(aux)
  int gelu_dg;
  int dbg2;                   // second word of A/B bits (experimental builds; vitamd_set_debug2)
  const unsigned* gelu_tab;   // EPI_GELU: device image of the erf-GELU table (vitamd_init; set by vitamd_gemm_nt_impl for every GELU launch)
};

struct GemmTnArgs {
  const void* L;   // [R, P] bf16 (row stride ldl)
  const void* Rm;  // [R, Q] bf16 (row stride ldr)
  float* out;      // [P, Q] fp32, ACCUMULATED into (atomics)
  int R, P, Q, ldl, ldr, ldo;
  int splits;      // 0 = auto
  float* ws;       // optional split-K workspace: [splits][tiles][256][256] fp32 partial tiles (plain stores) + reduce pass
  size_t ws_bytes;
  int accumulate;  // with a workspace: 1 = out += sum, 0 = out = sum (no pre-zeroing needed)
  int form;        // with a workspace: 0 = 8-wave ping-pong kernel, 1 = 12-wave loader form (VITAMD_TN_FORM_*)
};

int vitamd_gemm_nt_impl(const GemmNtArgs& p, hipStream_t stream);
int vitamd_gemm_nt_plan_impl(const GemmNtArgs& p);
int vitamd_init_impl(int device, hipStream_t stream);
int vitamd_gemm_tn_impl(const GemmTnArgs& p, hipStream_t stream);
