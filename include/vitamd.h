/* vitamd.h — C ABI of libvitamd.so, the MI355X (gfx950) kernels behind the ViT training hot path.
 *
 * The reference (SnakeOnex/vit-is-all-you-need) has no FFI layer: its hot path is a handful of
 * PyTorch ATen calls made from transformer.py and train_vit.py.  Each entry point below replaces
 * one (or a fused group) of those call sites; the citation after "replaces" is the reference
 * file:line.  A maintainer binds them with ctypes; the build's own
 * `vit-is-all-you-need_amd/vitamd/lib.py` is exactly that binding.
 *
 * Conventions
 *   - All pointers are DEVICE pointers (HBM), row-major, 16-byte aligned.  "bf16" buffers are
 *     passed as void*; fp32 buffers as float*.
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream).  Every entry point except vitamd_init only enqueues work:
 *     no allocation, no synchronisation, safe under HIP-graph capture.
 *   - vitamd_init(device, stream) is called once per device before the first GEMM with a GELU epilogue: it is the ONLY function that
 *     allocates (one 16-KiB table image per device) and synchronises.  A GELU launch on a device without it returns VITAMD_ERR_INIT.
 *   - Every function returns VITAMD_OK (0) or a VITAMD_ERR_* code and launches nothing on error.
 *   - The caller owns all memory.  Kernels never allocate.
 *   - ONE HIP runtime per process: load this library after the framework that owns the streams and pointers you pass in (PyTorch bundles
 *     its own libamdhip64; a second copy pulled in by an earlier dlopen of this library fails the first launch) - see INTEGRATION.md.
 */
#ifndef VITAMD_H
#define VITAMD_H

#ifdef __cplusplus
extern "C" {
#endif

#define VITAMD_OK 0
#define VITAMD_ERR_SHAPE 1   /* unsupported / inconsistent dimensions */
#define VITAMD_ERR_ARG 2     /* missing pointer or bad enum */
#define VITAMD_ERR_LAUNCH 3  /* HIP reported a launch error */
#define VITAMD_ERR_INIT 4    /* vitamd_init has not run for the current device (GELU epilogues need its table) */

/* ABI version of this header (bumped on any signature change). */
int vitamd_abi_version(void);

/* Per-device set-up, once per device and process (idempotent; device < 0 = the current device; never inside a stream capture): builds the
 * 4 096-entry table {bf16(gelu(x)), bf16(gelu'(x))} of every bf16 input 2^-13 <= |x| < 8, correctly rounded from float64, that all GELU
 * epilogues read (replaces the erf of nn.GELU(), transformer.py:38, exactly).  The only entry point that allocates or synchronises. */
int vitamd_init(int device, void* stream);

/* ---- Linear layers: C[M,N] = A[M,K] . B[N,K]^T, bf16 operands, fp32 accumulation (MFMA) ------
 * epilogue selectors (argument `epi`):                                                          */
#define VITAMD_EPI_BIAS_BF16 0 /* out bf16 = bf16(acc + bias)                      replaces transformer.py:21,27 (qkv Linear) */
#define VITAMD_EPI_GELU 1      /* out bf16 = pre-activation, out2 bf16 = erf-GELU  replaces transformer.py:37-38 (Linear + nn.GELU) */
#define VITAMD_EPI_RESID_F32 2 /* out f32 = aux_f32 + bf16(acc + bias)             replaces transformer.py:39-40,44 (Linear + Dropout(0) + residual add) */
#define VITAMD_EPI_DGELU 3     /* out bf16 = bf16(acc) * gelu'(aux_bf16); colsum += column sums   (backward of transformer.py:38-39) */
#define VITAMD_EPI_PATCH_F32 4 /* out f32[b*seq+extra+p] = bf16(acc+bias) + aux_f32[p]  replaces train_vit.py:39-41 (Conv2d patchify + rearrange + pos_emb) */
#define VITAMD_EPI_F32 5       /* out f32 = acc */
#define VITAMD_EPI_GELU_DG 6   /* as GELU, but out bf16 = gelu'(pre-activation): the derivative is evaluated here, where its exp is
                                  shared with the erf and the VALU work hides under the output stores */
#define VITAMD_EPI_DMUL 7      /* as DGELU, but aux_bf16 already holds gelu'(pre) (written by GELU_DG): out = bf16(bf16(acc) * aux) */

/* Requirements: K % 64 == 0, N % 4 == 0, ldo % 4 == 0.  bias may be NULL.  `tile`: 0 = auto; 128 = the 128x128 small-problem
 * kernel; 256 / 320 = the ping-pong kernel on 256- / 320-row tiles (320: bias, GELU, residual and dGELU epilogues only),
 * one workgroup per tile.  Auto launches problems with more tiles than CUs PERSISTENT (one workgroup per CU walking a strided tile
 * list: faster next to a second stream's kernels, but sensitive to CUs held by other long-running kernels, e.g. collectives; with a
 * short reduction dim (K <= 1536) and >= 3 tiles per CU in the form that requests the next tile's operands before the epilogue);
 * 512 = auto without persistent launches; 1024 = auto with persistent launches but without the seam / loader forms; 2048 = the
 * loader-wave form (256-row tiles, twelve waves per workgroup, four of them only issue the operand requests; bias / GELU / dGELU-multiply
 * epilogues, K % 128 == 0; VITAMD_ERR_SHAPE where it does not apply).  All forms give bit-identical results.
 * Forward of nn.Linear (x W^T + b): A = x, B = W.  Input gradient (dy W): A = dy, B = W^T. */
int vitamd_gemm_nt_bf16(const void* A, const void* B, void* out, void* out2, const float* bias, const void* aux,
                        float* colsum, int M, int N, int K, int ldo, int epi, int n_patches, int seq, int extra,
                        int tile, void* stream);

/* Which kernel the call above would launch for these arguments on the current device (nothing is launched, no pointer is needed):
 * VITAMD_NT_FORM_* | tile rows << 8, | VITAMD_NT_FORM_TAIL_SPLIT when the launch is cut into a head of whole rounds in that form and a
 * tail on 128x128 tiles; a negative value is -VITAMD_ERR_*. */
#define VITAMD_NT_FORM_SMALL 1          /* 128x128 tiles, 4 waves */
#define VITAMD_NT_FORM_PP 2             /* ping-pong kernel, one workgroup per tile */
#define VITAMD_NT_FORM_PP_PERSISTENT 3  /* ping-pong kernel, one workgroup per CU walking a tile list */
#define VITAMD_NT_FORM_SEAM 4           /* persistent, the next tile's fill requested before the epilogue */
#define VITAMD_NT_FORM_LOADER 5         /* persistent, 8 compute waves + 4 loader waves */
#define VITAMD_NT_FORM_TAIL_SPLIT 0x80
int vitamd_gemm_nt_plan(int M, int N, int K, int ldo, int epi, int tile);

/* fc2 with dropout: out f32 = resid + dropout_p(bf16(A.B^T + bias)), mask = hash(seed, row*N+col).
 * replaces transformer.py:39-40,44 (Linear + nn.Dropout(p) + residual add) in training mode.  `tile` as for vitamd_gemm_nt_bf16. */
int vitamd_linear_dropout_resid_bf16(const void* A, const void* B, float* out, const float* bias, const float* resid,
                                     int M, int N, int K, float dropout_p, unsigned long long seed, int tile, void* stream);

/* Weight gradient: out[P,Q] (fp32) += sum_r L[r,p] * Rm[r,q]   (dW = dY^T X).  Accumulates with
 * fp32 atomics, so `out` must hold the running gradient (zeros for a fresh one).
 * replaces the autograd backward of transformer.py:21,37,39 and train_vit.py:34.
 * Requirements: ldl % 8 == 0, ldr % 8 == 0.  `splits` 0 = auto. */
int vitamd_gemm_tn_bf16(const void* L, const void* Rm, float* out, int R, int P, int Q, int ldl, int ldr, int ldo,
                        int splits, void* stream);

/* Same GEMM with a caller-provided split-K workspace (>= splits * ceil(P/256) * ceil(Q/256) * 256 KiB,
 * vitamd_gemm_tn_ws_bytes tells): partial tiles are written with plain stores and summed by a second
 * pass (bitwise reproducible, ~4x the rate of the atomic form).  accumulate = 0 overwrites `out`.
 * `form` picks the kernel (same results): which one is faster depends on what runs beside the launch. */
#define VITAMD_TN_FORM_SHARED 0    /* 8 waves per workgroup, 2/3 of the register file: waves of other kernels (LayerNorm) can share the CU */
#define VITAMD_TN_FORM_EXCLUSIVE 1 /* 12 waves (4 of them only issue the LDS-DMA requests): 15 % faster alone, fills the CU; bit-identical */
int vitamd_gemm_tn_bf16_ws(const void* L, const void* Rm, float* out, int R, int P, int Q, int ldl, int ldr, int ldo,
                           int splits, float* ws, long ws_bytes, int accumulate, int form, void* stream);
long vitamd_gemm_tn_ws_bytes(int R, int P, int Q, int splits);

/* ---- LayerNorm (no affine, eps as given) on the fp32 residual stream -------------------------
 * forward: x = x_in (+ addend_bf16 -> also written to x_out); y = bf16(LN(x)); mean/rstd saved.
 * replaces transformer.py:43-44 `F.layer_norm(x, (n_embd,))` and the residual add of :43. */
int vitamd_layernorm_fwd(const float* x_in, const void* addend_bf16, float* x_out, void* y_bf16, float* mean,
                         float* rstd, int M, int D, float eps, void* stream);
/* backward: g_out = (g_res ? g_res : 0) + LN'(dy_bf16); optional bf16 copy of g_out and its column sums. */
int vitamd_layernorm_bwd(const void* dy_bf16, const float* x, const float* mean, const float* rstd,
                         const float* g_res, float* g_out, void* g_bf16, float* colsum, int M, int D, void* stream);
/* same; the bf16 copy additionally carries the dropout mask (p, seed; element index row*D+col) of the
 * Linear output whose gradient it is (backward of the nn.Dropout of transformer.py:40). */
int vitamd_layernorm_bwd_dropout(const void* dy_bf16, const float* x, const float* mean, const float* rstd,
                                 const float* g_res, float* g_out, void* g_bf16, float* colsum, int M, int D,
                                 float dropout_p, unsigned long long seed, void* stream);
/* Same backward with xhat read back from the forward's bf16 output y_bf16 (for this non-affine LayerNorm y IS xhat, and it is
 * saved anyway as the operand of the following Linear's weight gradient) instead of recomputed from x: 14 instead of 16 B per
 * element of an HBM-bound kernel; D in {256, 512, 768, 1024}; dropout_p = 0 for no mask on the bf16 copy. */
int vitamd_layernorm_bwd_xhat(const void* dy_bf16, const void* y_bf16, const float* rstd, const float* g_res, float* g_out,
                              void* g_bf16, float* colsum, int M, int D, float dropout_p, unsigned long long seed, void* stream);

/* Affine LayerNorm (nn.LayerNorm weight/bias) for the `blocks.py` surface (blocks.py:43,48,179,184).
 * forward: y = bf16(LN(x) * gamma + beta).  backward: g_out = (g_res or 0) + dLN/dx; dgamma, dbeta are
 * ACCUMULATED into (zero them first); optional bf16 copy of g_out and its column sums as above. */
int vitamd_layernorm_affine_fwd(const float* x, const float* gamma, const float* beta, void* y_bf16, float* mean,
                                float* rstd, int M, int D, float eps, void* stream);
int vitamd_layernorm_affine_bwd(const void* dy_bf16, const float* x, const float* mean, const float* rstd,
                                const float* gamma, const float* g_res, float* g_out, void* g_bf16, float* colsum,
                                float* dgamma, float* dbeta, int M, int D, void* stream);
/* fp32-in / fp32-out forms for a LayerNorm that is NOT followed by a GEMM: the tokenizers' ln_pre / ln_post
 * (blocks.py:247,253 / :320,326), whose output stays in the fp32 token stream.  dgamma / dbeta accumulated into. */
int vitamd_layernorm_affine_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                    float* rstd, int M, int D, float eps, void* stream);
int vitamd_layernorm_affine_bwd_f32(const float* dy, const float* x, const float* mean, const float* rstd,
                                    const float* gamma, float* g_out, float* dgamma, float* dbeta, int M, int D, void* stream);

/* ---- Attention on the packed fused-QKV layout ------------------------------------------------
 * qkv bf16 [B,N,3,H,64] (output-channel order (qkv, head, dh) of transformer.py:27), o bf16 [B,N,H*64],
 * lse2 fp32 [B,H,N].  head_dim must be 64, N <= 16384 (N <= 512: one LDS-resident chunk per head; longer: both sides tiled).  causal != 0 applies the strictly-upper -inf
 * mask of transformer.py:22-25.   replaces transformer.py:27-29 (rearrange + SDPA + rearrange).
 * dropout_p in [0,1): dropout on the softmax probabilities (SDPA's dropout_p); the mask is a stateless
 * hash of (seed, batch, head, query, key), so the backward call regenerates it from the same seed. */
int vitamd_attention_fwd(const void* qkv, void* o, float* lse2, int B, int N, int H, int head_dim, int causal,
                         float dropout_p, unsigned long long seed, void* stream);
/* Forward fused with the residual add that follows it in the layer (transformer.py:44 `x = x + attn(LN(x))`):
 * resid_out fp32 [B*N, H*64] = resid_in + o (o as rounded to bf16), written by the same workgroup that produces o, so the next
 * LayerNorm reads the stream once (6 instead of 12 B per element).  N <= 256 (the register-resident-softmax kernel). */
int vitamd_attention_fwd_resid(const void* qkv, void* o, float* lse2, const float* resid_in, float* resid_out, int B, int N, int H,
                               int head_dim, int causal, float dropout_p, unsigned long long seed, void* stream);
/* dqkv bf16 [B,N,3,H,64]; delta fp32 [B,H,N] is scratch written by the call; dbias (may be NULL) fp32
 * [3*H*64]: the column sums of dqkv (= gradient of the QKV bias, transformer.py:21) are ADDED to it. */
int vitamd_attention_bwd(const void* qkv, const void* o, const float* lse2, const void* d_o, void* dqkv,
                         float* delta, float* dbias, int B, int N, int H, int head_dim, int causal, float dropout_p,
                         unsigned long long seed, void* stream);

/* ---- helpers around the GEMMs ---------------------------------------------------------------- */
/* fp32 -> bf16 (autocast's per-step weight / activation cast, train_vit.py:100). */
int vitamd_cast_f32_bf16(const float* in, void* out_bf16, long n, void* stream);
/* bf16(x) * dropout mask (p, seed; element index) — the top layer's fc2 output gradient under dropout. */
int vitamd_cast_f32_bf16_dropout(const float* in, void* out_bf16, long n, float dropout_p, unsigned long long seed, void* stream);
/* out[i] = in[i] * keep(seed, i / group): nn.Dropout (group = 1; reference blocks.py:118 proj_drop, :168-170 Mlp.drop) and DropPath
 * (group = elements per sample; blocks.py:124-152) in training mode.  keep = 1/(1-p) or 0 from the stateless (seed, index) hash;
 * the backward is the same call on the gradient.  In place (out == in) is allowed. */
int vitamd_dropout_bf16(const void* in, void* out, long n, long group, float dropout_p, unsigned long long seed, void* stream);
int vitamd_dropout_f32(const float* in, float* out, long n, long group, float dropout_p, unsigned long long seed, void* stream);
/* W fp32 [N,K] -> bf16 [N,K] (wb, may be NULL) and transposed bf16 [K,N] (wbt, may be NULL). */
int vitamd_cast_transpose_weight(const float* w, void* wb, void* wbt, int N, int K, void* stream);
/* The same for n weights in ONE launch.  desc_dev: device array of n records
 * {const float* w; bf16* wb; bf16* wbt; int N, K, first_tile, tiles_k;} (40 bytes, 8-byte aligned),
 * first_tile = running sum of ceil(N/64)*ceil(K/64), tiles_k = ceil(K/64); total_tiles = the final sum. */
int vitamd_cast_transpose_batched(const void* desc_dev, int n, int total_tiles, void* stream);
/* images fp32 [B,C,H,W] -> patches bf16 [B*(H/p)*(W/p), C*p*p], vector order (c,kh,kw): the
 * contraction order of Conv2d(kernel=stride=p), train_vit.py:34,39. */
int vitamd_im2col_bf16(const float* img, void* out_bf16, int B, int C, int H, int W, int p, void* stream);
/* out[n] += sum_m X[m,n]  (bias gradients). */
int vitamd_colsum_bf16(const void* x_bf16, float* out, int M, int N, int ld, void* stream);
/* Backward of the token assembly train_vit.py:41-44: g fp32 [B,seq,D] -> dpos [seq-extra,D],
 * dextra [extra,D], compact bf16 patch rows dyp [B*(seq-extra),D], dbias[D] = their column sums.
 * dpos, dextra and dbias are ACCUMULATED into (atomics): zero them for a fresh gradient.
 * dbias_rows: zeroed scratch [seq-extra, D] (per-position partials of dbias, summed by a second pass). */
int vitamd_embed_bwd(const float* g, float* dpos, float* dextra, void* dyp_bf16, float* dbias, float* dbias_rows, int B,
                     int seq, int extra, int D, void* stream);

/* ---- VQ quantiser (TiTok / ViT-VQGAN) --------------------------------------
 * idx[m] = argmin_k ||x[m,:] - codebook[k,:]||^2, first minimum, fp32; d <= 1024; idx is int64.
 * replaces train_titok.py:53 / train_vit_vqgan.py:52 `torch.cdist(x, embedding).argmin(dim=-1)` and the
 * expanded-distance argmin of blocks.py:442-446 (blocks.VectorQuantizer). */
int vitamd_vq_nearest(const float* x, const float* codebook, long long* idx, int M, int K, int d, void* stream);

/* ---- 3x3 smoothing convolution of the pixel decoders (blocks.py surface) ----
 * NCHW fp32, stride 1, zero padding 1, Cin = Cout = 3 (anything else: VITAMD_ERR_SHAPE).
 * w [Cout,Cin,3,3], bias [Cout] or null.  replaces `self.conv_out = nn.Conv2d(3, 3, 3, padding=1)`
 * (blocks.py:333, applied at :355 and :402).
 * bwd: dx (or null) = input gradient, overwritten; dw [Cout,Cin,3,3] (or null) and db [Cout] (or null) are
 * ACCUMULATED into (atomics): zero them for a fresh gradient. */
int vitamd_conv3x3_fwd(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, int H, int W,
                       void* stream);
int vitamd_conv3x3_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int B, int Cin,
                       int Cout, int H, int W, void* stream);

/* ---- optimiser -------------------------------------------------------------------------------
 * One fused AdamW update (decoupled weight decay, bias correction for 1-based `step`) of n fp32
 * parameters in place; m, v are the optimiser state.  replaces train_vit.py:82,105 (torch.optim.AdamW). */
int vitamd_adamw_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                      float eps, float weight_decay, int step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VITAMD_H */
