// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels.  wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;   // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;      // 16x16 accumulator
typedef __attribute__((ext_vector_type(16))) float f32x16;    // 32x32 accumulator
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

// Timing-only / A-B debug knobs exist only in experimental builds (`make EXPERIMENTAL=1` -> libvitamd_exp.so); in the production
// library every knob reads as the constant 0 and the code behind it is compiled out.
#ifdef VITAMD_EXPERIMENTAL
extern int g_vitamd_debug;
extern int g_vitamd_debug2;
#define VITAMD_DBG(p) ((p).dbg)
#define VITAMD_GDBG g_vitamd_debug
#else
#define VITAMD_DBG(p) 0
#define VITAMD_GDBG 0
#endif

#define VITAMD_OK 0
#define VITAMD_ERR_SHAPE 1
#define VITAMD_ERR_ARG 2
#define VITAMD_ERR_LAUNCH 3
#define VITAMD_ERR_INIT 4

// fp32 -> bf16 round-to-nearest-even (plain cast: hipcc emits v_cvt_pk_bf16_f32, NaN-safe)
__device__ __forceinline__ __bf16 f2bf(float x) { return (__bf16)x; }
__device__ __forceinline__ float bf2f(__bf16 x) { return (float)x; }
__device__ __forceinline__ float round_bf16(float x) { return (float)(__bf16)x; }

__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned int, v);
}
__device__ __forceinline__ float bf16lo(unsigned int u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16hi(unsigned int u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// async global -> LDS copy, 16 B per lane; LDS destination = wave-uniform base + lane*16
__device__ __forceinline__ void glds16(const void* gptr, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)gptr, (LDS_AS void*)lds_wave_base, 16, 0, 0);
}

// same through a buffer descriptor (range-checked: out-of-range lanes write zeros); voff = per-lane
// byte offset, soff = wave-uniform byte offset.  Kept in a NON-template function on purpose: called
// from a dependent context hipcc (ROCm 7.2) silently drops the kernel's host stub.
__device__ __forceinline__ void buf_glds16(__amdgpu_buffer_rsrc_t rsrc, void* lds_wave_base, unsigned voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LDS_AS void*)lds_wave_base, 16, voff, soff, 0, 0);
}
// same addressing, but into registers (the conventional global -> VGPR -> ds_write staging path)
__device__ __forceinline__ u32x4 buf_load16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, int soff) {
  return __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}

// ---- LDS-DMA that hipcc does not see (the ping-pong GEMM main loops) -------------------------------------------------------
// hipcc (ROCm 7.2) models an LDS-DMA builtin as a pending LDS store and puts `s_waitcnt vmcnt(0)` in front of the next ds_read
// of the same __shared__ array - which drains a pipeline built on COUNTED waits.  Issued from inline asm the loads are invisible
// to that bookkeeping: the kernel counts them by hand (s_waitcnt vmcnt(N) + s_barrier before any read of the landed bytes;
// cdna_hip_programming.md section 5.7 item 1) and drains them with vmcnt(0) before LDS is reused or the wave ends.
typedef __attribute__((ext_vector_type(4))) unsigned int srd_t;     // the four descriptor dwords, held in SGPRs
__device__ __forceinline__ srd_t make_srd(const void* base, size_t bytes) {
  const unsigned long long b = (unsigned long long)base;
  srd_t r;
  r[0] = __builtin_amdgcn_readfirstlane((unsigned)b);
  r[1] = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xffffu);   // base[47:32], stride 0
  r[2] = __builtin_amdgcn_readfirstlane((unsigned)bytes);                 // num_records: range check, out-of-range lanes write zeros
  r[3] = 0x00020000u;
  return r;
}
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(LDS_AS const void*)p; }
// 16 B per lane from srd[voff + soff] to LDS byte address lds_dst (wave-uniform) + 16*lane.  M0 is saved and restored inside the
// statement (compiler-reserved); s_nop: SGPR/M0 write -> VMEM read wait states.
__device__ __forceinline__ void asm_glds16(srd_t srd, unsigned lds_dst, unsigned voff, unsigned soff) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 2\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(lds_dst), "v"(voff), "s"(srd), "s"(soff) : "memory");
}

// the same with a cache policy on the load (POL 1 = nt, 2 = sc1, 3 = sc0 sc1: all served from L2 without keeping the line in this CU's L1)
template <int POL>
__device__ __forceinline__ void asm_glds16_pol(srd_t srd, unsigned lds_dst, unsigned voff, unsigned soff) {
  unsigned keep;
  if constexpr (POL == 1)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 2\n\tbuffer_load_dwordx4 %2, %3, %4 offen nt lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_dst), "v"(voff), "s"(srd), "s"(soff) : "memory");
  else if constexpr (POL == 2)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 2\n\tbuffer_load_dwordx4 %2, %3, %4 offen sc1 lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_dst), "v"(voff), "s"(srd), "s"(soff) : "memory");
  else if constexpr (POL == 3)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 2\n\tbuffer_load_dwordx4 %2, %3, %4 offen sc0 sc1 lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_dst), "v"(voff), "s"(srd), "s"(soff) : "memory");
  else
    asm_glds16(srd, lds_dst, voff, soff);
}

// 4 B per lane from srd[voff + soff] to LDS byte address lds_dst (wave-uniform) + 4*lane: one row of 64 floats per wave-instruction
__device__ __forceinline__ void asm_glds4(srd_t srd, unsigned lds_dst, unsigned voff, unsigned soff) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 2\n\tbuffer_load_dword %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(lds_dst), "v"(voff), "s"(srd), "s"(soff) : "memory");
}

// 16 B per lane from srd[voff] into registers, from inline asm (invisible to hipcc's wait bookkeeping, like asm_glds16): the caller waits with a
// counted s_waitcnt statement that names the destination as a "+v" operand before the first use (cdna_hip_programming.md section 5.7, form ii).
// Out-of-range offsets return zeros.
__device__ __forceinline__ u32x4 asm_bload16(srd_t srd, unsigned voff, unsigned soff = 0u) {
  u32x4 r;
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(r) : "v"(voff), "s"(srd), "s"(soff) : "memory");
  return r;
}

// 16 B per lane to srd[voff + soff], non-temporal, from inline asm.  The s_nop 1 inside the statement is the store-data hazard pad (the next
// instruction may overwrite the data registers; cdna_hip_programming.md section 5.7 item 1): with a REGISTER soffset hipcc does not pad its
// own buffer stores either, and on gfx950 such a store was seen sending the next row's index in place of a data dword.
__device__ __forceinline__ void asm_bstore16_nt(u32x4 data, srd_t srd, unsigned voff, unsigned soff) {
  asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen nt\n\ts_nop 1" :: "v"(data), "v"(voff), "s"(srd), "s"(soff) : "memory");
}

// dynamic-LDS opt-in for a kernel; called on every launch (a cheap host call) so it holds for whichever device is current
template <typename K>
static inline int set_lds(K kern, int bytes) {
  return hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess ? VITAMD_OK : VITAMD_ERR_LAUNCH;
}

// erf with |abs err| < 1.5e-7 (Abramowitz & Stegun 7.1.26): one v_exp, one v_rcp, 5 fma.
// Outputs are rounded to bf16 (2^-9 relative) so this is exact for our purposes.
__device__ __forceinline__ float fast_erf(float x) {
  float ax = __builtin_fabsf(x);
  float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  float p = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  float e = __expf(-ax * ax);
  float r = 1.0f - p * e;
  return __builtin_copysignf(r, x);
}
__device__ __forceinline__ float gelu_fwd(float x) { return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752f)); }
// gelu(x) and gelu'(x) together: the exp(-x^2/2) inside the erf approximation IS the Gaussian of the derivative,
// so the pair costs one v_exp + one v_rcp (the derivative rides along for ~4 more VALU ops).
__device__ __forceinline__ float gelu_fwd_grad(float x, float& dg) {
  const float ax = __builtin_fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  const float pl = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float e = __expf(-ax * ax);                        // exp(-x^2 / 2)
  const float cdf = 0.5f * (1.0f + __builtin_copysignf(1.0f - pl * e, x));
  dg = cdf + x * (0.39894228040143268f * e);
  return x * cdf;
}
__device__ __forceinline__ float gelu_grad(float x) {
  // d/dx [x Phi(x)] = Phi(x) + x phi(x)
  float cdf = 0.5f * (1.0f + fast_erf(x * 0.70710678118654752f));
  float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// Counter-based dropout: element `idx` of a tensor is kept with probability 1-p, decided by a
// stateless 64->32-bit integer hash of (idx, seed), so forward and backward regenerate the same mask
// from (seed, index) and nothing is stored.  thresh = p * 2^32; returns 1/(1-p) (kept) or 0 (dropped).
__device__ __forceinline__ unsigned mix32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float dropout_keep(unsigned long long idx, unsigned seed_lo, unsigned seed_hi, unsigned thresh, float scale) {
  unsigned h = mix32((unsigned)idx ^ seed_lo);
  h = mix32(h ^ (unsigned)(idx >> 32) ^ seed_hi);
  return h >= thresh ? scale : 0.f;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// linear tile id -> (tile_m, tile_n).  Plain order walks N fastest.  Grouped order (wide N): blocks
// of GROUP_M row panels x all N tiles, M fastest inside a block, so the ~32 tiles an XCD works on at a
// time cover ~8 A panels x 4 B tiles (4.5 MB at K = 768) instead of 2.7 panels x 12 B tiles (5.6 MB,
// re-fetching every weight tile from beyond L2 for each small group of panels).
__device__ __forceinline__ void tile_coords(int tile, int tiles_m, int tiles_n, bool grouped, int& tm, int& tn) {
  constexpr int GROUP_M = 8;
  if (!grouped) { tm = tile / tiles_n; tn = tile % tiles_n; return; }
  const int per_group = GROUP_M * tiles_n;
  const int g = tile / per_group, r = tile - g * per_group;
  const int gm0 = g * GROUP_M;
  const int gsz = min(GROUP_M, tiles_m - gm0);
  tm = gm0 + r % gsz;
  tn = r / gsz;
}

// bijective XCD-aware remap of a linear workgroup id: consecutive remapped ids share an XCD
// (blocks b and b+8 share an XCD under round-robin dispatch).  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}
