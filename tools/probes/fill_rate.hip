// How fast can ONE CU be fed?  A workgroup of W waves per CU streams a per-workgroup region through (a) LDS-DMA (buffer_load_dwordx4 ... lds,
// 1 KiB per wave-instruction, as the GEMM main loops issue them) or (b) plain 16-B-per-lane loads into registers, with G pieces in flight per wave,
// and reports bytes per clock per CU.  Region sizes: L2-resident (every workgroup re-reads its own 64 KiB: 16 MiB over the chip) and streaming
// (each workgroup walks its own 16 MiB: HBM / MALL).  No MFMAs, no ds_reads: the ceiling of the path alone.
// Build: hipcc -O3 --offload-arch=gfx950 fill_rate.hip -o fill_rate     
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(4))) unsigned int srd_t;
__device__ __forceinline__ srd_t make_srd(const void* base, size_t bytes) {
  const unsigned long long b = (unsigned long long)base;
  srd_t r;
  r[0] = __builtin_amdgcn_readfirstlane((unsigned)b);
  r[1] = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xffffu);
  r[2] = __builtin_amdgcn_readfirstlane((unsigned)bytes);
  r[3] = 0x00020000u;
  return r;
}
__device__ __forceinline__ void glds16(srd_t srd, unsigned lds_dst, unsigned voff, unsigned soff) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 2\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(lds_dst), "v"(voff), "s"(srd), "s"(soff) : "memory");
}
template <int MODE, int G>      // MODE 0: LDS-DMA, 1: registers, 2: LDS-DMA with GEMM-shaped pieces (two 512-B row segments, rows `share`-strided: see main);  G pieces in flight per wave
__global__ __launch_bounds__(1024) void fill_kernel(const char* src, size_t region, int iters, unsigned long long* cycles, unsigned* sink, int share) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  // share > 1: the workgroups b, b + 8, ... of one XCD (round-robin dispatch) read the SAME region in groups of `share`, as GEMM tiles share operand panels
  const char* mine = MODE == 2 ? src + (size_t)((((unsigned)share >> 28 & 1) ? (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x) / 36 % 7) * region : src + (size_t)(share > 1 ? (blockIdx.x & 7) + 8 * ((blockIdx.x >> 3) / share) : blockIdx.x) * region;
  const srd_t srd = make_srd(mine, region);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + wave * (G * 1024);
  const unsigned step = nw * 1024;                       // the workgroup's waves interleave 1-KiB pieces
  unsigned off = wave * 1024 + lane * 16, acc = 0;
  // MODE 2: the region is a row-major matrix with rows of 6144 B (a [R x 3072] bf16 operand); this workgroup's panel = 512 B of every row (column block
  // blockIdx.x % 12), a piece = rows r, r + 1 of the panel, the workgroup's waves walk down the rows
  // geometry packed in `share` for MODE 2: bits 0-15 row bytes, bits 16-27 panel width in bytes (512: a piece = 2 rows, 1024: a piece = 1 row, 256: 4 rows), bit 28: co-locate
  const unsigned rowb = MODE == 2 ? (unsigned)share & 0xffffu : 6144u, panelw = MODE == 2 ? ((unsigned)share >> 16) & 0xfffu : 512u;
  const unsigned ncb = rowb / panelw, rpp = 1024u / panelw;          // column blocks per row, rows per piece
  unsigned wg = blockIdx.x;
  if (MODE == 2 && ((unsigned)share >> 28 & 1)) { const unsigned q = gridDim.x >> 3, x = wg & 7; wg = x * q + (wg >> 3); }      // consecutive ids share an XCD (round-robin dispatch)
  if (MODE == 2) off = (wg % ncb) * panelw + (rpp * wave + lane / (panelw / 16)) * rowb + (lane % (panelw / 16)) * 16;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  u32x4 r[G];
#pragma unroll
  for (int g = 0; g < G; ++g) r[g] = (u32x4){0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (MODE == 0 || MODE == 2) glds16(srd, lds0 + g * 1024, off, 0u);
      else asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(r[g]) : "v"(off), "s"(srd) : "memory");
      if (MODE == 2) { off += rpp * nw * rowb; if (off >= region) off -= (unsigned)region; }
      else { off += step; if (off >= region) off -= (unsigned)region; }
    }
    if (MODE != 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G / 2) : "memory");      // keep half of them in flight
    else {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G / 2) : "memory");
#pragma unroll
      for (int g = 0; g < G / 2; ++g) acc += r[g][0];
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  if (acc == 0x12345u) sink[0] = acc + smem[lane];
}

// The same LDS-DMA stream from 4 loader waves (waves 0-3, one per SIMD) while waves 4-11 issue back-to-back MFMAs on registers (no memory traffic):
// does matrix work on the SIMDs slow the requests down?  (the GEMM loops get their pieces for ~60 cycles each, the bare stream above for ~20)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int G>
__global__ __launch_bounds__(768) void fill_mfma_kernel(const char* src, size_t region, int iters, unsigned long long* cycles, float* sink, int compute_waves) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __shared__ int done;
  if (threadIdx.x == 0) done = 0;
  __syncthreads();
  if (wave < 4) {
    const char* mine = src + (size_t)blockIdx.x * region;
    const srd_t srd = make_srd(mine, region);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + wave * (G * 1024);
    unsigned off = wave * 1024 + lane * 16;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        glds16(srd, lds0 + g * 1024, off, 0u);
        off += 4096;
        if (off >= region) off -= (unsigned)region;
      }
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G / 2) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (lane == 0) __hip_atomic_fetch_add(&done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  } else if (wave < 4 + compute_waves) {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(lane + j); b[j] = (__bf16)(float)(lane - j); }
    while (__hip_atomic_load(&done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4) {
#pragma unroll
      for (int k = 0; k < 16; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 1234.5f) sink[0] = acc[0][0];
  }
}
int main() {
  const int cus = 256;
  const size_t big = (size_t)cus * (16u << 20);
  char* src; unsigned long long* cyc; unsigned* sink;
  (void)hipMalloc(&src, big); (void)hipMemset(src, 1, big); (void)hipMalloc(&cyc, cus * 8); (void)hipMalloc(&sink, 64);
  std::vector<unsigned long long> h(cus);
  auto run = [&](auto kern, int waves, size_t region, int G, const char* what, int share = 1) {
    const int iters = 4096 / G * 4;
    const int lds = waves * G * 1024;                    // every wave its own G one-KiB landing slots (<= 160 KiB)
    if (lds > 160 * 1024) { printf("skip\n"); return; }
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int rep = 0; rep < 2; ++rep) {
      hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
      (void)hipEventRecord(a);
      hipLaunchKernelGGL(kern, dim3(cus), dim3(waves * 64), lds, 0, src, region, iters, cyc, sink, share);
      (void)hipEventRecord(b); (void)hipEventSynchronize(b);
      float ms; (void)hipEventElapsedTime(&ms, a, b);
      (void)hipMemcpy(h.data(), cyc, cus * 8, hipMemcpyDeviceToHost);
      double c = 0; for (auto v : h) c += v; c /= cus;
      const double bytes = (double)waves * iters * G * 1024;
      if (rep) printf("%-10s %2d waves  G=%2d  share %2d  region %6zu KiB: %6.1f B/clk/CU  (%.0f cycles per 1-KiB piece per CU)  %7.2f TB/s chip\n", what, waves, G, share, region >> 10,
                      bytes / c, c / (bytes / 1024), bytes * cus / (ms * 1e-3) / 1e12);
    }
  };
  for (size_t region : {(size_t)64 << 10, (size_t)16 << 20})
    for (int waves : {4, 8, 12, 16}) {
      run(fill_kernel<0, 8>, waves, region, 8, "LDS-DMA");
      run(fill_kernel<1, 8>, waves, region, 8, "registers");
    }
  // operand panels shared inside an XCD, streamed once (16 MiB per group): what a split-K weight-gradient tile row / column does
  for (int share : {1, 4, 12, 32}) run(fill_kernel<0, 16>, 4, (size_t)16 << 20, 16, "LDS-DMA", share);
  for (int share : {4, 12, 32}) run(fill_kernel<0, 16>, 4, (size_t)4 << 20, 16, "LDS-DMA", share);
  // GEMM-shaped: 7 row ranges (split-K) x 36 tiles; every workgroup streams the 512-B column block (blockIdx % 12) of its range's rows, two rows per piece:
  // panels shared by the 3 workgroups with equal blockIdx % 12 in a range, ranges of 48 MiB (8 192 rows x 6 144 B)
  // rows of 6144 B (R-like: 12 column blocks of 512 B, 3 sharers) / 1536 B (L-like: 3 column blocks, 12 sharers); pieces of 2 rows x 512 B, 1 row x 1 KiB, 4 rows x 256 B
  for (int coloc : {0, 1})
    for (unsigned rowb : {6144u, 1536u})
      for (unsigned pw : {512u, 1024u, 256u}) {
        if (rowb % pw) continue;
        const int geo = (int)(rowb | (pw << 16) | ((unsigned)coloc << 28));
        printf("rows of %u B, panel %u B, sharers co-located %d:  ", rowb, pw, coloc);
        run(fill_kernel<2, 16>, 4, (size_t)rowb * 8192, 16, "DMA-geom", geo);
      }
  run(fill_kernel<0, 16>, 4, (size_t)64 << 10, 16, "LDS-DMA");
  run(fill_kernel<0, 16>, 4, (size_t)16 << 20, 16, "LDS-DMA");
  for (int cw : {0, 4, 8}) {
    auto kern = fill_mfma_kernel<16>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    const int iters = 4096;
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(kern, dim3(cus), dim3(768), 64 * 1024, 0, src, (size_t)64 << 10, iters, cyc, (float*)sink, cw);
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(h.data(), cyc, cus * 8, hipMemcpyDeviceToHost);
      double c = 0; for (auto v : h) c += v; c /= cus;
      const double bytes = 4.0 * iters * 16 * 1024;
      if (rep) printf("LDS-DMA from 4 loader waves (L2-resident 64 KiB per CU) beside %d waves of back-to-back MFMAs: %6.1f B/clk/CU  (%.0f cycles per 1-KiB piece)\n", cw, bytes / c, c / (bytes / 1024));
    }
  }
  return 0;
}
