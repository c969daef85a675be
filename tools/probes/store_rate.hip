// Store-path probe (MI355X): how fast can ONE CU / a fraction of the CUs / the whole chip drain an epilogue-shaped store burst?
// Each 512-thread workgroup writes `kb` KiB as 16-B-per-lane stores covering whole 512-B row segments (the GEMM epilogue's shape).
// build: hipcc -O3 --offload-arch=gfx950 store_rate.hip -o store_rate ; run: ./store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__global__ __launch_bounds__(512) void store_burst(char* out, int kb, int ld_bytes, int nt) {
  const int tid = threadIdx.x;
  char* base = out + (size_t)blockIdx.x * kb * 1024;          // private region per workgroup
  const int per_iter = 512 * 16;                                // 8 KiB per workgroup-wide store
  u32x4 v = {(unsigned)tid, 1u, 2u, 3u};
  for (int off = 0; off < kb * 1024; off += per_iter) {
    u32x4* p = (u32x4*)(base + off + tid * 16);
    if (nt) __builtin_nontemporal_store(v, p); else *p = v;
  }
}
__global__ __launch_bounds__(512) void spin(int cycles) {       // keeps the other CUs busy computing nothing
  long long t0 = clock64();
  while (clock64() - t0 < cycles) {}
}
int main() {
  const int kb = 320;                                            // ~ one 320x256 GELU tile (2 bf16 outputs)
  char* buf; hipMalloc(&buf, (size_t)4096 * kb * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int nt = 0; nt < 2; ++nt)
    for (int grid : {8, 32, 64, 128, 256, 512, 1024, 2048}) {
      for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(store_burst, dim3(grid), dim3(512), 0, 0, buf, kb, 0, nt);
      hipDeviceSynchronize();
      hipEventRecord(a);
      const int reps = 20;
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(store_burst, dim3(grid), dim3(512), 0, 0, buf, kb, 0, nt);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      const double us = ms * 1e3 / reps, bytes = (double)grid * kb * 1024;
      printf("nt=%d grid %5d: %8.1f us per launch  %7.2f TB/s  %6.1f GB/s per active CU (first round)\n", nt, grid, us, bytes / us / 1e6,
             bytes / us / 1e3 / (grid < 256 ? grid : 256));
    }
  return 0;
}
