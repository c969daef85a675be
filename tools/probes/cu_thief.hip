// A stand-in for RCCL's persistent channel workgroups on a one-GPU box: N workgroups of 256 threads with an RCCL-like footprint
// (21 KiB of LDS, ~104 VGPRs) that spin for a given wall-clock time.  tools/cu_thief.py runs the training step beside it to see
// what N communication channels cost the compute kernels.  Build: hipcc --offload-arch=gfx950 -shared -fPIC cu_thief.hip -o libcuthief.so
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(256) void thief_kernel(unsigned long long ticks_100mhz, unsigned* sink) {
  __shared__ unsigned buf[21184 / 4];
  buf[threadIdx.x] = threadIdx.x;
  asm volatile("v_mov_b32 v103, 0" ::: "v103");          // claims registers v0..v103 for every wave
  const unsigned long long t0 = wall_clock64();
  unsigned acc = 0;
  while (wall_clock64() - t0 < ticks_100mhz) {
    acc += buf[(threadIdx.x + acc) & 255];
    __builtin_amdgcn_s_sleep(32);
  }
  if (acc == 0xffffffffu) sink[0] = acc;
}
extern "C" int thief_launch(int nwg, double milliseconds, void* sink, void* stream) {
  hipLaunchKernelGGL(thief_kernel, dim3(nwg), dim3(256), 0, (hipStream_t)stream, (unsigned long long)(milliseconds * 1e5), (unsigned*)sink);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
