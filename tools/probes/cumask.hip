// Which XCD does each bit of a HIP CU mask select on MI355X?  For a few masks, launch 2048 small workgroups on a masked stream and
// histogram HW_REG_XCC_ID (and count distinct (xcc, se, cu) slots seen).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <set>
__global__ void where(unsigned* out) {
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  for (volatile int i = 0; i < 2000; ++i) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid; }
}
int main() {
  unsigned* d; hipMalloc(&d, 2 * 4096 * 4);
  std::vector<unsigned> h(2 * 4096);
  auto run = [&](const char* name, std::vector<uint32_t> mask) {
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, mask.size(), mask.data()) != hipSuccess) { printf("%s: create failed\n", name); return; }
    hipMemsetAsync(d, 0xff, 2 * 4096 * 4, s);
    hipLaunchKernelGGL(where, dim3(4096), dim3(64), 0, s, d);
    hipStreamSynchronize(s);
    hipMemcpy(h.data(), d, 2 * 4096 * 4, hipMemcpyDeviceToHost);
    int hist[16] = {0}; std::set<unsigned long long> slots;
    for (int i = 0; i < 4096; ++i) { hist[h[2 * i] & 15]++; slots.insert(((unsigned long long)(h[2 * i] & 15) << 32) | (h[2 * i + 1] & 0xfff0ff00u)); }
    printf("%-28s xcc histogram:", name);
    for (int x = 0; x < 8; ++x) printf(" %4d", hist[x]);
    printf("   distinct (xcc,se,cu..) %zu\n", slots.size());
    hipStreamDestroy(s);
  };
  run("all 256", std::vector<uint32_t>(8, 0xffffffffu));
  run("bits 0-127", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0});
  run("bits 128-255", {0, 0, 0, 0, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu});
  run("bits 0-31", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0});
  run("bits 32-63", {0, 0xffffffffu, 0, 0, 0, 0, 0, 0});
  run("even bits", std::vector<uint32_t>(8, 0x55555555u));
  run("bits = 0..3 mod 8", std::vector<uint32_t>(8, 0x0f0f0f0fu));
  run("bits 0-7", {0xffu, 0, 0, 0, 0, 0, 0, 0});
  run("bit%8==0", std::vector<uint32_t>(8, 0x01010101u));
  return 0;
}
