// does hipExtAnyOrderLaunch let two independent kernels of ONE stream overlap on gfx950?  (authoring tool)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(unsigned long long ticks, int *sink) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (ticks == 1) sink[0] = 1;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
  int *sink; CK(hipMalloc(&sink, 4));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int flags = 0; flags < 2; ++flags) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < 10; ++i) {
        hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, 2000ULL, sink);                        // 20 us at 100 MHz
        hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, nullptr, nullptr, flags, 2000ULL, sink);  // independent twin
      }
      CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("flags=%d: 10 pairs of 20-us kernels: %.1f us (serial = 400+, overlapped pairs = 200+)\n", flags, ms * 1e3);
    }
  }
  return 0;
}
