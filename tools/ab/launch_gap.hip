// dependent-launch cost on 1 .. 4 streams (authoring tool): hipcc --offload-arch=gfx950 -O3 tools/ab/launch_gap.hip -o /tmp/launch_gap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ void spin(int cycles, int *sink) {
  const long long t0 = clock64();
  while (clock64() - t0 < cycles) {}
  if (cycles < 0) sink[0] = 1;
}
int main(int argc, char **argv) {
  int *sink; hipMalloc(&sink, 4);
  hipStream_t st[8];
  for (int i = 0; i < 8; ++i) hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
  const int N = 2000;
  for (int wgs : {1, 64, 512}) for (int cyc : {0, 10000, 40000}) for (int ns : {1, 2, 3, 4, 5, 6, 8}) {
    for (int w = 0; w < 2; ++w) {  // second pass timed
      hipDeviceSynchronize();
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < N; ++i) for (int s = 0; s < ns; ++s) hipLaunchKernelGGL(spin, dim3(wgs), dim3(256), 0, st[s], cyc, sink);
      hipDeviceSynchronize();
      auto t1 = std::chrono::steady_clock::now();
      if (w) printf("wgs %3d spin %5d cyc (%.1f us), %d streams: %.2f us per launch per stream\n", wgs, cyc, cyc / 100.0 / 1.0 * 0.01 * 100 / 100, ns,
                    std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
    }
  }
  return 0;
}
