// Store-pattern micro-benchmark (authoring tool, not part of the library):
// every workgroup writes one 256-row x 128-channel fp16 tile of a [rows][ld] matrix, the way the GEMM epilogue does,
// with different per-instruction footprints.  build: hipcc --offload-arch=gfx950 -O3 tools/ab/store_pattern.hip -o /tmp/sp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned short u16;
struct alignas(8) h4 { u16 a, b, c, d; };
struct alignas(16) h8 { u16 v[8]; };

// A: the current epilogue: lane (col = lane&31, half = lane>>5) stores 8 B at row col, channels 8q+4half  (16 B/row/instr)
__global__ __launch_bounds__(256) void pat_a(u16 *out, int ld, int ntr) {
  const int tile = blockIdx.x, tr = tile % ntr, tc = tile / ntr;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
  for (int cb = 0; cb < 4; ++cb)
    for (int rb = 0; rb < 2; ++rb) {
      const size_t row = (size_t)tr * 256 + wave * 64 + rb * 32 + col;
      for (int q = 0; q < 4; ++q) {
        h4 v = {(u16)row, (u16)cb, (u16)q, (u16)half};
        *(h4 *)(out + row * ld + tc * 128 + cb * 32 + 8 * q + 4 * half) = v;
      }
    }
}
// B: 64 B per row per instruction (4 lanes x 16 B), 16 rows per instruction
__global__ __launch_bounds__(256) void pat_b(u16 *out, int ld, int ntr) {
  const int tile = blockIdx.x, tr = tile % ntr, tc = tile / ntr;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int cb = 0; cb < 4; ++cb)
    for (int i = 0; i < 4; ++i) {
      const size_t row = (size_t)tr * 256 + wave * 64 + i * 16 + (lane >> 2);
      h8 v; for (int j = 0; j < 8; ++j) v.v[j] = (u16)(row + j);
      *(h8 *)(out + row * ld + tc * 128 + cb * 32 + 8 * (lane & 3)) = v;
    }
}
// C: 256 B per row per instruction (16 lanes x 16 B), 4 rows per instruction
__global__ __launch_bounds__(256) void pat_c(u16 *out, int ld, int ntr) {
  const int tile = blockIdx.x, tr = tile % ntr, tc = tile / ntr;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = 0; i < 16; ++i) {
    const size_t row = (size_t)tr * 256 + wave * 64 + i * 4 + (lane >> 4);
    h8 v; for (int j = 0; j < 8; ++j) v.v[j] = (u16)(row + j);
    *(h8 *)(out + row * ld + tc * 128 + 8 * (lane & 15)) = v;
  }
}
// D: 128 B per row per instruction (8 lanes x 16 B), 8 rows per instruction
__global__ __launch_bounds__(256) void pat_d(u16 *out, int ld, int ntr) {
  const int tile = blockIdx.x, tr = tile % ntr, tc = tile / ntr;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int cbp = 0; cbp < 2; ++cbp)
    for (int i = 0; i < 8; ++i) {
      const size_t row = (size_t)tr * 256 + wave * 64 + i * 8 + (lane >> 3);
      h8 v; for (int j = 0; j < 8; ++j) v.v[j] = (u16)(row + j);
      *(h8 *)(out + row * ld + tc * 128 + cbp * 64 + 8 * (lane & 7)) = v;
    }
}
// G: 32 B per row per instruction: lane (col, half) stores 16 B at channels 16p+8half (after a permlane32 swap)
__global__ __launch_bounds__(256) void pat_g(u16 *out, int ld, int ntr) {
  const int tile = blockIdx.x, tr = tile % ntr, tc = tile / ntr;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
  for (int cb = 0; cb < 4; ++cb)
    for (int rb = 0; rb < 2; ++rb) {
      const size_t row = (size_t)tr * 256 + wave * 64 + rb * 32 + col;
      for (int p = 0; p < 2; ++p) {
        h8 v; for (int j = 0; j < 8; ++j) v.v[j] = (u16)(row + j);
        *(h8 *)(out + row * ld + tc * 128 + cb * 32 + 16 * p + 8 * half) = v;
      }
    }
}
// F: 2 B per lane, 64 B per row, 2 rows per instruction (channel-per-lane accumulator layout)
__global__ __launch_bounds__(256) void pat_f(u16 *out, int ld, int ntr) {
  const int tile = blockIdx.x, tr = tile % ntr, tc = tile / ntr;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
  for (int cb = 0; cb < 4; ++cb)
    for (int rb = 0; rb < 2; ++rb)
      for (int r = 0; r < 16; ++r) {
        const size_t row = (size_t)tr * 256 + wave * 64 + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        out[row * ld + tc * 128 + cb * 32 + col] = (u16)(row + r);
      }
}
// E: like A but tile = 256 rows x 512 channels visited tile-column-major by one WG?  no: like A with XCD-swizzled tile id
__global__ __launch_bounds__(256) void pat_e(u16 *out, int ld, int ntr) {
  const int nt = gridDim.x, per = nt / 8;
  const int tile = (blockIdx.x % 8) * per + blockIdx.x / 8;
  const int tr = tile % ntr, tc = tile / ntr;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
  for (int cb = 0; cb < 4; ++cb)
    for (int rb = 0; rb < 2; ++rb) {
      const size_t row = (size_t)tr * 256 + wave * 64 + rb * 32 + col;
      for (int q = 0; q < 4; ++q) {
        h4 v = {(u16)row, (u16)cb, (u16)q, (u16)half};
        *(h4 *)(out + row * ld + tc * 128 + cb * 32 + 8 * q + 4 * half) = v;
      }
    }
}

template <typename F> static void run(const char *name, F k, u16 *out, int rows, int ld) {
  const int ntr = rows / 256, ntc = ld / 128;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(ntr * ntc), dim3(256), 0, 0, out, ld, ntr);
  hipEventRecord(e0, 0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(ntr * ntc), dim3(256), 0, 0, out, ld, ntr);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps, gb = (double)rows * ld * 2 / 1e9;
  printf("%s rows %d ld %d: %.1f us  %.0f GB/s\n", name, rows, ld, us, gb / (us * 1e-6));
}

int main(int argc, char **argv) {
  const int rows = 65536;
  u16 *out; hipMalloc(&out, (size_t)rows * 1024 * 2 * 2);
  for (int ld : {128, 256, 512, 1024}) {
    run("A 16B/row ", pat_a, out, rows, ld);
    run("E A+xcd   ", pat_e, out, rows, ld);
    run("G 32B/row ", pat_g, out, rows, ld);
    run("F 2B lane ", pat_f, out, rows, ld);
    run("B 64B/row ", pat_b, out, rows, ld);
    run("D 128B/row", pat_d, out, rows, ld);
    run("C 256B/row", pat_c, out, rows, ld);
  }
  return 0;
}
