// L2 -> LDS fill-rate micro-benchmark (authoring tool, not part of the library).
// Every workgroup (4 waves) streams the 32-deep K chunks of a 256-row fp16 X tile + a 64-row W tile (20 KB per chunk, the
// GEMM ring's stage image) into a two-stage LDS ring, three workgroups per CU, eight workgroups sharing each X tile like
// the column tiles of a GEMM.  Variants: A = global_load_lds_dwordx4 (LDS-DMA, what the ring kernels use),
// B = global_load_dwordx4 into VGPRs + ds_write_b128, C = 3 of 5 pieces by DMA and 2 through registers, D = 2 DMA + 3 reg.
// build: hipcc --offload-arch=gfx950 -O3 tools/lds_fill.hip -o /tmp/lds_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned short u16;
#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))
struct alignas(16) v16 { unsigned int x, y, z, w; };

template <int NDMA>
__global__ __launch_bounds__(256) void fill(const u16 *__restrict__ X, const u16 *__restrict__ W, int ld, int nchunk, int reps,
                                            unsigned int *sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // XCD-aware map (workgroup b runs on XCD b % 8): the eight column tiles of a row tile share one XCD's L2
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3, ct = q & 7, rt = (q >> 3) * 8 + xcd;
  // piece j of wave wv: rows (j * 4 + wv) * 16 + lane / 4, 16 bytes at column (lane & 3) * 8 of the chunk
  const u16 *gp[5];
#pragma unroll
  for (int j = 0; j < 4; ++j) gp[j] = X + ((size_t)rt * 256 + (j * 4 + wv) * 16 + (lane >> 2)) * ld + (lane & 3) * 8;
  gp[4] = W + ((size_t)ct * 64 + wv * 16 + (lane >> 2)) * ld + (lane & 3) * 8;
  constexpr int STAGE = 20 * 1024;
  unsigned int acc = 0;
  v16 r[5];
  for (int rep = 0; rep < reps; ++rep) {
    for (int kc = 0; kc < nchunk; ++kc) {
      const int st = kc & 1;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        if (j < NDMA)
          __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)(gp[j] + kc * 32),
                                           (LDS_AS void *)(smem + st * STAGE + (j * 4 + wv) * 1024), 16, 0, 0);
        else
          r[j] = *reinterpret_cast<const v16 *>(gp[j] + kc * 32);
      }
#pragma unroll
      for (int j = NDMA; j < 5; ++j)
        *reinterpret_cast<v16 *>(smem + st * STAGE + (j * 4 + wv) * 1024 + lane * 16) = r[j];
      __builtin_amdgcn_s_waitcnt(0);  // everything of this chunk landed (the ring kernels keep one chunk in flight instead)
      __syncthreads();
      if ((kc & 15) == 15) acc += *reinterpret_cast<const unsigned int *>(smem + st * STAGE + threadIdx.x * 16);
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

// pipelined form: the loads of chunk k + 1 are in flight while chunk k is awaited (counted vmcnt), like the ring kernels
template <int NDMA>
__global__ __launch_bounds__(256) void fill_pipe(const u16 *__restrict__ X, const u16 *__restrict__ W, int ld, int nchunk,
                                                 int reps, unsigned int *sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // XCD-aware map (workgroup b runs on XCD b % 8): the eight column tiles of a row tile share one XCD's L2
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3, ct = q & 7, rt = (q >> 3) * 8 + xcd;
  const u16 *gp[5];
#pragma unroll
  for (int j = 0; j < 4; ++j) gp[j] = X + ((size_t)rt * 256 + (j * 4 + wv) * 16 + (lane >> 2)) * ld + (lane & 3) * 8;
  gp[4] = W + ((size_t)ct * 64 + wv * 16 + (lane >> 2)) * ld + (lane & 3) * 8;
  constexpr int STAGE = 20 * 1024;
  unsigned int acc = 0;
  v16 r[5];
  auto issue = [&](int kc) {
    const int st = kc & 1;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      if (j < NDMA)
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)(gp[j] + kc * 32),
                                         (LDS_AS void *)(smem + st * STAGE + (j * 4 + wv) * 1024), 16, 0, 0);
      else
        r[j] = *reinterpret_cast<const v16 *>(gp[j] + kc * 32);
    }
  };
  for (int rep = 0; rep < reps; ++rep) {
    issue(0);
    for (int kc = 0; kc < nchunk; ++kc) {
      const int st = kc & 1;
      // register pieces of chunk kc: wait for them, park them in LDS, then put chunk kc + 1 in flight
      __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
#pragma unroll
      for (int j = NDMA; j < 5; ++j)
        *reinterpret_cast<v16 *>(smem + st * STAGE + (j * 4 + wv) * 1024 + lane * 16) = r[j];
      __syncthreads();
      if (kc + 1 < nchunk) issue(kc + 1);
      if ((kc & 15) == 15) acc += *reinterpret_cast<const unsigned int *>(smem + st * STAGE + threadIdx.x * 16);
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

// one-stage blocking fill with RB bytes per row and chunk (64 = half a 128-byte line, 128 = whole lines, 256 = two lines):
// does the cap move with the footprint of a request?
template <int RB>
__global__ __launch_bounds__(256) void fill_rb(const u16 *__restrict__ X, const u16 *__restrict__ W, int ld, int nchunk, int reps,
                                               unsigned int *sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3, ct = q & 7, rt = (q >> 3) * 8 + xcd;
  constexpr int NROW = RB <= 128 ? 320 : 160;  // stage <= 40 KB
  constexpr int NX = NROW * 4 / 5;
  constexpr int LPR = RB / 16, RPI = 64 / LPR, NI = NROW / RPI / 4;  // lanes per row, rows per wave instruction, instructions per wave
  unsigned int acc = 0;
  for (int rep = 0; rep < reps; ++rep)
    for (int kc = 0; kc < nchunk * 64 / RB * (320 / NROW); ++kc) {
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int row = (j * 4 + wv) * RPI + lane / LPR;  // 0..319: rows >= 256 are the W tile
        const u16 *src = (row < NX ? X + ((size_t)rt * 256 + row) * ld : W + ((size_t)ct * 64 + row - NX) * ld) + (lane % LPR) * 8 +
                         (kc % (512 * 2 / RB)) * (RB / 2);
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)src, (LDS_AS void *)(smem + (j * 4 + wv) * 1024), 16, 0, 0);
      }
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();
      if ((kc & 7) == 7) acc += *reinterpret_cast<const unsigned int *>(smem + threadIdx.x * 16);
    }
  if (acc == 0x12345678u) sink[0] = acc;
}

// chunk-major X: [K / 32][rows][32] fp16 -- the 16 rows x 64 B of one wave instruction are 1 KB of consecutive memory;
// same 20 KB stage image and 32-deep chunks as the ring kernels
__global__ __launch_bounds__(256) void fill_cm(const u16 *__restrict__ X, const u16 *__restrict__ W, int rows_total, int wrows,
                                               int nchunk, int reps, unsigned int *sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3, ct = q & 7, rt = (q >> 3) * 8 + xcd;
  const u16 *gp[5];
  size_t cs[5];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    gp[j] = X + ((size_t)rt * 256 + (j * 4 + wv) * 16 + (lane >> 2)) * 32 + (lane & 3) * 8;
    cs[j] = (size_t)rows_total * 32;
  }
  gp[4] = W + ((size_t)ct * 64 + wv * 16 + (lane >> 2)) * 32 + (lane & 3) * 8;
  cs[4] = (size_t)wrows * 32;
  unsigned int acc = 0;
  for (int rep = 0; rep < reps; ++rep)
    for (int kc = 0; kc < nchunk; ++kc) {
      const int st = kc & 1;
#pragma unroll
      for (int j = 0; j < 5; ++j)
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)(gp[j] + kc * cs[j]),
                                         (LDS_AS void *)(smem + st * 20480 + (j * 4 + wv) * 1024), 16, 0, 0);
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();
      if ((kc & 15) == 15) acc += *reinterpret_cast<const unsigned int *>(smem + st * 20480 + threadIdx.x * 16);
    }
  if (acc == 0x12345678u) sink[0] = acc;
}

// X straight into registers in the MFMA operand layout (no LDS for X: a wave's 64 rows are private to it), W by LDS-DMA:
// lane (row l & 31, half l >> 5) loads the 16 bytes k = 16 s + 8 half .. +7 of its row for k16-step s; chunk-major X
__global__ __launch_bounds__(256) void fill_xreg(const u16 *__restrict__ X, const u16 *__restrict__ W, int rows_total, int wrows,
                                                 int nchunk, int reps, unsigned int *sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3, ct = q & 7, rt = (q >> 3) * 8 + xcd;
  const u16 *xp[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) xp[rb] = X + ((size_t)rt * 256 + wv * 64 + rb * 32 + col) * 32 + half * 8;
  const u16 *wp = W + ((size_t)ct * 64 + wv * 16 + (lane >> 2)) * 32 + (lane & 3) * 8;
  const size_t xcs = (size_t)rows_total * 32, wcs = (size_t)wrows * 32;
  unsigned int acc = 0;
  v16 r[2][2][2];  // [buffer][rb][s]
  auto issue = [&](int kc, int buf) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) r[buf][rb][s2] = *reinterpret_cast<const v16 *>(xp[rb] + kc * xcs + s2 * 16);
    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)(wp + kc * wcs), (LDS_AS void *)(smem + (kc & 1) * 4096 + wv * 1024), 16, 0, 0);
  };
  for (int rep = 0; rep < reps; ++rep) {
    issue(0, 0);
    for (int kc = 0; kc < nchunk; ++kc) {
      __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
      __syncthreads();
      if (kc + 1 < nchunk) issue(kc + 1, (kc + 1) & 1);
      const int b = kc & 1;
      acc += r[b][0][0].x ^ r[b][0][1].y ^ r[b][1][0].z ^ r[b][1][1].w;
      if ((kc & 15) == 15) acc += *reinterpret_cast<const unsigned int *>(smem + b * 4096 + threadIdx.x * 16);
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// register-X pipeline as attn_tail_rx_kernel runs it (four chunks ahead, W through a five-stage LDS-DMA ring, one barrier per chunk),
// without the MFMAs.  FRAG = 0: chunk-major X [K/32][rows][32] (a wave instruction touches 32 B of each of 32 rows 64 B apart);
// FRAG = 1: FRAGMENT-major X [K/32][rows/32][2 k16-steps][64 lanes][8 halves] (a wave instruction reads 1 KB of consecutive memory)
template <int FRAG, int DEPTH>
__global__ __launch_bounds__(256, 2) void fill_xr_pipe(const u16 *__restrict__ X, const u16 *__restrict__ W, int rows_total, int wrows,
                                                       int nchunk, int reps, unsigned int *sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3, ct = q & 7, rt = (q >> 3) * 8 + xcd;
  const u16 *xp[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    if (FRAG) xp[rb] = X + ((size_t)(rt * 8 + wv * 2 + rb) * 2 * 64 + lane) * 8;   // 32-row block (rt*8 + wv*2 + rb), k16-step 0
    else xp[rb] = X + ((size_t)rt * 256 + wv * 64 + rb * 32 + col) * 32 + half * 8;
  }
  const u16 *wp = W + ((size_t)ct * 64 + wv * 16 + (lane >> 2)) * 32 + (lane & 3) * 8;
  const size_t xcs = (size_t)rows_total * 32, wcs = (size_t)wrows * 32;
  constexpr int S2OFF = FRAG ? 512 : 16;  // halves between the two k16-steps of a chunk
  unsigned int acc = 0;
  u32x4 r[DEPTH][2][2];
  auto issue = [&](int kc, u32x4 (&x)[2][2]) __attribute__((always_inline)) {
    const int k = kc < nchunk ? kc : nchunk - 1;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const u16 *p = xp[rb] + k * xcs;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(x[rb][0]) : "v"(p) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(x[rb][1]) : "v"(p + S2OFF) : "memory");
    }
    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)(wp + k * wcs), (LDS_AS void *)(smem + (kc % (DEPTH + 1)) * 4096 + wv * 1024), 16, 0, 0);
  };
  for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) issue(j, r[j]);
    for (int c0 = 0; c0 < nchunk; c0 += DEPTH) {
#pragma unroll
      for (int j = 0; j < DEPTH; ++j) {
        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r[j][0][0]), "+v"(r[j][0][1]), "+v"(r[j][1][0]), "+v"(r[j][1][1]) : "n"((DEPTH - 1) * 5) : "memory");
        __builtin_amdgcn_s_barrier();
        acc += r[j][0][0].x ^ r[j][0][1].y ^ r[j][1][0].z ^ r[j][1][1].w;
        acc += *reinterpret_cast<const unsigned int *>(smem + ((c0 + j) % (DEPTH + 1)) * 4096 + threadIdx.x * 16);
        issue(c0 + j + DEPTH, r[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < DEPTH; ++j)
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[j][0][0]), "+v"(r[j][0][1]), "+v"(r[j][1][0]), "+v"(r[j][1][1]) :: "memory");
    __syncthreads();
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <typename F>
void run(const char *name, F kern, int wgs, const u16 *X, const u16 *W, int ld, int nchunk, int reps, unsigned int *sink) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 48 * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 41 * 1024, 0, X, W, ld, nchunk, 1, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 41 * 1024, 0, X, W, ld, nchunk, reps, sink);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)wgs * nchunk * reps * 20480.0;
  printf("%-28s %4d WGs: %.1f us, %.2f TB/s into LDS, %.1f KB/us/CU, %.1f B/clk/CU (2.4 GHz)\n", name, wgs, ms * 1e3,
         bytes / ms / 1e9, bytes / (ms * 1e3) / 256 / 1024, bytes / (ms * 1e-3) / 256 / 2.4e9);
}

int main() {
  const int ld = 544, nchunk = 16, reps = 40, ntr = 96;
  u16 *X, *W; unsigned int *sink;
  CK(hipMalloc(&X, (size_t)ntr * 256 * ld * 2)); CK(hipMalloc(&W, (size_t)8 * 64 * ld * 2)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(X, 0, (size_t)ntr * 256 * ld * 2)); CK(hipMemset(W, 0, (size_t)8 * 64 * ld * 2));
  for (int wgs : {768, 512, 256}) {
    run("A  5 DMA (blocking)", fill<5>, wgs, X, W, ld, nchunk, reps, sink);
    run("B  5 reg (blocking)", fill<0>, wgs, X, W, ld, nchunk, reps, sink);
    run("A' 5 DMA (pipelined)", fill_pipe<5>, wgs, X, W, ld, nchunk, reps, sink);
    run("B' 5 reg (pipelined)", fill_pipe<0>, wgs, X, W, ld, nchunk, reps, sink);
    run("C' 3 DMA + 2 reg", fill_pipe<3>, wgs, X, W, ld, nchunk, reps, sink);
    run("D' 2 DMA + 3 reg", fill_pipe<2>, wgs, X, W, ld, nchunk, reps, sink);
    run("E' 4 DMA + 1 reg", fill_pipe<4>, wgs, X, W, ld, nchunk, reps, sink);
    {
      auto k = fill_cm;
      CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 48 * 1024));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 41 * 1024, 0, X, W, ntr * 256, 8 * 64, nchunk, 1, sink);
      CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 41 * 1024, 0, X, W, ntr * 256, 8 * 64, nchunk, reps, sink);
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double bytes = (double)wgs * nchunk * reps * 20480.0;
      printf("%-28s %4d WGs: %.1f us, %.2f TB/s into LDS, %.1f KB/us/CU, %.1f B/clk/CU (2.4 GHz)\n", "CM chunk-major [K/32][rows][32]", wgs,
             ms * 1e3, bytes / ms / 1e9, bytes / (ms * 1e3) / 256 / 1024, bytes / (ms * 1e-3) / 256 / 2.4e9);
    }
    {
      auto k = fill_xreg;
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 41 * 1024, 0, X, W, ntr * 256, 8 * 64, nchunk, 1, sink);
      CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 41 * 1024, 0, X, W, ntr * 256, 8 * 64, nchunk, reps, sink);
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double bytes = (double)wgs * nchunk * reps * 20480.0;
      printf("%-28s %4d WGs: %.1f us, %.2f TB/s, %.1f KB/us/CU, %.1f B/clk/CU (2.4 GHz)\n", "XR X->VGPR (MFMA layout) + W DMA", wgs,
             ms * 1e3, bytes / ms / 1e9, bytes / (ms * 1e3) / 256 / 1024, bytes / (ms * 1e-3) / 256 / 2.4e9);
    }
    for (int v = 0; v < 4; ++v) {
      typedef void (*KT)(const u16 *, const u16 *, int, int, int, int, unsigned int *);
      KT k = v == 0 ? fill_xr_pipe<0, 4> : v == 1 ? fill_xr_pipe<1, 4> : v == 2 ? fill_xr_pipe<0, 2> : fill_xr_pipe<1, 2>;
      const char *nm = v == 0 ? "XP chunk-major X->VGPR depth 4" : v == 1 ? "XF FRAGMENT-major X->VGPR depth 4" : v == 2 ? "XP chunk-major depth 2" : "XF FRAGMENT-major depth 2";
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 41 * 1024, 0, X, W, ntr * 256, 8 * 64, nchunk, 1, sink);
      CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 41 * 1024, 0, X, W, ntr * 256, 8 * 64, nchunk, reps, sink);
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double bytes = (double)wgs * nchunk * reps * 20480.0;
      printf("%-36s %4d WGs: %.1f us, %.2f TB/s, %.1f KB/us/CU, %.1f B/clk/CU (2.4 GHz)\n", nm, wgs,
             ms * 1e3, bytes / ms / 1e9, bytes / (ms * 1e3) / 256 / 1024, bytes / (ms * 1e-3) / 256 / 2.4e9);
    }
    run("R64  one stage, 64 B/row", fill_rb<64>, wgs, X, W, ld, nchunk, reps, sink);
    run("R128 one stage, 128 B/row", fill_rb<128>, wgs, X, W, ld, nchunk, reps, sink);
    run("R256 one stage, 256 B/row", fill_rb<256>, wgs, X, W, ld, nchunk, reps, sink);
  }
  return 0;
}
