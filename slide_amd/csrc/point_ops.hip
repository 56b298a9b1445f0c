// point_ops.hip -- gfx950 kernels behind the `pointnet2_ops._ext` operator API (+ pytorch3d knn).
//
// Written for CDNA4: 64-wide wavefronts, one workgroup per (batch element, tile of queries) instead of
// the reference's one-block-per-batch-element launches, coalesced channel-minor traffic, LDS-staged
// search sets.  Index semantics are bit-exact with oracle/ops_cpu.c (which restates
// _ext-src/src/*.cu); distances use the shared recipe fmaf(dz,dz,fmaf(dy,dy,dx*dx)) and this file is
// compiled with -ffp-contract=off so nothing else is contracted.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/slide_hip.h"

#define LAUNCH_STATUS() ((int)hipGetLastError())

namespace {

__device__ __forceinline__ float sqdist3(float ax, float ay, float az, float bx, float by, float bz) {
  const float dx = ax - bx, dy = ay - by, dz = az - bz;
  return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

// ---------------------------------------------------------------------------------------------- gather
// out[b,c,j] = points[b,c,idx[b,j]]   (_ext-src/src/sampling_gpu.cu:8-20)
__global__ void gather_points_kernel(int c, int n, int m, const float *__restrict__ points,
                                     const int *__restrict__ idx, float *__restrict__ out) {
  const int b = blockIdx.z;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const int a = idx[(size_t)b * m + j];
  for (int l = blockIdx.y; l < c; l += gridDim.y)
    out[((size_t)b * c + l) * m + j] = points[((size_t)b * c + l) * n + a];
}

// Four consecutive outputs per thread: the index quad is read once (16 B), every channel costs four 4-byte gathers
// (the point row is L2-/L1-resident) and ONE 16-byte non-temporal store -- the stores are what the op moves through HBM.
typedef float pf4 __attribute__((ext_vector_type(4)));
typedef int pi4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void gather_points_v4_kernel(int c, int n, int m, const float *__restrict__ points,
                                                               const int *__restrict__ idx, float *__restrict__ out) {
  const int b = blockIdx.z;
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (j >= m) return;
  const pi4 a = *reinterpret_cast<const pi4 *>(idx + (size_t)b * m + j);
  for (int l = blockIdx.y; l < c; l += gridDim.y) {
    const float *p = points + ((size_t)b * c + l) * n;
    const pf4 v = {p[a[0]], p[a[1]], p[a[2]], p[a[3]]};
    __builtin_nontemporal_store(v, reinterpret_cast<pf4 *>(out + ((size_t)b * c + l) * m + j));
  }
}

// grad_points[b,c,idx[b,j]] += grad_out[b,c,j]   (sampling_gpu.cu:34-47)
__global__ void gather_points_grad_kernel(int c, int n, int m, const float *__restrict__ grad_out,
                                          const int *__restrict__ idx, float *__restrict__ grad_points) {
  const int b = blockIdx.z;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const int a = idx[(size_t)b * m + j];
  for (int l = blockIdx.y; l < c; l += gridDim.y)
    atomicAdd(grad_points + ((size_t)b * c + l) * n + a, grad_out[((size_t)b * c + l) * m + j]);
}

// ---------------------------------------------------------------------------------------------- FPS
// Reference: sampling_gpu.cu:59-173.  The reference's answer depends on its block size bs =
// opt_n_threads(n) only through the ORDER in which ties are resolved: per lane the smallest k wins,
// across lanes the shared-memory tree keeps the lower slot, i.e. the lane with the smallest
// bit-reversed id.  We encode that order in a 64-bit key (value bits | ~rank) and take a plain max, so
// the thread->point mapping here is free (coalesced, points and running distances live in registers,
// the candidate set is staged once in LDS) and one barrier per iteration suffices.
__device__ __forceinline__ unsigned fps_rank(int k, int bs, int lg, int L) {
  const unsigned t = (unsigned)k & (unsigned)(bs - 1);
  const unsigned rev = lg ? (__brev(t) >> (32 - lg)) : 0u;
  return rev * (unsigned)L + (unsigned)(k / bs);
}
__device__ __forceinline__ int fps_unrank(unsigned r, int bs, int lg, int L) {
  const unsigned rb = r / (unsigned)L, kk = r % (unsigned)L;
  const unsigned t = lg ? (__brev(rb) >> (32 - lg)) : 0u;
  return (int)(kk * (unsigned)bs + t);
}

// max of a 64-bit key over the wave, result in every lane: DPP row operations inside each row of 16 lanes (no LDS
// round trip per step as with ds_bpermute shuffles), one ds_swizzle across the rows of a half, scalar lanes 0 / 32 last.
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#define SLIDE_KEY_STEP(MOVE)                                                      \
  {                                                                                \
    const unsigned hi = (unsigned)(MOVE((int)(v >> 32)));                          \
    const unsigned lo = (unsigned)(MOVE((int)(unsigned)v));                        \
    const unsigned long long o = ((unsigned long long)hi << 32) | lo;              \
    v = o > v ? o : v;                                                             \
  }
#define SLIDE_DPP_B1(x) __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true)
#define SLIDE_DPP_4E(x) __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true)
#define SLIDE_DPP_141(x) __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true)
#define SLIDE_DPP_140(x) __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true)
#define SLIDE_SWZ_16(x) __builtin_amdgcn_ds_swizzle(x, 0x401F)
  SLIDE_KEY_STEP(SLIDE_DPP_B1)
  SLIDE_KEY_STEP(SLIDE_DPP_4E)
  SLIDE_KEY_STEP(SLIDE_DPP_141)
  SLIDE_KEY_STEP(SLIDE_DPP_140)
  SLIDE_KEY_STEP(SLIDE_SWZ_16)
#undef SLIDE_KEY_STEP
#undef SLIDE_DPP_B1
#undef SLIDE_DPP_4E
#undef SLIDE_DPP_141
#undef SLIDE_DPP_140
#undef SLIDE_SWZ_16
  const unsigned long long a = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(v >> 32), 0) << 32) |
                               (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 0);
  const unsigned long long c = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(v >> 32), 32) << 32) |
                               (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 32);
  return a > c ? a : c;
}

template <int PPT, int NT>
__global__ __launch_bounds__(NT) void fps_kernel(int n, int m, int bs, int lg, int L, int use_lds, int skip_origin,
                                                 const int *__restrict__ start, const float *__restrict__ dataset,
                                                 float *__restrict__ temp, int *__restrict__ idxs) {
  extern __shared__ __attribute__((aligned(16))) float spts[];
  __shared__ unsigned long long wkeys[2][NT / 64 > 0 ? NT / 64 : 1];
  const int b = blockIdx.x, tid = threadIdx.x;
  dataset += (size_t)b * n * 3;
  temp += (size_t)b * n;
  idxs += (size_t)b * m;

  float px[PPT], py[PPT], pz[PPT], td[PPT];
  unsigned low[PPT];
  bool valid[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = tid + i * NT;
    valid[i] = false;
    px[i] = py[i] = pz[i] = 0.f;
    td[i] = 0.f;
    low[i] = 0;
    if (k < n) {
      px[i] = dataset[k * 3 + 0];
      py[i] = dataset[k * 3 + 1];
      pz[i] = dataset[k * 3 + 2];
      td[i] = temp[k];
      const float mag = fmaf(pz[i], pz[i], fmaf(py[i], py[i], px[i] * px[i]));
      valid[i] = !skip_origin || !((double)mag <= 1e-3);  // sampling_gpu.cu:100-101 (double literal)
      low[i] = 0xFFFFFFFFu - fps_rank(k, bs, lg, L);
      if (use_lds) {
        spts[k * 3 + 0] = px[i];
        spts[k * 3 + 1] = py[i];
        spts[k * 3 + 2] = pz[i];
      }
    }
  }
  const float *src = use_lds ? spts : dataset;
  int old = start ? min(max(start[b], 0), n - 1) : 0;  // (clamped: a start index outside [0, n) must not read out of bounds)
  if (tid == 0) idxs[0] = old;
  __syncthreads();
  for (int j = 1; j < m; ++j) {
    const float x1 = src[old * 3 + 0], y1 = src[old * 3 + 1], z1 = src[old * 3 + 2];
    unsigned long long best = 0ull;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      if (valid[i]) {
        const float d = sqdist3(px[i], py[i], pz[i], x1, y1, z1);
        const float d2 = fminf(d, td[i]);
        td[i] = d2;
        const unsigned long long key =
            ((unsigned long long)(__float_as_uint(d2) + 1u) << 32) | (unsigned long long)low[i];
        best = key > best ? key : best;
      }
    }
    best = wave_max_u64(best);
    if (NT > 64) {
      if ((tid & 63) == 0) wkeys[j & 1][tid >> 6] = best;
      __syncthreads();
#pragma unroll
      for (int w = 0; w < NT / 64; ++w) {
        const unsigned long long o = wkeys[j & 1][w];
        best = o > best ? o : best;
      }
    }
    old = best == 0ull ? 0 : fps_unrank(0xFFFFFFFFu - (unsigned)best, bs, lg, L);
    if (tid == 0) idxs[j] = old;
  }
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = tid + i * NT;
    if (k < n && valid[i]) temp[k] = td[i];
  }
}

// n > 8192: running distances stay in global memory (rarely used; decode tops out at 4096 points)
__global__ __launch_bounds__(1024) void fps_kernel_large(int n, int m, int bs, int lg, int L, int skip_origin,
                                                         const int *__restrict__ start,
                                                         const float *__restrict__ dataset,
                                                         float *__restrict__ temp, int *__restrict__ idxs) {
  __shared__ unsigned long long wkeys[2][16];
  const int b = blockIdx.x, tid = threadIdx.x;
  dataset += (size_t)b * n * 3;
  temp += (size_t)b * n;
  idxs += (size_t)b * m;
  int old = start ? min(max(start[b], 0), n - 1) : 0;  // (clamped: a start index outside [0, n) must not read out of bounds)
  if (tid == 0) idxs[0] = old;
  for (int j = 1; j < m; ++j) {
    const float x1 = dataset[old * 3 + 0], y1 = dataset[old * 3 + 1], z1 = dataset[old * 3 + 2];
    unsigned long long best = 0ull;
    for (int k = tid; k < n; k += 1024) {
      const float x2 = dataset[k * 3 + 0], y2 = dataset[k * 3 + 1], z2 = dataset[k * 3 + 2];
      const float mag = fmaf(z2, z2, fmaf(y2, y2, x2 * x2));
      if (skip_origin && (double)mag <= 1e-3) continue;
      const float d2 = fminf(sqdist3(x2, y2, z2, x1, y1, z1), temp[k]);
      temp[k] = d2;
      const unsigned long long key = ((unsigned long long)(__float_as_uint(d2) + 1u) << 32) |
                                     (unsigned long long)(0xFFFFFFFFu - fps_rank(k, bs, lg, L));
      best = key > best ? key : best;
    }
    best = wave_max_u64(best);
    if ((tid & 63) == 0) wkeys[j & 1][tid >> 6] = best;
    __syncthreads();
    for (int w = 0; w < 16; ++w) {
      const unsigned long long o = wkeys[j & 1][w];
      best = o > best ? o : best;
    }
    old = best == 0ull ? 0 : fps_unrank(0xFFFFFFFFu - (unsigned)best, bs, lg, L);
    if (tid == 0) idxs[j] = old;
  }
}

// ---------------------------------------------------------------------------------------------- search kernels: block map
// XCD-aware (batch element, query tile) map of the search kernels (round 6; VERDICT r5 item 9: ball_query moved 6.5x its algorithmic
// bytes at n = 8192).  Every workgroup of a batch element streams that element's WHOLE search set; consecutive workgroup ids go to
// consecutive XCDs (MI355X_MICROARCH.md: block b runs on XCD b % 8 -- relied on for speed only), so with a (tile, b) grid the eight
// query tiles of an element ran on eight XCDs and each private L2 fetched the set once more.  Here the grid is linear and id L ->
// XCD L % 8 takes the elements b = 8 k + L % 8: all query tiles of an element run on ONE XCD and its set is fetched once.
struct SearchBlock { int b, tile; bool valid; };
__device__ __forceinline__ SearchBlock search_block(int nb, int tiles) {
  const int L = blockIdx.x, x = L & 7, q = L >> 3;
  SearchBlock r;
  r.tile = q % tiles;
  r.b = (q / tiles) * 8 + x;
  r.valid = r.b < nb;
  return r;
}
static inline unsigned search_grid(int nb, int tiles) { return (unsigned)(8 * ((nb + 7) / 8) * tiles); }

// ---------------------------------------------------------------------------------------------- ball query
// ball_query_gpu.cu:9-47
constexpr int BQ_TILE = 1024;  // (three_nn's LDS tile)
// WAVEFRONT SCAN (round 6).  Rounds 1-5 ran one THREAD per query over LDS-broadcast points: a wave ran until its slowest query was
// done, and a query's nsample result slots -- 4 nsample contiguous bytes -- were written one at a time, hits apart, so every store
// instruction touched 4 bytes of 64 different lines and a line left the L2 half-written many times (4.8x the algorithmic bytes at
// n = 8192, profiles/r05_ops_roofline.md).  Here a WAVE owns 64 queries and walks the search set in chunks of 256 points held in
// REGISTERS (four points per lane, loaded coalesced; no LDS, no workgroup barrier): for every unfinished query the chunk is
// evaluated in four 64-wide steps against the query's coordinates (scalar registers), the hits of a step are ranked by a ballot
// prefix count and written side by side, and a query leaves the scan the step its ball is full.  A wave's 64 queries keep their
// state (coordinates, count, first hit) in the lanes of five registers.  Same result as the reference's sequential scan
// (ball_query_gpu.cu:25-46): hits in index order, the first hit in every unused slot, zero rows / zero counts for empty balls (the
// host wrapper zero-fills idx and counts).
// STAGED (nsample <= 32, n < 65536): a query's hits arrive spread over the whole scan, so written in place its row of idx would leave
// the L2 half-written again and again; the wave builds its 64 x nsample block in LDS as 16-bit indices (4 KB: eight workgroups per
// CU as without it -- 32-bit staging at five per CU measured 20 % slower) and stores it once, coalesced, at the end.
constexpr int BQ_PPL = 4;  // points per lane and chunk
template <bool STAGED>
__global__ __launch_bounds__(256) void ball_query_kernel(int nb, int n, int m, float radius2, int nsample,
                                                         const float *__restrict__ new_xyz,
                                                         const float *__restrict__ xyz, int *__restrict__ idx,
                                                         int *__restrict__ counts) {
  const SearchBlock sb = search_block(nb, (m + 255) / 256);
  if (!sb.valid) return;
  const int b = sb.b, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q0 = sb.tile * 256 + wave * 64;  // this wave's queries q0 .. q0 + 63; lane i keeps the state of query q0 + i
  const int nq = min(64, m - q0);
  if (nq <= 0) return;
  extern __shared__ __attribute__((aligned(16))) unsigned short bq_stage[];  // STAGED: [4 waves][64 queries][nsample]
  unsigned short *const stg = bq_stage + wave * 64 * nsample;
  if (STAGED)
    for (int e = lane; e < 64 * nsample; e += 64) stg[e] = 0;  // (wave-private: no barrier, LDS operations of a wave are in order)
  xyz += (size_t)b * n * 3;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (lane < nq) {
    const float *q = new_xyz + ((size_t)b * m + q0 + lane) * 3;
    qx = q[0]; qy = q[1]; qz = q[2];
  }
  int cntv = lane < nq ? 0 : nsample, firstv = 0;  // (lanes without a query count as full)
  for (int k0 = 0; k0 < n; k0 += 64 * BQ_PPL) {
    if (__ballot(cntv < nsample) == 0) break;  // every ball of this wave is full
    float px[BQ_PPL], py[BQ_PPL], pz[BQ_PPL];
#pragma unroll
    for (int u = 0; u < BQ_PPL; ++u) {
      const int k = k0 + u * 64 + lane;
      const float *c = xyz + (size_t)(k < n ? k : n - 1) * 3;
      px[u] = c[0]; py[u] = c[1]; pz[u] = c[2];
    }
    for (int i = 0; i < nq; ++i) {
      int cnt = __builtin_amdgcn_readlane(cntv, i);
      if (cnt >= nsample) continue;
      const float nx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qx), i));
      const float ny = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qy), i));
      const float nz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qz), i));
      bool hit[BQ_PPL];
      unsigned long long mask[BQ_PPL], any = 0;
#pragma unroll
      for (int u = 0; u < BQ_PPL; ++u) {
        hit[u] = k0 + u * 64 + lane < n && sqdist3(nx, ny, nz, px[u], py[u], pz[u]) < radius2;
        mask[u] = __ballot(hit[u]);
        any |= mask[u];
      }
      if (any == 0) continue;
      int first = __builtin_amdgcn_readlane(firstv, i);
      int *const oi = idx + ((size_t)b * m + q0 + i) * nsample;
      unsigned short *const os = stg + i * nsample;
#pragma unroll
      for (int u = 0; u < BQ_PPL; ++u) {
        if (mask[u] == 0 || cnt >= nsample) continue;
        if (cnt == 0) first = k0 + u * 64 + (int)__builtin_ctzll(mask[u]);
        const int r = cnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask[u] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask[u], 0u));
        if (hit[u] && r < nsample) {
          if (STAGED) os[r] = (unsigned short)(k0 + u * 64 + lane);
          else oi[r] = k0 + u * 64 + lane;
        }
        cnt += (int)__builtin_popcountll(mask[u]);
      }
      cnt = cnt < nsample ? cnt : nsample;
      cntv = lane == i ? cnt : cntv;
      firstv = lane == i ? first : firstv;
    }
  }
  if (lane < nq && cntv > 0) counts[(size_t)b * m + q0 + lane] = cntv;
  // the unused slots of a non-empty ball repeat its first hit
  for (int i = 0; i < nq; ++i) {
    const int cnt = __builtin_amdgcn_readlane(cntv, i), first = __builtin_amdgcn_readlane(firstv, i);
    if (cnt == 0 || cnt >= nsample) continue;
    int *const oi = idx + ((size_t)b * m + q0 + i) * nsample;
    for (int l = cnt + lane; l < nsample; l += 64) {
      if (STAGED) stg[i * nsample + l] = (unsigned short)first;
      else oi[l] = first;
    }
  }
  if (STAGED) {
    int *const ob = idx + ((size_t)b * m + q0) * nsample;
    for (int e = lane; e < nq * nsample; e += 64) ob[e] = (int)stg[e];
  }
}

// ---------------------------------------------------------------------------------------------- grouping
// out[b,l,j,k] = points[b,l,idx[b,j,k]]  (group_points_gpu.cu:8-28).  One thread per (j,k) slot keeps its
// index in a register and walks a chunk of channels: index reads and output writes are coalesced.
__global__ __launch_bounds__(256) void group_points_kernel(int c, int n, int slots, int cchunk,
                                                           const float *__restrict__ points,
                                                           const int *__restrict__ idx, float *__restrict__ out) {
  const int b = blockIdx.z;
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= slots) return;
  const int ii = idx[(size_t)b * slots + s];
  const int l0 = blockIdx.y * cchunk, l1 = min(c, l0 + cchunk);
  const float *p = points + (size_t)b * c * n;
  float *o = out + (size_t)b * c * slots;
  for (int l = l0; l < l1; ++l) o[(size_t)l * slots + s] = p[(size_t)l * n + ii];
}

__global__ __launch_bounds__(256) void group_points_v4_kernel(int c, int n, int slots, int cchunk,
                                                              const float *__restrict__ points,
                                                              const int *__restrict__ idx, float *__restrict__ out) {
  const int b = blockIdx.z;
  const int s = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (s >= slots) return;
  const pi4 ii = *reinterpret_cast<const pi4 *>(idx + (size_t)b * slots + s);
  const int l0 = blockIdx.y * cchunk, l1 = min(c, l0 + cchunk);
  const float *p = points + (size_t)b * c * n;
  float *o = out + (size_t)b * c * slots + s;
#pragma unroll 2
  for (int l = l0; l < l1; ++l) {
    const float *pl = p + (size_t)l * n;
    const pf4 v = {pl[ii[0]], pl[ii[1]], pl[ii[2]], pl[ii[3]]};
    __builtin_nontemporal_store(v, reinterpret_cast<pf4 *>(o + (size_t)l * slots));
  }
}

// Row-in-LDS gather (group_points, and gather_points = the ns == 1 case): a block owns QPT*NT*4 consecutive output slots
// of one batch element -- their indices stay in registers -- and walks a chunk of channels.  Per channel the n-float point
// row is staged in LDS with coalesced 16-byte loads (read from L2 once per block instead of 4 bytes per gathered element),
// gathered with ds_read_b32, and written with 16-byte non-temporal stores: the op then moves little more than its
// output through HBM.  QPT = index quads per thread.
template <int QPT, int NT = 256>
__global__ __launch_bounds__(NT) void group_points_lds_kernel(int c, int n, int slots, int cchunk,
                                                               const float *__restrict__ points,
                                                               const int *__restrict__ idx, float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float row[];
  const int b = blockIdx.z;
  const int s0 = blockIdx.x * (QPT * NT * 4);
  pi4 ii[QPT];
#pragma unroll
  for (int q = 0; q < QPT; ++q) {
    const int s = s0 + (q * NT + threadIdx.x) * 4;
    ii[q] = s < slots ? *reinterpret_cast<const pi4 *>(idx + (size_t)b * slots + s) : pi4{0, 0, 0, 0};
  }
  const int l0 = blockIdx.y * cchunk, l1 = min(c, l0 + cchunk);
  const float *p = points + (size_t)b * c * n;
  float *o = out + (size_t)b * c * slots;
  for (int l = l0; l < l1; ++l) {
    __syncthreads();  // the previous channel's gathers are done
    const float *pl = p + (size_t)l * n;
    if ((n & 3) == 0) {
      for (int i = threadIdx.x * 4; i < n; i += NT * 4) *reinterpret_cast<pf4 *>(row + i) = *reinterpret_cast<const pf4 *>(pl + i);
    } else {
      for (int i = threadIdx.x; i < n; i += NT) row[i] = pl[i];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < QPT; ++q) {
      const int s = s0 + (q * NT + threadIdx.x) * 4;
      if (s < slots) {
        const pf4 v = {row[ii[q][0]], row[ii[q][1]], row[ii[q][2]], row[ii[q][3]]};
        __builtin_nontemporal_store(v, reinterpret_cast<pf4 *>(o + (size_t)l * slots + s));
      }
    }
  }
}

__global__ __launch_bounds__(256) void group_points_grad_kernel(int c, int n, int slots, int cchunk,
                                                                const float *__restrict__ grad_out,
                                                                const int *__restrict__ idx,
                                                                float *__restrict__ grad_points) {
  const int b = blockIdx.z;
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= slots) return;
  const int ii = idx[(size_t)b * slots + s];
  const int l0 = blockIdx.y * cchunk, l1 = min(c, l0 + cchunk);
  const float *g = grad_out + (size_t)b * c * slots;
  float *gp = grad_points + (size_t)b * c * n;
  for (int l = l0; l < l1; ++l) atomicAdd(gp + (size_t)l * n + ii, g[(size_t)l * slots + s]);
}

// ---------------------------------------------------------------------------------------------- three_nn
// interpolate_gpu.cu:9-59.  (double 1e40 bests == +inf in float for every comparison that can occur.)
__global__ __launch_bounds__(256) void three_nn_kernel(int nb, int n, int m, const float *__restrict__ unknown,
                                                       const float *__restrict__ known,
                                                       float *__restrict__ dist2, int *__restrict__ idx) {
  __shared__ float4 tile[BQ_TILE];
  const SearchBlock sb = search_block(nb, (n + 255) / 256);
  if (!sb.valid) return;
  const int b = sb.b;
  const int j = sb.tile * blockDim.x + threadIdx.x;
  known += (size_t)b * m * 3;
  const bool active = j < n;
  float ux = 0.f, uy = 0.f, uz = 0.f;
  if (active) {
    const float *u = unknown + ((size_t)b * n + j) * 3;
    ux = u[0]; uy = u[1]; uz = u[2];
  }
  float best1 = INFINITY, best2 = INFINITY, best3 = INFINITY;
  int besti1 = 0, besti2 = 0, besti3 = 0;
  for (int t0 = 0; t0 < m; t0 += BQ_TILE) {
    const int tn = min(BQ_TILE, m - t0);
    __syncthreads();
    for (int i = threadIdx.x; i < tn; i += blockDim.x) {
      const float *c = known + (size_t)(t0 + i) * 3;
      tile[i] = make_float4(c[0], c[1], c[2], 0.f);
    }
    __syncthreads();
    if (active) {
      for (int k = 0; k < tn; ++k) {
        const float4 c = tile[k];
        const float d = sqdist3(ux, uy, uz, c.x, c.y, c.z);
        const int kk = t0 + k;
        if (d < best1) {
          best3 = best2; besti3 = besti2;
          best2 = best1; besti2 = besti1;
          best1 = d; besti1 = kk;
        } else if (d < best2) {
          best3 = best2; besti3 = besti2;
          best2 = d; besti2 = kk;
        } else if (d < best3) {
          best3 = d; besti3 = kk;
        }
      }
    }
  }
  if (active) {
    float *od = dist2 + ((size_t)b * n + j) * 3;
    int *oi = idx + ((size_t)b * n + j) * 3;
    od[0] = best1; od[1] = best2; od[2] = best3;
    oi[0] = besti1; oi[1] = besti2; oi[2] = besti3;
  }
}

// out[b,l,j] = p[l,i1]*w1 + p[l,i2]*w2 + p[l,i3]*w3 (interpolate_gpu.cu:72-101), shared fma recipe.
__global__ __launch_bounds__(256) void three_interpolate_kernel(int c, int m, int n, int cchunk,
                                                                const float *__restrict__ points,
                                                                const int *__restrict__ idx,
                                                                const float *__restrict__ weight,
                                                                float *__restrict__ out) {
  const int b = blockIdx.z;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int *id = idx + ((size_t)b * n + j) * 3;
  const float *w = weight + ((size_t)b * n + j) * 3;
  const int i1 = id[0], i2 = id[1], i3 = id[2];
  const float w1 = w[0], w2 = w[1], w3 = w[2];
  const int l0 = blockIdx.y * cchunk, l1 = min(c, l0 + cchunk);
  const float *p = points + (size_t)b * c * m;
  float *o = out + (size_t)b * c * n;
  for (int l = l0; l < l1; ++l) {
    const float *pl = p + (size_t)l * m;
    o[(size_t)l * n + j] = fmaf(pl[i3], w3, fmaf(pl[i2], w2, pl[i1] * w1));
  }
}

__global__ __launch_bounds__(256) void three_interpolate_v4_kernel(int c, int m, int n, int cchunk,
                                                                   const float *__restrict__ points,
                                                                   const int *__restrict__ idx,
                                                                   const float *__restrict__ weight,
                                                                   float *__restrict__ out) {
  const int b = blockIdx.z;
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (j >= n) return;
  const pi4 *id = reinterpret_cast<const pi4 *>(idx + ((size_t)b * n + j) * 3);
  const pf4 *w = reinterpret_cast<const pf4 *>(weight + ((size_t)b * n + j) * 3);
  const pi4 ia = id[0], ib = id[1], ic = id[2];  // (i1 i2 i3 | i1') (i2' i3' | i1'' i2'') (i3'' | i1''' i2''' i3''')
  const pf4 wa = w[0], wb = w[1], wc = w[2];
  const int l0 = blockIdx.y * cchunk, l1 = min(c, l0 + cchunk);
  const float *p = points + (size_t)b * c * m;
  float *o = out + (size_t)b * c * n + j;
#pragma unroll 2
  for (int l = l0; l < l1; ++l) {
    const float *pl = p + (size_t)l * m;
    pf4 v;
    v[0] = fmaf(pl[ia[2]], wa[2], fmaf(pl[ia[1]], wa[1], pl[ia[0]] * wa[0]));
    v[1] = fmaf(pl[ib[1]], wb[1], fmaf(pl[ib[0]], wb[0], pl[ia[3]] * wa[3]));
    v[2] = fmaf(pl[ic[0]], wc[0], fmaf(pl[ib[3]], wb[3], pl[ib[2]] * wb[2]));
    v[3] = fmaf(pl[ic[3]], wc[3], fmaf(pl[ic[2]], wc[2], pl[ic[1]] * wc[1]));
    __builtin_nontemporal_store(v, reinterpret_cast<pf4 *>(o + (size_t)l * n));
  }
}

// three_interpolate with the m-float feature row of the current channel staged in LDS (same scheme as
// group_points_lds_kernel): 2 output quads per thread, their 24 indices and weights in registers.
__global__ __launch_bounds__(256) void three_interpolate_lds_kernel(int c, int m, int n, int cchunk,
                                                                    const float *__restrict__ points,
                                                                    const int *__restrict__ idx,
                                                                    const float *__restrict__ weight,
                                                                    float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float row[];
  const int b = blockIdx.z;
  const int j0 = blockIdx.x * 2048;
  pi4 ia[2], ib[2], ic[2];
  pf4 wa[2], wb[2], wc[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int j = j0 + (q * 256 + threadIdx.x) * 4;
    if (j < n) {
      const pi4 *id = reinterpret_cast<const pi4 *>(idx + ((size_t)b * n + j) * 3);
      const pf4 *w = reinterpret_cast<const pf4 *>(weight + ((size_t)b * n + j) * 3);
      ia[q] = id[0]; ib[q] = id[1]; ic[q] = id[2];
      wa[q] = w[0]; wb[q] = w[1]; wc[q] = w[2];
    } else {
      ia[q] = ib[q] = ic[q] = pi4{0, 0, 0, 0};
      wa[q] = wb[q] = wc[q] = pf4{0.f, 0.f, 0.f, 0.f};
    }
  }
  const int l0 = blockIdx.y * cchunk, l1 = min(c, l0 + cchunk);
  const float *p = points + (size_t)b * c * m;
  float *o = out + (size_t)b * c * n;
  for (int l = l0; l < l1; ++l) {
    __syncthreads();
    const float *pl = p + (size_t)l * m;
    if ((m & 3) == 0) {
      for (int i = threadIdx.x * 4; i < m; i += 1024) *reinterpret_cast<pf4 *>(row + i) = *reinterpret_cast<const pf4 *>(pl + i);
    } else {
      for (int i = threadIdx.x; i < m; i += 256) row[i] = pl[i];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int j = j0 + (q * 256 + threadIdx.x) * 4;
      if (j < n) {
        pf4 v;
        v[0] = fmaf(row[ia[q][2]], wa[q][2], fmaf(row[ia[q][1]], wa[q][1], row[ia[q][0]] * wa[q][0]));
        v[1] = fmaf(row[ib[q][1]], wb[q][1], fmaf(row[ib[q][0]], wb[q][0], row[ia[q][3]] * wa[q][3]));
        v[2] = fmaf(row[ic[q][0]], wc[q][0], fmaf(row[ib[q][3]], wb[q][3], row[ib[q][2]] * wb[q][2]));
        v[3] = fmaf(row[ic[q][3]], wc[q][3], fmaf(row[ic[q][2]], wc[q][2], row[ic[q][1]] * wc[q][1]));
        __builtin_nontemporal_store(v, reinterpret_cast<pf4 *>(o + (size_t)l * n + j));
      }
    }
  }
}

__global__ __launch_bounds__(256) void three_interpolate_grad_kernel(int c, int n, int m, int cchunk,
                                                                     const float *__restrict__ grad_out,
                                                                     const int *__restrict__ idx,
                                                                     const float *__restrict__ weight,
                                                                     float *__restrict__ grad_points) {
  const int b = blockIdx.z;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int *id = idx + ((size_t)b * n + j) * 3;
  const float *w = weight + ((size_t)b * n + j) * 3;
  const int i1 = id[0], i2 = id[1], i3 = id[2];
  const float w1 = w[0], w2 = w[1], w3 = w[2];
  const int l0 = blockIdx.y * cchunk, l1 = min(c, l0 + cchunk);
  const float *g = grad_out + (size_t)b * c * n;
  float *gp = grad_points + (size_t)b * c * m;
  for (int l = l0; l < l1; ++l) {
    const float go = g[(size_t)l * n + j];
    atomicAdd(gp + (size_t)l * m + i1, go * w1);
    atomicAdd(gp + (size_t)l * m + i2, go * w2);
    atomicAdd(gp + (size_t)l * m + i3, go * w3);
  }
}

// ---------------------------------------------------------------------------------------------- kNN
// pytorch3d knn_points semantics (see oracle/ops_cpu.c ora_knn_points): K nearest of the search set per query, sorted by
// squared distance, ties -> lower index, empty slots (fewer than K search points) zero.  One thread per query, search
// set LDS-tiled.  The per-query K-best list lives in REGISTERS as KT (= K rounded up to 4 / 8 / 16 / 32 / 64) 64-bit keys
//     key = (float bits of d2) << 32 | index
// d2 >= 0, so key order == (distance, index) lexicographic order; keys are compared as IEEE doubles (a non-negative
// float pattern in the high word is a finite non-negative double with the same ordering), which makes one step of the
// sorted insertion `lo = min(L[p], c); c = max(L[p], c)` -- two full-rate v_min/max_f64, no index shuffling, and the
// tie rule comes for free.  Candidates that beat the current K-th distance are QUEUED per lane in LDS (8 distances are
// evaluated per chunk from 16-byte broadcast tile reads) and the wave inserts in batches when some lane's queue runs
// full: with immediate insertion one lane of the wave would insert while 63 wait.
constexpr int KNN_TILE = 512;
constexpr int KNN_CH = 8;   // candidates per distance chunk
// queue slots per lane (a flush is due once a lane holds more than KNN_Q - KNN_CH): deeper queues amortise the long
// insertion passes of big K better, shallow ones leave LDS for more resident waves
constexpr int knn_queue_slots(int KT) { return KT >= 32 ? 32 : 16; }  // (round 4: 32 = the sort width of the batch merge)

__device__ __forceinline__ double knn_key(float d, int i) {
  return __longlong_as_double(((long long)__float_as_int(d) << 32) | (unsigned int)i);
}
// opaque to the compiler on purpose: as builtins every operand would first be canonicalised (one more f64 op per slot)
__device__ __forceinline__ void knn_minmax(double &l, double &c) {
  double hi;
  asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(l), "v"(c));
  asm("v_min_f64 %0, %0, %1" : "+v"(l) : "v"(c));  // in place: the list stays in its registers across the loop
  c = hi;
}

// Round 4: BATCH MERGE instead of per-candidate insertion for K >= 16.  Inserting one candidate costs a full pass over the KT
// list slots (2 KT f64 ops) in every lane of the wave for as many passes as the FULLEST lane's queue holds -- ~3/4 of the kernel's
// time at K = 32.  A flush now (i) sorts the lane's queued keys with a bitonic network (QS = 16 or 32 slots, unused ones +inf),
// (ii) takes the element-wise minimum of the ascending list and the REVERSED sorted queue -- the half-cleaner of a bitonic merge:
// exactly the KT smallest of the union, as a bitonic sequence -- and (iii) sorts that sequence with log2(KT) merge stages: for
// KT = 32, QS = 32: 480 + 32 + 160 f64 ops per flush whatever the fill, against 64 per queued candidate of the fullest lane
// (up to 1536).  Keys are unique (index in the low word), so the K smallest of the union are the same set in the same order as
// sequential insertion gives: bit-identical results.
template <int N>
__device__ __forceinline__ void knn_bitonic_sort(double (&a)[N]) {
#pragma unroll
  for (int k = 2; k <= N; k <<= 1)
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1)
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const int l = i ^ j;
        if (l > i) {
          if ((i & k) == 0) knn_minmax(a[i], a[l]);  // ascending block: min to the lower slot
          else knn_minmax(a[l], a[i]);               // descending block
        }
      }
}
template <int N>
__device__ __forceinline__ void knn_bitonic_merge(double (&a)[N]) {  // a bitonic (up, then down) -> ascending
#pragma unroll
  for (int j = N >> 1; j > 0; j >>= 1)
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int l = i ^ j;
      if (l > i) knn_minmax(a[i], a[l]);
    }
}

template <int NT, int KT>
__global__ __launch_bounds__(NT) void knn_key_kernel(int nb, int n1, int n2, int K, const float *__restrict__ p1,
                                                     const float *__restrict__ p2,
                                                     const int64_t *__restrict__ lengths2,
                                                     float *__restrict__ dists, int64_t *__restrict__ idx) {
  constexpr int KNN_Q = knn_queue_slots(KT);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float4 *tile = reinterpret_cast<float4 *>(smem);                     // KNN_TILE points (x, y, z, -)
  double *queue = reinterpret_cast<double *>(smem + KNN_TILE * 4);      // [KNN_Q + 1][NT] keys
  const SearchBlock sb = search_block(nb, (n1 + NT - 1) / NT);
  if (!sb.valid) return;
  const int b = sb.b, tid = threadIdx.x;
  const int i = sb.tile * NT + tid;
  p2 += (size_t)b * n2 * 3;
  const int len2 = lengths2 ? (int)lengths2[b] : n2;
  const bool active = i < n1;
  float ax = 0.f, ay = 0.f, az = 0.f;
  if (active) {
    const float *a = p1 + ((size_t)b * n1 + i) * 3;
    ax = a[0]; ay = a[1]; az = a[2];
  }
  const double pad = knn_key(INFINITY, 0x7fffffff);
  double L[KT];
#pragma unroll
  for (int p = 0; p < KT; ++p) L[p] = pad;
  float worst = INFINITY;  // the K-th best distance so far (+inf while fewer than K candidates were seen)
  int qn = 0;
  for (int t0 = 0; t0 < len2; t0 += KNN_TILE) {
    const int tn = min(KNN_TILE, len2 - t0);
    __syncthreads();
    const int tnp = (tn + KNN_CH - 1) / KNN_CH * KNN_CH;  // padded with points at "infinity": never closer than anything
    for (int q = tid; q < tnp; q += NT) {
      const float *c = p2 + (size_t)(t0 + min(q, tn - 1)) * 3;
      tile[q] = q < tn ? make_float4(c[0], c[1], c[2], 0.f) : make_float4(3e38f, 3e38f, 3e38f, 0.f);
    }
    __syncthreads();
    float4 c[KNN_CH];  // the next chunk's points are fetched while the current chunk is queued / inserted
#pragma unroll
    for (int u = 0; u < KNN_CH; ++u) c[u] = tile[u];
    for (int k0 = 0; k0 < tn; k0 += KNN_CH) {
      float d[KNN_CH];
#pragma unroll
      for (int u = 0; u < KNN_CH; ++u) d[u] = sqdist3(ax, ay, az, c[u].x, c[u].y, c[u].z);
      if (k0 + KNN_CH < tn) {
#pragma unroll
        for (int u = 0; u < KNN_CH; ++u) c[u] = tile[k0 + KNN_CH + u];
      }
      // branch-free queueing: every candidate's key is written to the lane's next free slot, the slot is only consumed
      // when the candidate beats the K-th distance (slot KNN_Q is scratch for a full queue)
#pragma unroll
      for (int u = 0; u < KNN_CH; ++u) {
        queue[qn * NT + tid] = knn_key(d[u], t0 + k0 + u);
        qn += (active && d[u] < worst) ? 1 : 0;
      }
      const bool last = k0 + KNN_CH >= tn && t0 + tn >= len2;
      if (__any(qn > KNN_Q - KNN_CH) || last) {
        if constexpr (KT >= 16) {
          constexpr int QS = KNN_Q > 16 ? 32 : 16;   // sort slots (queue depth rounded up to a power of two)
          constexpr int QN = QS < KT ? QS : KT;      // queue entries that can still enter the list
          double Qr[QS];
#pragma unroll
          for (int j = 0; j < QS; ++j) Qr[j] = (j < KNN_Q && j < qn) ? queue[j * NT + tid] : pad;
          knn_bitonic_sort<QS>(Qr);
#pragma unroll
          for (int p = KT - QN; p < KT; ++p) asm("v_min_f64 %0, %0, %1" : "+v"(L[p]) : "v"(Qr[KT - 1 - p]));
          knn_bitonic_merge<KT>(L);
        } else {
          for (int j = 0; __any(j < qn); ++j) {
            double c = j < qn ? queue[j * NT + tid] : pad;
#pragma unroll
            for (int p = 0; p < KT; ++p) knn_minmax(L[p], c);
          }
        }
        qn = 0;
        double w = L[KT - 1];  // K-th entry by a uniform select chain (K is a runtime value <= KT)
#pragma unroll
        for (int p = KT - 2; p >= 0; --p) w = (p == K - 1) ? L[p] : w;
        worst = __int_as_float((int)(__double_as_longlong(w) >> 32));
      }
    }
  }
  // Output through LDS (round 5): a lane's K results are K x 12 bytes at a stride of K elements from the next lane's, so written
  // straight from the registers every store instruction touched 4 / 8 bytes of 64 different cache lines -- the launch moved 5.2x its
  // algorithmic bytes (profiles/r04_ops_roofline.md: 152 vs 29 MB at B 256, n2 1024, K 32), nearly all of it partial-line writes.
  // The workgroup's NT x K block of distances (and of indices) is CONTIGUOUS in the outputs: the keys go to LDS as [K][NT + 1]
  // (the queue region, free by now) and leave as coalesced stores, element e of the block by thread e % NT.
  __syncthreads();
  double *const ol = queue;
#pragma unroll
  for (int p = 0; p < KT; ++p)
    if (p < K) ol[p * (NT + 1) + tid] = L[p];
  __syncthreads();
  const int q0 = sb.tile * NT;
  const int nq = min(NT, n1 - q0);
  const int cnt = len2 < K ? len2 : K;
  float *const od = dists + ((size_t)b * n1 + q0) * K;
  int64_t *const oi = idx + ((size_t)b * n1 + q0) * K;
  for (int e = tid; e < nq * K; e += NT) {
    const int q = e / K, p = e - q * K;
    const long long key = __double_as_longlong(ol[p * (NT + 1) + q]);
    od[e] = p < cnt ? __int_as_float((int)(key >> 32)) : 0.f;
    oi[e] = p < cnt ? (int64_t)(key & 0xffffffffLL) : 0;
  }
}

__global__ __launch_bounds__(256) void knn_gather_kernel(int n2, int u, size_t total,
                                                         const float *__restrict__ x,
                                                         const int64_t *__restrict__ idx,
                                                         float *__restrict__ out, int n1K) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (size_t)gridDim.x * blockDim.x) {
    const size_t slot = e / u;
    const int ch = (int)(e % u);
    const size_t b = slot / n1K;
    out[e] = x[(b * n2 + (size_t)idx[slot]) * u + ch];
  }
}

// gather of whole ROWS: points (b, n, c) row-major, idx (b, m) -> out (b, m, c).  The (B, C, N) layout of gather_points costs a
// 64-byte sector per gathered 4-byte element (with m = n / 4 random indices nearly every sector of a channel row holds a selected
// element: the whole row is fetched, 2.5x the algorithmic bytes at best); a row-major table moves exactly what it gathers.
// One 16-byte piece per thread (c % 4 == 0, 16-byte aligned rows), else one element per thread.
template <bool V4>
__global__ __launch_bounds__(256) void gather_rows_kernel(int n, int m, int c, size_t total, const float *__restrict__ points,
                                                          const int *__restrict__ idx, float *__restrict__ out) {
  const int per = V4 ? c >> 2 : c;  // work items per row
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const size_t slot = e / per;  // (batch, output row)
    const int piece = (int)(e - slot * per);
    const size_t b = slot / m;
    const float *src = points + (b * n + (size_t)idx[slot]) * c;
    if (V4) {
      const pf4 v = __builtin_nontemporal_load(reinterpret_cast<const pf4 *>(src) + piece);
      __builtin_nontemporal_store(v, reinterpret_cast<pf4 *>(out + slot * c) + piece);
    } else {
      out[slot * c + piece] = src[piece];
    }
  }
}

inline int ilog2(int v) {
  int l = 0;
  while ((1 << (l + 1)) <= v) ++l;
  return l;
}
// include/cuda_utils.h:13-19 (floor(log2) computed exactly; the reference's double-log version agrees
// for every w < 65536)
inline int opt_n_threads(int w) {
  int v = 1 << ilog2(w < 1 ? 1 : w);
  return v > 512 ? 512 : (v < 1 ? 1 : v);
}

}  // namespace

extern "C" {

const char *slide_hip_version(void) { return "slide_hip 0.1 (gfx950)"; }

int slide_hip_device_ok(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return 0;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 0;
  const char *a = p.gcnArchName;
  return a[0] == 'g' && a[1] == 'f' && a[2] == 'x' && a[3] == '9' && a[4] == '5' && a[5] == '0';
}

static inline int pick_cchunk(int c, int blocks_other);
static int launch_group_lds(int b, int c, int n, int slots, const float *points, const int *idx, float *out,
                            hipStream_t stream) {
  // index quads per thread: a block's tile of QPT*1024 slots shares one staged n-float row, so long rows want large tiles
  // (the staging traffic is n / tile of the output: 2x the output at n = 8192 with QPT = 4); past 16 quads the index
  // registers cost occupancy, so the longest rows run 512-thread blocks instead (tile = 32768 slots)
  int qpt = slots >= 4096 ? 4 : (slots >= 2048 ? 2 : 1);
  while (qpt < 16 && qpt * 1024 < 4 * n && slots >= qpt * 2048) qpt *= 2;
  const bool wide = qpt == 16 && n >= 8192 && slots >= 65536;
  const int tile = qpt * (wide ? 2048 : 1024);
  const int gx = (slots + tile - 1) / tile;
  int chunks = (1024 + gx * b - 1) / (gx * b);  // >= 4 blocks per CU overall; long channel walks amortise the index reads
  chunks = chunks < 1 ? 1 : (chunks > c ? c : chunks);
  const int cchunk = (c + chunks - 1) / chunks;
  const dim3 grid(gx, (c + cchunk - 1) / cchunk, b);
  const size_t shm = (size_t)((n + 3) & ~3) * 4;
#define GL(Q)                                                                                                        \
  hipLaunchKernelGGL((group_points_lds_kernel<Q>), grid, dim3(256), shm, stream, c, n, slots, cchunk, points, idx, out)
  if (wide)
    hipLaunchKernelGGL((group_points_lds_kernel<16, 512>), grid, dim3(512), shm, stream, c, n, slots, cchunk, points, idx, out);
  else if (qpt == 16) GL(16); else if (qpt == 8) GL(8); else if (qpt == 4) GL(4); else if (qpt == 2) GL(2); else GL(1);
#undef GL
  return LAUNCH_STATUS();
}

int gather_points_kernel_wrapper(int b, int c, int n, int npoints, const float *points, const int *idx,
                                 float *out, slide_stream_t stream) {
  if (b <= 0 || c <= 0 || npoints <= 0) return 0;
  if (npoints % 4 == 0 && (((uintptr_t)idx | (uintptr_t)out | (uintptr_t)points) & 15) == 0 && n <= 16384 && npoints >= 1024) {
    return launch_group_lds(b, c, n, npoints, points, idx, out, (hipStream_t)stream);  // gather = grouping with ns = 1
  }
  if (npoints % 4 == 0 && (((uintptr_t)idx | (uintptr_t)out) & 15) == 0) {
    const int gx = (npoints / 4 + 255) / 256;
    int gy = (2048 + gx * b - 1) / (gx * b);  // enough blocks for the chip, each walking a strided set of channels
    gy = gy < 1 ? 1 : (gy > c ? c : gy);
    hipLaunchKernelGGL(gather_points_v4_kernel, dim3(gx, gy, b), dim3(256), 0, (hipStream_t)stream, c, n, npoints,
                       points, idx, out);
    return LAUNCH_STATUS();
  }
  dim3 grid((npoints + 255) / 256, c < 64 ? c : 64, b);
  hipLaunchKernelGGL(gather_points_kernel, grid, dim3(256), 0, (hipStream_t)stream, c, n, npoints, points,
                     idx, out);
  return LAUNCH_STATUS();
}

int gather_points_grad_kernel_wrapper(int b, int c, int n, int npoints, const float *grad_out,
                                      const int *idx, float *grad_points, slide_stream_t stream) {
  if (b <= 0 || c <= 0 || npoints <= 0) return 0;
  dim3 grid((npoints + 255) / 256, c < 64 ? c : 64, b);
  hipLaunchKernelGGL(gather_points_grad_kernel, grid, dim3(256), 0, (hipStream_t)stream, c, n, npoints,
                     grad_out, idx, grad_points);
  return LAUNCH_STATUS();
}

static int fps_launch(int b, int n, int m, int bs, int skip_origin, const int *start, const float *dataset,
                      float *temp, int *idxs, hipStream_t s) {
  const int lg = ilog2(bs);
  const int L = (n + bs - 1) / bs;
  const int use_lds = (size_t)n * 12 <= 60000;
  const size_t shm = use_lds ? (size_t)n * 12 : 0;
#define FPS_LAUNCH(PPT, NT)                                                                                   \
  hipLaunchKernelGGL((fps_kernel<PPT, NT>), dim3(b), dim3(NT), shm, s, n, m, bs, lg, L, use_lds, skip_origin, \
                     start, dataset, temp, idxs)
  if (n <= 64) FPS_LAUNCH(1, 64);
  else if (n <= 256) FPS_LAUNCH(1, 256);
  else if (n <= 512) FPS_LAUNCH(2, 256);
  else if (n <= 1024) FPS_LAUNCH(4, 256);
  // (round 4, measured and NOT kept -- tools/time_fps.py: 512- / 1024-thread workgroups for long clouds, i.e. fewer points per
  //  thread, are SLOWER: n = 4096 at 256 clouds 1.54 vs 1.13 us per selection, 8.4 vs 3.6 at 2048 clouds -- the cross-wave step
  //  (one barrier + one key per wave) grows with the wave count faster than the per-thread update shrinks, and fewer clouds fit a
  //  CU; only n = 8192 at <= 256 clouds gains, 1.81 vs 2.09)
  else if (n <= 2048) FPS_LAUNCH(8, 256);
  else if (n <= 4096) FPS_LAUNCH(16, 256);
  else if (n <= 8192) FPS_LAUNCH(32, 256);
  else
    hipLaunchKernelGGL(fps_kernel_large, dim3(b), dim3(1024), 0, s, n, m, bs, lg, L, skip_origin, start, dataset, temp,
                       idxs);
#undef FPS_LAUNCH
  return LAUNCH_STATUS();
}

int furthest_point_sampling_kernel_wrapper(int b, int n, int m, const float *dataset, float *temp, int *idxs,
                                           slide_stream_t stream) {
  if (b <= 0 || m <= 0 || n <= 0) return 0;
  return fps_launch(b, n, m, opt_n_threads(n), 1, nullptr, dataset, temp, idxs, (hipStream_t)stream);
}

int slide_sample_farthest_points(int b, int n, int K, const float *points, const int *start_idx, float *temp, int *idx,
                                 slide_stream_t stream) {
  if (b <= 0 || K <= 0 || n <= 0) return 0;
  // plain FPS: no near-origin skip; ties -> lowest index (rank == k with a virtual block size of 1)
  return fps_launch(b, n, K, 1, 0, start_idx, points, temp, idx, (hipStream_t)stream);
}

int query_ball_point_kernel_wrapper(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                                    const float *xyz, int *idx, int *counts, slide_stream_t stream) {
  if (b <= 0 || m <= 0) return 0;
  const float radius2 = radius * radius;
  if (nsample <= 32 && n < 65536)
    hipLaunchKernelGGL(ball_query_kernel<true>, dim3(search_grid(b, (m + 255) / 256)), dim3(256), (size_t)nsample * 512,
                       (hipStream_t)stream, b, n, m, radius2, nsample, new_xyz, xyz, idx, counts);
  else
    hipLaunchKernelGGL(ball_query_kernel<false>, dim3(search_grid(b, (m + 255) / 256)), dim3(256), 0, (hipStream_t)stream, b, n, m,
                       radius2, nsample, new_xyz, xyz, idx, counts);
  return LAUNCH_STATUS();
}

static inline int pick_cchunk(int c, int blocks_other) {
  // enough workgroups to cover 256 CUs several times over, but keep >= 8 channels per thread
  int chunks = (2048 + blocks_other - 1) / blocks_other;
  if (chunks < 1) chunks = 1;
  int cchunk = (c + chunks - 1) / chunks;
  if (cchunk < 8) cchunk = c < 8 ? c : 8;
  return cchunk;
}

int group_points_kernel_wrapper(int b, int c, int n, int npoints, int nsample, const float *points,
                                const int *idx, float *out, slide_stream_t stream) {
  const int slots = npoints * nsample;
  if (b <= 0 || c <= 0 || slots <= 0) return 0;
  if (slots % 4 == 0 && (((uintptr_t)idx | (uintptr_t)out | (uintptr_t)points) & 15) == 0 && n <= 16384 && slots >= 2048) {
    return launch_group_lds(b, c, n, slots, points, idx, out, (hipStream_t)stream);
  }
  if (slots % 4 == 0 && (((uintptr_t)idx | (uintptr_t)out) & 15) == 0) {
    const int gx = (slots / 4 + 255) / 256;
    const int cchunk = pick_cchunk(c, gx * b);
    hipLaunchKernelGGL(group_points_v4_kernel, dim3(gx, (c + cchunk - 1) / cchunk, b), dim3(256), 0,
                       (hipStream_t)stream, c, n, slots, cchunk, points, idx, out);
    return LAUNCH_STATUS();
  }
  const int gx = (slots + 255) / 256;
  const int cchunk = pick_cchunk(c, gx * b);
  hipLaunchKernelGGL(group_points_kernel, dim3(gx, (c + cchunk - 1) / cchunk, b), dim3(256), 0,
                     (hipStream_t)stream, c, n, slots, cchunk, points, idx, out);
  return LAUNCH_STATUS();
}

int group_points_grad_kernel_wrapper(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                                     const int *idx, float *grad_points, slide_stream_t stream) {
  const int slots = npoints * nsample;
  if (b <= 0 || c <= 0 || slots <= 0) return 0;
  const int gx = (slots + 255) / 256;
  const int cchunk = pick_cchunk(c, gx * b);
  hipLaunchKernelGGL(group_points_grad_kernel, dim3(gx, (c + cchunk - 1) / cchunk, b), dim3(256), 0,
                     (hipStream_t)stream, c, n, slots, cchunk, grad_out, idx, grad_points);
  return LAUNCH_STATUS();
}

// three_interpolate for n = 256 / 512 / 1024 outputs per row: the block's 256 threads cover R = 1024 / n channel rows at a
// time (one output quad per thread), each row's m known features staged in LDS -- the scattered 4-byte reads then hit
// LDS instead of the vector cache (12 gathers per 16 bytes stored).  Indices / weights of the thread's quad stay in registers
// over the block's channel chunk.
__global__ __launch_bounds__(256) void three_interpolate_rows_kernel(int c, int m, int n, int cchunk, int mp,
                                                                     const float *__restrict__ points,
                                                                     const int *__restrict__ idx,
                                                                     const float *__restrict__ weight,
                                                                     float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float row[];
  const int b = blockIdx.z;
  const int qn = n >> 2, R = 256 / qn;            // quads per row, rows per pass
  const int r = threadIdx.x / qn, j = (threadIdx.x - r * qn) * 4;
  const pi4 *id = reinterpret_cast<const pi4 *>(idx + ((size_t)b * n + j) * 3);
  const pf4 *w = reinterpret_cast<const pf4 *>(weight + ((size_t)b * n + j) * 3);
  const pi4 ia = id[0], ib = id[1], ic = id[2];
  const pf4 wa = w[0], wb = w[1], wc = w[2];
  const int l0 = blockIdx.y * cchunk, l1 = min(c, l0 + cchunk);
  const float *p = points + (size_t)b * c * m;
  float *o = out + (size_t)b * c * n;
  const float *rw = row + r * mp;
  for (int l = l0; l < l1; l += R) {
    __syncthreads();
    const int nr = min(R, l1 - l);               // rows l .. l + nr - 1 are consecutive in memory: one flat copy
    const float *pl = p + (size_t)l * m;
    if ((m & 3) == 0) {
      for (int i = threadIdx.x * 4; i < nr * m; i += 1024) {
        const int rr = i / m, k = i - rr * m;
        *reinterpret_cast<pf4 *>(row + rr * mp + k) = *reinterpret_cast<const pf4 *>(pl + i);
      }
    } else {
      for (int i = threadIdx.x; i < nr * m; i += 256) row[(i / m) * mp + i % m] = pl[i];
    }
    __syncthreads();
    if (r < nr) {
      pf4 v;
      v[0] = fmaf(rw[ia[2]], wa[2], fmaf(rw[ia[1]], wa[1], rw[ia[0]] * wa[0]));
      v[1] = fmaf(rw[ib[1]], wb[1], fmaf(rw[ib[0]], wb[0], rw[ia[3]] * wa[3]));
      v[2] = fmaf(rw[ic[0]], wc[0], fmaf(rw[ib[3]], wb[3], rw[ib[2]] * wb[2]));
      v[3] = fmaf(rw[ic[3]], wc[3], fmaf(rw[ic[2]], wc[2], rw[ic[1]] * wc[1]));
      __builtin_nontemporal_store(v, reinterpret_cast<pf4 *>(o + (size_t)(l + r) * n + j));
    }
  }
}

int three_nn_kernel_wrapper(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                            int *idx, slide_stream_t stream) {
  if (b <= 0 || n <= 0) return 0;
  hipLaunchKernelGGL(three_nn_kernel, dim3(search_grid(b, (n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, b, n, m,
                     unknown, known, dist2, idx);
  return LAUNCH_STATUS();
}

int three_interpolate_kernel_wrapper(int b, int c, int m, int n, const float *points, const int *idx,
                                     const float *weight, float *out, slide_stream_t stream) {
  if (b <= 0 || c <= 0 || n <= 0) return 0;
  if (n % 4 == 0 && (((uintptr_t)idx | (uintptr_t)weight | (uintptr_t)out | (uintptr_t)points) & 15) == 0 && m <= 16384 &&
      n >= 2048) {
    const int gx = (n + 2047) / 2048;
    int chunks = (1024 + gx * b - 1) / (gx * b);
    chunks = chunks < 1 ? 1 : (chunks > c ? c : chunks);
    const int cchunk = (c + chunks - 1) / chunks;
    hipLaunchKernelGGL(three_interpolate_lds_kernel, dim3(gx, (c + cchunk - 1) / cchunk, b), dim3(256),
                       (size_t)((m + 3) & ~3) * 4, (hipStream_t)stream, c, m, n, cchunk, points, idx, weight, out);
    return LAUNCH_STATUS();
  }
  // (R staged rows of mp floats: the launch stays inside the 64 KB of dynamic LDS every device grants without an attribute --
  //  n = 256 with m > 4092 or n = 512 with m > 8188 fall through to the generic kernels below)
  if ((n == 256 || n == 512 || n == 1024) && (((uintptr_t)idx | (uintptr_t)weight | (uintptr_t)out | (uintptr_t)points) & 15) == 0 &&
      m <= 8192 && (size_t)(1024 / n) * (size_t)(((m + 3) & ~3) + 4) * 4 <= 64 * 1024) {
    const int R = 1024 / n;
    int chunks = (1024 + b - 1) / b;  // >= 4 blocks per CU overall
    chunks = chunks < 1 ? 1 : (chunks > (c + R - 1) / R ? (c + R - 1) / R : chunks);
    int cchunk = (c + chunks - 1) / chunks;
    cchunk = (cchunk + R - 1) / R * R;   // whole passes
    const int mp = ((m + 3) & ~3) + 4;   // row pitch (floats)
    hipLaunchKernelGGL(three_interpolate_rows_kernel, dim3(1, (c + cchunk - 1) / cchunk, b), dim3(256), (size_t)R * mp * 4,
                       (hipStream_t)stream, c, m, n, cchunk, mp, points, idx, weight, out);
    return LAUNCH_STATUS();
  }
  if (n % 4 == 0 && (((uintptr_t)idx | (uintptr_t)weight | (uintptr_t)out) & 15) == 0) {
    const int gx = (n / 4 + 255) / 256;
    const int cchunk = pick_cchunk(c, gx * b);
    hipLaunchKernelGGL(three_interpolate_v4_kernel, dim3(gx, (c + cchunk - 1) / cchunk, b), dim3(256), 0,
                       (hipStream_t)stream, c, m, n, cchunk, points, idx, weight, out);
    return LAUNCH_STATUS();
  }
  const int gx = (n + 255) / 256;
  const int cchunk = pick_cchunk(c, gx * b);
  hipLaunchKernelGGL(three_interpolate_kernel, dim3(gx, (c + cchunk - 1) / cchunk, b), dim3(256), 0,
                     (hipStream_t)stream, c, m, n, cchunk, points, idx, weight, out);
  return LAUNCH_STATUS();
}

int three_interpolate_grad_kernel_wrapper(int b, int c, int n, int m, const float *grad_out, const int *idx,
                                          const float *weight, float *grad_points, slide_stream_t stream) {
  if (b <= 0 || c <= 0 || n <= 0) return 0;
  const int gx = (n + 255) / 256;
  const int cchunk = pick_cchunk(c, gx * b);
  hipLaunchKernelGGL(three_interpolate_grad_kernel, dim3(gx, (c + cchunk - 1) / cchunk, b), dim3(256), 0,
                     (hipStream_t)stream, c, n, m, cchunk, grad_out, idx, weight, grad_points);
  return LAUNCH_STATUS();
}

int slide_knn_points(int b, int n1, int n2, int K, const float *p1, const float *p2, const int64_t *lengths2,
                     float *dists, int64_t *idx, slide_stream_t stream) {
  if (b <= 0 || n1 <= 0 || K <= 0) return 0;
  if (K > 64) return -2;
  hipStream_t s = (hipStream_t)stream;
#define KNN_KEY(NT, KT)                                                                                  \
  hipLaunchKernelGGL((knn_key_kernel<NT, KT>), dim3(search_grid(b, (n1 + NT - 1) / NT)), dim3(NT),         \
                     (size_t)KNN_TILE * 16 + (size_t)((knn_queue_slots(KT) + 1) * NT > KT * (NT + 1) ? (knn_queue_slots(KT) + 1) * NT : KT * (NT + 1)) * 8, \
                     s, b, n1, n2, K, p1, p2, lengths2, dists, idx)
  // few queries per sample: 64-thread workgroups keep more of them on different compute units
  if (n1 > 64) {
    if (K <= 4) KNN_KEY(256, 4); else if (K <= 8) KNN_KEY(256, 8); else if (K <= 16) KNN_KEY(256, 16);
    else if (K <= 32) KNN_KEY(256, 32); else KNN_KEY(128, 64);
  } else {
    if (K <= 4) KNN_KEY(64, 4); else if (K <= 8) KNN_KEY(64, 8); else if (K <= 16) KNN_KEY(64, 16);
    else if (K <= 32) KNN_KEY(64, 32); else KNN_KEY(64, 64);
  }
#undef KNN_KEY
  return LAUNCH_STATUS();
}

int slide_knn_gather(int b, int n2, int u, int n1, int K, const float *x, const int64_t *idx, float *out,
                     slide_stream_t stream) {
  const size_t total = (size_t)b * n1 * K * u;
  if (total == 0) return 0;
  size_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(knn_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n2, u, total,
                     x, idx, out, n1 * K);
  return LAUNCH_STATUS();
}

int slide_gather_rows(int b, int n, int m, int c, const float *points, const int *idx, float *out, slide_stream_t stream) {
  if (b <= 0 || m <= 0 || c <= 0) return 0;
  const bool v4 = c % 4 == 0 && (((uintptr_t)points | (uintptr_t)out) & 15) == 0;
  const size_t total = (size_t)b * m * (v4 ? c / 4 : c);
  size_t blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  if (v4)
    hipLaunchKernelGGL(gather_rows_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n, m, c, total, points,
                       idx, out);
  else
    hipLaunchKernelGGL(gather_rows_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n, m, c, total, points,
                       idx, out);
  return LAUNCH_STATUS();
}

}  // extern "C"
