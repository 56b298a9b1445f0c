// rows_ops.hip -- gfx950 kernels of the ROW-MAJOR module-level path (pointnet2_ops.pointnet2_modules / attention).
//
// The reference keeps activations as (B, C, npoint, K) tensors and runs nn.Conv2d(1x1) / GroupNorm / cat / softmax on
// them (pointnet2_ops_lib/pointnet2_ops/pointnet2_modules.py:71-176, attention.py:35-96).  Here a grouped activation is
// ONE matrix [B * npoint * K][ld] (ld = channels rounded up to 32, pad columns zero) that is produced row-major by the
// grouping kernel, consumed row-major by the MFMA GEMMs, normalised in place and reduced over K by the attention
// kernel: no layout change between layers, no concatenation copies of K-expanded tensors.  HBM-bound byte movers: one
// thread owns 4 (GroupNorm) or 8 (grouping) consecutive channels of a row, a wave reads whole rows.
//
//   ROWS_FROM_NCX / ROWS_TO_NCX  module boundary: (B, C, P) fp32 <-> [B*P][ld]
//   ROWS_GROUP    QueryAndGroup 'nn' feature assembly (pointnet2_utils.py:383-430) and group_knn (:497-524)
//   ROWS_GN       [ReLU] GroupNorm over (group x rows of a sample) [ReLU] [+ per-sample vector] [+ residual rows]
//   ROWS_CONCAT_QK  relu([query(point) | key(point, neighbour)])    (attention.py:86-88)
//   ROWS_ATTN     softmax over the K neighbours + weighted sum      (attention.py:89-95)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/slide_engine.h"

namespace {

typedef _Float16 half_t;

template <typename T, int N>
struct alignas(sizeof(T) * N) Pack { T v[N]; };

// ------------------------------------------------------------------------------------------ layout change at the boundary
template <typename T>
__global__ __launch_bounds__(256) void rows_from_ncx_kernel(int C, int P, int ld, const float *__restrict__ in,
                                                            T *__restrict__ out) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, c0 = blockIdx.x * 32, p0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  in += (size_t)b * C * P;
  out += (size_t)b * P * ld;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, p = p0 + tx;
    tile[ty + 8 * i][tx] = (c < C && p < P) ? in[(size_t)c * P + p] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = p0 + ty + 8 * i, c = c0 + tx;
    if (p < P && c < ld) out[(size_t)p * ld + c] = (T)tile[tx][ty + 8 * i];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void rows_to_ncx_kernel(int C, int P, int ld, const T *__restrict__ in,
                                                          float *__restrict__ out) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, c0 = blockIdx.x * 32, p0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  in += (size_t)b * P * ld;
  out += (size_t)b * C * P;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = p0 + ty + 8 * i, c = c0 + tx;
    if (p < P && c < C) tile[ty + 8 * i][tx] = (float)in[(size_t)p * ld + c];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, p = p0 + tx;
    if (p < P && c < C) out[(size_t)c * P + p] = tile[tx][ty + 8 * i];
  }
}

// ------------------------------------------------------------------------------------------ grouping
// row (b, p, k) of out = [feat[b][idx[b][p][k]][0..C) | coordinate channels | 0 ...]; coordinate channels:
//   SA  (flags & 1) == 0:  rel(3) = nbr - centre, then abs(3) if flags & 2, then centre(3) if flags & 4; none if flags & 8
//   FP  (flags & 1):       d2, w = (1/(d2+1e-8)) / sum_k(1/(d2+1e-8)), abs(3), rel(3), centre(3)
// feat == NULL: C = 0.  flags & 16: idx is int32 (ball_query) instead of int64 (knn_points).  counts != NULL: a centre whose
// ball is empty (counts[b][p] == 0) gets itself as its only neighbour, with zero features (QueryAndGroup subset=False,
// pointnet2_utils.py:396-421).  One thread per 8 output columns.
template <typename T>
__global__ __launch_bounds__(256) void rows_group_kernel(int N, int np, int K, int C, int ldf, int ldg, int flags,
                                                         const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                         const T *__restrict__ feat, const void *__restrict__ idx,
                                                         const float *__restrict__ d2, const int *__restrict__ counts,
                                                         T *__restrict__ out, size_t total) {
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int ppr = ldg >> 3;
  const size_t row = gid / ppr;
  const int c0 = (int)(gid - row * ppr) * 8;
  const size_t pt = row / K;  // b * np + p
  const int b = (int)(pt / np);
  const int nb = (flags & 16) ? static_cast<const int *>(idx)[row] : (int)static_cast<const int64_t *>(idx)[row];
  const bool empty = counts && counts[pt] == 0;
  T *o = out + row * ldg + c0;
  const T *f = feat + ((size_t)b * N + nb) * ldf;
  if (c0 + 8 <= C && !empty) {
    *reinterpret_cast<Pack<T, 8> *>(o) = *reinterpret_cast<const Pack<T, 8> *>(f + c0);
    return;
  }
  Pack<T, 8> r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r.v[j] = (T)0.f;
  if (c0 < C + 11) {  // this piece holds coordinate channels
    const float *ctr = new_xyz + pt * 3;
    const float *q = empty ? ctr : xyz + ((size_t)b * N + nb) * 3;
    float cv[11];
    int ncv;
    if (flags & 1) {
      const float *dd = d2 + pt * K;
      float s = 0.f;
      for (int k = 0; k < K; ++k) s += 1.0f / (dd[k] + 1e-8f);
      const float dk = d2[row];
      cv[0] = dk;
      cv[1] = (1.0f / (dk + 1e-8f)) / s;
#pragma unroll
      for (int j = 0; j < 3; ++j) { cv[2 + j] = q[j]; cv[5 + j] = q[j] - ctr[j]; cv[8 + j] = ctr[j]; }
      ncv = 11;
    } else if (flags & 8) {
      ncv = 0;
    } else {
      ncv = 3;
#pragma unroll
      for (int j = 0; j < 3; ++j) cv[j] = q[j] - ctr[j];
      if (flags & 2) {
#pragma unroll
        for (int j = 0; j < 3; ++j) cv[ncv + j] = q[j];
        ncv += 3;
      }
      if (flags & 4) {
#pragma unroll
        for (int j = 0; j < 3; ++j) cv[ncv + j] = ctr[j];
        ncv += 3;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = c0 + j;
      if (c < C) r.v[j] = empty ? (T)0.f : f[c];
      else if (c - C < ncv) {
        float v = 0.f;
#pragma unroll
        for (int u = 0; u < 11; ++u) v = (c - C == u) ? cv[u] : v;
        r.v[j] = (T)v;
      }
    }
  }
  *reinterpret_cast<Pack<T, 8> *>(o) = r;
}

// ------------------------------------------------------------------------------------------ GroupNorm on rows
// Thread layout shared by both passes: a thread owns VEC = 16 bytes of consecutive channels (4 fp32 / 8 fp16), cn = ld / VEC
// threads cover one row, rt = 256 / cn rows are in flight per workgroup step (x2 unrolled for memory-level parallelism).
// Pass 1: per (sample, row chunk) channel sums of x (or relu(x)) -> part[b][chunk][ld][2]; deterministic (no atomics).
template <typename T>
__global__ __launch_bounds__(256) void rows_gn_stats_kernel(int S, int ld, int rpc, int pre_relu, const T *__restrict__ x,
                                                            float *__restrict__ part) {
  constexpr int VEC = 16 / sizeof(T);
  __shared__ float red[256 * VEC * 2];
  const int cn = ld / VEC, rt = 256 / cn;
  const int pr = threadIdx.x / cn, pc = threadIdx.x - pr * cn;
  const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
  const int r0 = chunk * rpc, r1 = min(S, r0 + rpc);
  float s[VEC], q[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) s[j] = q[j] = 0.f;
  if (pr < rt) {
    const T *xp = x + ((size_t)b * S) * ld + pc * VEC;
    int r = r0 + pr;
    for (; r + rt < r1; r += 2 * rt) {
      const Pack<T, VEC> v0 = *reinterpret_cast<const Pack<T, VEC> *>(xp + (size_t)r * ld);
      const Pack<T, VEC> v1 = *reinterpret_cast<const Pack<T, VEC> *>(xp + (size_t)(r + rt) * ld);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float f0 = (float)v0.v[j], f1 = (float)v1.v[j];
        if (pre_relu) { f0 = fmaxf(f0, 0.f); f1 = fmaxf(f1, 0.f); }
        s[j] += f0; q[j] += f0 * f0;
        s[j] += f1; q[j] += f1 * f1;
      }
    }
    if (r < r1) {
      const Pack<T, VEC> v0 = *reinterpret_cast<const Pack<T, VEC> *>(xp + (size_t)r * ld);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float f0 = (float)v0.v[j];
        if (pre_relu) f0 = fmaxf(f0, 0.f);
        s[j] += f0; q[j] += f0 * f0;
      }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      red[(pr * ld + pc * VEC + j) * 2 + 0] = s[j];
      red[(pr * ld + pc * VEC + j) * 2 + 1] = q[j];
    }
  }
  __syncthreads();
  float *po = part + (((size_t)b * nchunk + chunk) * ld) * 2;
  for (int c = threadIdx.x; c < ld; c += 256) {
    float ss = 0.f, qq = 0.f;
    for (int r = 0; r < rt; ++r) {
      ss += red[(r * ld + c) * 2 + 0];
      qq += red[(r * ld + c) * 2 + 1];
    }
    po[c * 2 + 0] = ss;
    po[c * 2 + 1] = qq;
  }
}

// Between the passes, once per sample: partial sums (of pass 1, or the per-tile sums a GEMM's STATS epilogue published) ->
// scale = gamma * rstd and shift = beta - mean * scale per channel, ss[b][0 | 1][ld].  (Every workgroup of pass 2 used to
// redo this reduction: up to 64 x ld x 2 floats read per workgroup -- as many bytes as its share of the tensor.)
__global__ __launch_bounds__(256) void rows_gn_finalize_kernel(int S, int ld, int nchunk_stats, int G, int n_norm,
                                                               const float *__restrict__ part,
                                                               const float *__restrict__ gamma,
                                                               const float *__restrict__ beta,
                                                               const float *__restrict__ tsum, const float *__restrict__ tsq,
                                                               int tps, float *__restrict__ ss, float *__restrict__ mr) {
  __shared__ float lsum[1024], lsq[1024], lmean[64], lrstd[64];
  const int b = blockIdx.x;
  const float *pp = part + ((size_t)b * nchunk_stats * ld) * 2;
  for (int c = threadIdx.x; c < n_norm; c += 256) {
    float s1 = 0.f, q1 = 0.f;
    if (tsum) {  // per-256-row-tile sums published by the producing GEMM's epilogue (STATS mode): tps tiles per sample
      for (int k = 0; k < tps; ++k) {
        s1 += tsum[((size_t)b * tps + k) * ld + c];
        q1 += tsq[((size_t)b * tps + k) * ld + c];
      }
    } else {
      for (int k = 0; k < nchunk_stats; ++k) {
        s1 += pp[((size_t)k * ld + c) * 2 + 0];
        q1 += pp[((size_t)k * ld + c) * 2 + 1];
      }
    }
    lsum[c] = s1;
    lsq[c] = q1;
  }
  __syncthreads();
  const int gs = n_norm / G;
  const float inv = 1.0f / ((float)gs * (float)S);
  for (int g = threadIdx.x; g < G; g += 256) {
    float s1 = 0.f, q1 = 0.f;
    for (int j = 0; j < gs; ++j) {
      s1 += lsum[g * gs + j];
      q1 += lsq[g * gs + j];
    }
    const float mean = s1 * inv;
    const float var = fmaxf(q1 * inv - mean * mean, 0.f);
    lmean[g] = mean;
    lrstd[g] = 1.0f / sqrtf(var + 1e-5f);
    if (mr) {  // the training step's backward reads the statistics back (slide_gn_rows_bwd): [b][64][mean | rstd]
      mr[((size_t)b * 64 + g) * 2 + 0] = mean;
      mr[((size_t)b * 64 + g) * 2 + 1] = lrstd[g];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < ld; c += 256) {
    float sc = 1.f, sh = 0.f;
    if (c < n_norm) {
      const int g = c / gs;
      sc = gamma[c] * lrstd[g];
      sh = beta[c] - lmean[g] * sc;
    }
    ss[((size_t)b * 2 + 0) * ld + c] = sc;
    ss[((size_t)b * 2 + 1) * ld + c] = sh;
  }
}

// Joint GroupNorm of the attention's VIRTUAL concatenation [q(point) broadcast over the K neighbours | k(point, neighbour)]
// (weight_conv.1 of AttentionModule, attention.py:45-47) from the per-tile channel sums the two producing GEMMs published: the
// concatenated tensor is never built (round 5; the concat + ReLU pass and the statistics pass over it were 16 % of the decode leg's
// bytes).  Per sample: channel sums of the q half = mult_q x its tile sums (every point row stands K times in the concatenation),
// of the k half = its tile sums; groups may straddle the two halves; scale / shift go to TWO tables in the producers' own layouts
// (ssq [B][2][ldq], ssk [B][2][ldk]) for the consumers' deferred-normalisation loaders.
__global__ __launch_bounds__(256) void rows_gn_joint_kernel(int C1, int ldq, int tq, float mult_q, int C2, int ldk, int tk, int G,
                                                            int n_norm, float inv_rows, const float *__restrict__ qsum,
                                                            const float *__restrict__ qsq, const float *__restrict__ ksum,
                                                            const float *__restrict__ ksq, const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, float *__restrict__ ssq,
                                                            float *__restrict__ ssk) {
  __shared__ float lsum[1024], lsq[1024], lmean[64], lrstd[64];
  const int b = blockIdx.x, C = C1 + C2;
  for (int c = threadIdx.x; c < C; c += 256) {
    float s1 = 0.f, q1 = 0.f;
    if (c < C1) {
      for (int t = 0; t < tq; ++t) { s1 += qsum[((size_t)b * tq + t) * ldq + c]; q1 += qsq[((size_t)b * tq + t) * ldq + c]; }
      s1 *= mult_q; q1 *= mult_q;
    } else {
      for (int t = 0; t < tk; ++t) { s1 += ksum[((size_t)b * tk + t) * ldk + (c - C1)]; q1 += ksq[((size_t)b * tk + t) * ldk + (c - C1)]; }
    }
    lsum[c] = s1; lsq[c] = q1;
  }
  __syncthreads();
  const int gs = n_norm / G;
  const float inv = inv_rows / (float)gs;
  for (int g = threadIdx.x; g < G; g += 256) {
    float s1 = 0.f, q1 = 0.f;
    for (int j = 0; j < gs; ++j) { s1 += lsum[g * gs + j]; q1 += lsq[g * gs + j]; }
    const float mean = s1 * inv;
    const float var = fmaxf(q1 * inv - mean * mean, 0.f);
    lmean[g] = mean;
    lrstd[g] = 1.0f / sqrtf(var + 1e-5f);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < ldq + ldk; c += 256) {
    const bool isq = c < ldq;
    const int cl = isq ? c : c - ldq;              // channel inside its producer's row
    const bool valid = isq ? cl < C1 : cl < C2;    // (pad columns: identity -- their rows are zero and so are their weights)
    const int cj = isq ? cl : C1 + cl;             // logical channel of the concatenation
    float sc = 1.f, sh = 0.f;
    if (valid && cj < n_norm) {
      const int g = cj / gs;
      sc = gamma[cj] * lrstd[g];
      sh = beta[cj] - lmean[g] * sc;
    }
    float *dst = isq ? ssq + (size_t)b * 2 * ldq + cl : ssk + (size_t)b * 2 * ldk + cl;
    dst[0] = sc;
    dst[isq ? ldq : ldk] = sh;
  }
}

// Pass 2: y = post_relu( pre_relu(x) * scale + shift ) + addvec[b] + residual[row]; ss == NULL: no normalisation at all --
// the plain ReLU / add epilogue of a layer without GroupNorm.
template <typename T>
__global__ __launch_bounds__(256) void rows_gn_apply_kernel(int S, int ld, int rpc, int flags, const T *__restrict__ x,
                                                            const float *__restrict__ ss,
                                                            const float *__restrict__ addvec, int addvec_ld,
                                                            const T *__restrict__ res, int res_ld, T *__restrict__ y) {
  constexpr int VEC = 16 / sizeof(T);
  const int cn = ld / VEC, rt = 256 / cn;
  const int pr = threadIdx.x / cn, pc = threadIdx.x - pr * cn;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const bool pre_relu = flags & 1, post_relu = flags & 2;
  if (pr >= rt) return;
  float sc[VEC], sh[VEC], av[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const int c = pc * VEC + j;
    sc[j] = ss ? ss[((size_t)b * 2 + 0) * ld + c] : 1.f;
    sh[j] = ss ? ss[((size_t)b * 2 + 1) * ld + c] : 0.f;
    av[j] = (addvec && c < addvec_ld) ? addvec[(size_t)b * addvec_ld + c] : 0.f;
  }
  const int r0 = chunk * rpc, r1 = min(S, r0 + rpc);
  const size_t base = (size_t)b * S;
  auto one = [&](Pack<T, VEC> v, const Pack<T, VEC> &rv) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float f = (float)v.v[j];
      if (pre_relu) f = fmaxf(f, 0.f);
      f = f * sc[j] + sh[j];
      if (post_relu) f = fmaxf(f, 0.f);
      f += av[j];
      if (res) f += (float)rv.v[j];
      v.v[j] = (T)f;
    }
    return v;
  };
  int r = r0 + pr;
  for (; r + rt < r1; r += 2 * rt) {
    const size_t o0 = (base + r) * ld + pc * VEC, o1 = (base + r + rt) * ld + pc * VEC;
    const Pack<T, VEC> v0 = *reinterpret_cast<const Pack<T, VEC> *>(x + o0);
    const Pack<T, VEC> v1 = *reinterpret_cast<const Pack<T, VEC> *>(x + o1);
    Pack<T, VEC> a0 = v0, a1 = v1;
    if (res) {
      a0 = *reinterpret_cast<const Pack<T, VEC> *>(res + (base + r) * res_ld + pc * VEC);
      a1 = *reinterpret_cast<const Pack<T, VEC> *>(res + (base + r + rt) * res_ld + pc * VEC);
    }
    *reinterpret_cast<Pack<T, VEC> *>(y + o0) = one(v0, a0);
    *reinterpret_cast<Pack<T, VEC> *>(y + o1) = one(v1, a1);
  }
  if (r < r1) {
    const size_t o0 = (base + r) * ld + pc * VEC;
    const Pack<T, VEC> v0 = *reinterpret_cast<const Pack<T, VEC> *>(x + o0);
    Pack<T, VEC> a0 = v0;
    if (res) a0 = *reinterpret_cast<const Pack<T, VEC> *>(res + (base + r) * res_ld + pc * VEC);
    *reinterpret_cast<Pack<T, VEC> *>(y + o0) = one(v0, a0);
  }
}

// ------------------------------------------------------------------------------------------ attention glue
// out[row][0..C1) = relu(q[row / K]), out[row][C1..C1+C2) = relu(k[row]), rest 0.  One thread per 16 bytes of an output
// row; pieces that lie inside one source and are 16-byte aligned there move as one vector load (requests, not bytes, bound
// these kernels: tools/lds_fill.hip), the pieces at the seam element-wise.
template <typename T>
__global__ __launch_bounds__(256) void rows_concat_qk_kernel(int K, int C1, int ldq, int C2, int ldk, int ldo,
                                                             const T *__restrict__ q, const T *__restrict__ k,
                                                             T *__restrict__ out, size_t total) {
  constexpr int VEC = 16 / sizeof(T);
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int ppr = ldo / VEC;
  const size_t row = gid / ppr;
  const int c0 = (int)(gid - row * ppr) * VEC;
  const T *qr = q + (row / K) * ldq;
  const T *kr = k + row * ldk;
  Pack<T, VEC> r;
  if (c0 + VEC <= C1) {
    r = *reinterpret_cast<const Pack<T, VEC> *>(qr + c0);
  } else if (c0 >= C1 && c0 + VEC <= C1 + C2 && (C1 % VEC) == 0) {
    r = *reinterpret_cast<const Pack<T, VEC> *>(kr + (c0 - C1));
  } else {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int c = c0 + j;
      r.v[j] = c < C1 ? qr[c] : c < C1 + C2 ? kr[c - C1] : (T)0.f;
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) r.v[j] = (T)fmaxf((float)r.v[j], 0.f);
  *reinterpret_cast<Pack<T, VEC> *>(out + row * ldo + c0) = r;
}

// out[pt][c] = sum_k softmax_k(S[pt*K + k][c]) * V[pt*K + k][c]; one thread per (point, 16 bytes of channels): 16-byte
// loads (the kernel is bound by memory requests, not bytes), three passes over the point's K score rows (cache hits).
// counts != NULL: only the first max(1, counts[pt]) neighbour slots take part (the reference masks the others with -1e9
// before the softmax, attention.py:89-93: their weights are exactly 0 in fp32)
template <typename T>
__global__ __launch_bounds__(256) void rows_attn_kernel(int K, int C, int lds_, int ldv, int ldo, const T *__restrict__ Sx,
                                                        const T *__restrict__ V, const int *__restrict__ counts,
                                                        T *__restrict__ out, size_t total, const float *__restrict__ vss,
                                                        int pps, int v_relu) {
  constexpr int VEC = 16 / sizeof(T);
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int ppr = ldo / VEC;
  const size_t pt = gid / ppr;
  const int c0 = (int)(gid - pt * ppr) * VEC;
  const int kk = counts ? max(1, min(K, counts[pt])) : K;
  // deferred normalisation of the values (vss [sample][scale | shift][ldv], pps points per sample): v' = relu?(v * scale + shift)
  float vs[VEC], vh[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    vs[j] = vss ? vss[((size_t)(pt / pps) * 2 + 0) * ldv + c0 + j] : 1.f;
    vh[j] = vss ? vss[((size_t)(pt / pps) * 2 + 1) * ldv + c0 + j] : 0.f;
  }
  const T *sp = Sx + pt * K * lds_ + c0;
  const T *vp = V + pt * K * ldv + c0;
  float m[VEC], l[VEC], acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { m[j] = -INFINITY; l[j] = 0.f; acc[j] = 0.f; }
  for (int k = 0; k < kk; ++k) {
    const Pack<T, VEC> sv = *reinterpret_cast<const Pack<T, VEC> *>(sp + (size_t)k * lds_);
#pragma unroll
    for (int j = 0; j < VEC; ++j) m[j] = fmaxf(m[j], (float)sv.v[j]);
  }
  for (int k = 0; k < kk; ++k) {
    const Pack<T, VEC> sv = *reinterpret_cast<const Pack<T, VEC> *>(sp + (size_t)k * lds_);
#pragma unroll
    for (int j = 0; j < VEC; ++j) l[j] += expf((float)sv.v[j] - m[j]);
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) l[j] = 1.0f / l[j];
  for (int k = 0; k < kk; ++k) {
    const Pack<T, VEC> sv = *reinterpret_cast<const Pack<T, VEC> *>(sp + (size_t)k * lds_);
    const Pack<T, VEC> vv = *reinterpret_cast<const Pack<T, VEC> *>(vp + (size_t)k * ldv);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float v = (float)vv.v[j];
      if (vss) {
        v = v * vs[j] + vh[j];
        if (v_relu) v = fmaxf(v, 0.f);
      }
      acc[j] += v * (expf((float)sv.v[j] - m[j]) * l[j]);
    }
  }
  Pack<T, VEC> r;
#pragma unroll
  for (int j = 0; j < VEC; ++j) r.v[j] = (c0 + j < C) ? (T)acc[j] : (T)0.f;
  *reinterpret_cast<Pack<T, VEC> *>(out + pt * ldo + c0) = r;
}

// pooling over the K neighbour rows of a point (pooling_features, pointnet2_modules.py:179-211): mode 0 max over all K
// slots, 1 mean over the first max(1, counts[pt]) slots (all K without counts), 2 = max for the first C / 2 channels and
// mean for the rest.  One thread per (point, 16 bytes of channels).
template <typename T>
__global__ __launch_bounds__(256) void rows_pool_kernel(int K, int C, int ldx, int ldo, int mode, const T *__restrict__ x,
                                                        const int *__restrict__ counts, T *__restrict__ out, size_t total) {
  constexpr int VEC = 16 / sizeof(T);
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int ppr = ldo / VEC;
  const size_t pt = gid / ppr;
  const int c0 = (int)(gid - pt * ppr) * VEC;
  const T *xp = x + pt * K * ldx + c0;
  const int n = counts ? max(1, min(K, counts[pt])) : K;
  float mx[VEC], sm[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { mx[j] = -INFINITY; sm[j] = 0.f; }
  for (int k = 0; k < K; ++k) {
    const Pack<T, VEC> v = *reinterpret_cast<const Pack<T, VEC> *>(xp + (size_t)k * ldx);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      mx[j] = fmaxf(mx[j], (float)v.v[j]);
      if (k < n) sm[j] += (float)v.v[j];
    }
  }
  Pack<T, VEC> r;
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const int c = c0 + j;
    const bool use_max = mode == 0 || (mode == 2 && c < C / 2);
    r.v[j] = c < C ? (T)(use_max ? mx[j] : sm[j] / (float)n) : (T)0.f;
  }
  *reinterpret_cast<Pack<T, VEC> *>(out + pt * ldo + c0) = r;
}

// ------------------------------------------------------------------------------------------ pair expansion (round 6)
// A 1 x 1 convolution over a GROUPED input is linear in [features of the neighbour | coordinate channels], and every coordinate
// channel is linear in xyz[neighbour] and xyz[centre] (SA: rel | abs | centre; FP: d2 | w | abs | rel | centre: the layouts of
// rows_group_kernel above), so its output for row (b, p, k) separates:
//     y(b, p, k) = A[b][idx(b, p, k)]  +  bias  +  Cc . xyz[centre]  (+ wd d2 + ww w)
// with A = W_features . features + Cq . xyz one row per SOURCE point (a GEMM on B * N rows instead of B * npoint * K: 1 / 64 of the
// MACs at 1024 centres x 16 neighbours of 256 points; the coordinate part added in fp32 by the host program), Cq = W_rel + W_abs,
// Cc = W_centre - W_rel.  This kernel builds y from the fp32 table A
// and the coordinates -- the grouped matrix (the widest tensor of a level) is never written or read, and the coordinate terms are
// evaluated in fp32 from fp32 coordinates (the grouped matrix held them rounded to the activation type).  The pair decomposition of
// the fused DDPM plan (gemm_gx.hip), for the module-level path's arbitrary N / npoint / K.
// One workgroup per 256-row tile: thread r first derives row r's neighbour row, coordinates and slot scalars into LDS, then a thread
// owns FOUR channels and walks its share of the rows (a wave reads 1 KB of one A row and writes 512 B of one output row per step);
// ReLU (flags & 1) and the per-tile channel sums of the GroupNorm that follows (the GEMM epilogue's STATS mode) ride along.
template <typename T, int CH>  // CH = channels per thread: 8 (16-byte stores: wide rows) or 4 (more threads per row: narrow rows)
__global__ __launch_bounds__(256) void rows_pair_expand_kernel(int N, int np, int K, int ld, int ldA, int flags, int tps, size_t rows,
                                                               const float *__restrict__ A, const float *__restrict__ bias,
                                                               const float *__restrict__ coef, const float *__restrict__ xyz,
                                                               const float *__restrict__ new_xyz, const void *__restrict__ idx,
                                                               const float *__restrict__ d2, T *__restrict__ out,
                                                               float *__restrict__ st_sum, float *__restrict__ st_sq) {
  __shared__ __attribute__((aligned(16))) float rs[256][8];  // qx qy qz cx cy cz d2 w
  __shared__ int rsrc[256];
  __shared__ float red[2][2048];
  const int tid = threadIdx.x;
  // XCD-aware tile map (samples of whole tiles): every tile of a sample runs on ONE XCD (workgroup id mod 8; for speed only), so the
  // sample's table -- N x ld floats, gathered 16 x K / N times over -- stays in that XCD's L2 instead of being fetched by all eight
  size_t tile = blockIdx.x;
  if (tps > 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int smp = (j / tps) * 8 + xcd;
    if ((size_t)smp * tps * 256 >= rows) return;
    tile = (size_t)smp * tps + j % tps;
  }
  const size_t row0 = tile * 256;
  {
    const size_t row = row0 + tid;
    int src = -1;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (row < rows) {
      const size_t pt = row / K;
      const int b = (int)(pt / np);
      const int nb = (flags & 16) ? static_cast<const int *>(idx)[row] : (int)static_cast<const int64_t *>(idx)[row];
      src = b * N + nb;
      const float *q = xyz + (size_t)src * 3, *c = new_xyz + pt * 3;
      v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = c[0]; v[4] = c[1]; v[5] = c[2];
      if (flags & 2) {  // group_knn's slot scalars (pointnet2_utils.py:510-513)
        const float *dd = d2 + pt * K;
        float sum = 0.f;
        for (int k = 0; k < K; ++k) sum += 1.0f / (dd[k] + 1e-8f);
        v[6] = d2[row];
        v[7] = (1.0f / (v[6] + 1e-8f)) / sum;
      }
    }
    rsrc[tid] = src;
    *reinterpret_cast<float4 *>(&rs[tid][0]) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4 *>(&rs[tid][4]) = make_float4(v[4], v[5], v[6], v[7]);
  }
  __syncthreads();
  const int pieces = ld / CH;                       // pieces per row (ld <= 1024, a multiple of 32)
  const int nsub = 256 / pieces > 0 ? 256 / pieces : 1;
  const bool relu = (flags & 1) != 0, fp = (flags & 2) != 0, want_stats = st_sum != nullptr;
  const int pc = tid % pieces, sub = tid / pieces;
  const bool active = sub < nsub;
  float s[CH], ss[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) s[j] = ss[j] = 0.f;
  if (active) {
    const int c0 = pc * CH;
    // coef[c] = (unused: the neighbour's coordinate term is part of A | Cc = W_centre - W_rel | w_d2 | w_w)
    float cc[CH][3], wd[CH], ww[CH], bi[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const float4 u1 = *reinterpret_cast<const float4 *>(coef + (size_t)(c0 + j) * 8 + 4);
      cc[j][0] = coef[(size_t)(c0 + j) * 8 + 3]; cc[j][1] = u1.x; cc[j][2] = u1.y; wd[j] = u1.z; ww[j] = u1.w;
      bi[j] = bias[c0 + j];
    }
    // a thread's rows are CONSECUTIVE (chunk of the tile): the K neighbour rows of a point share the centre term
    const int chunk = (256 + nsub - 1) / nsub, r_end = (sub + 1) * chunk < 256 ? (sub + 1) * chunk : 256;
    float ct[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) ct[j] = 0.f;
    float pcx = __builtin_nanf(""), pcy = 0.f, pcz = 0.f;
    // four rows per trip: their table rows are requested together (a trip is one L2 round trip, not four)
    constexpr int UB = 4;
    for (int r8 = sub * chunk; r8 < r_end; r8 += UB) {
      int src[UB];
      float4 a4[UB][CH == 8 ? 2 : 1];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        src[u] = r8 + u < r_end ? rsrc[r8 + u] : -1;
        const float *ap = A + (size_t)(src[u] < 0 ? 0 : src[u]) * ldA + c0;
        a4[u][0] = *reinterpret_cast<const float4 *>(ap);
        if (CH == 8) a4[u][1] = *reinterpret_cast<const float4 *>(ap + 4);
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        if (src[u] < 0) continue;  // (rows beyond the chunk / the end of the matrix)
        const int rr = r8 + u;
        const float4 r1 = *reinterpret_cast<const float4 *>(&rs[rr][4]);
        const float cx = rs[rr][3];
        if (!(cx == pcx && r1.x == pcy && r1.y == pcz)) {  // a new centre
          pcx = cx; pcy = r1.x; pcz = r1.y;
#pragma unroll
          for (int j = 0; j < CH; ++j) ct[j] = fmaf(cc[j][2], pcz, fmaf(cc[j][1], pcy, fmaf(cc[j][0], pcx, bi[j])));
        }
        float av[8] = {a4[u][0].x, a4[u][0].y, a4[u][0].z, a4[u][0].w, 0.f, 0.f, 0.f, 0.f};
        if (CH == 8) { av[4] = a4[u][1].x; av[5] = a4[u][1].y; av[6] = a4[u][1].z; av[7] = a4[u][1].w; }
        Pack<T, CH> o;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          float t = av[j] + ct[j];
          if (fp) t = fmaf(ww[j], r1.w, fmaf(wd[j], r1.z, t));
          if (relu) t = fmaxf(t, 0.f);
          s[j] += t; ss[j] = fmaf(t, t, ss[j]);
          o.v[j] = (T)t;
        }
        *reinterpret_cast<Pack<T, CH> *>(out + (row0 + rr) * ld + c0) = o;
      }
    }
  }
  if (want_stats) {  // the tile's channel sums: the row subsets meet in LDS
    if (active) {
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        red[0][sub * ld + pc * CH + j] = s[j];
        red[1][sub * ld + pc * CH + j] = ss[j];
      }
    }
    __syncthreads();
    for (int c = tid; c < ld; c += 256) {
      float a_ = 0.f, b_ = 0.f;
      for (int u = 0; u < nsub; ++u) { a_ += red[0][u * ld + c]; b_ += red[1][u * ld + c]; }
      st_sum[tile * ld + c] = a_;
      st_sq[tile * ld + c] = b_;
    }
  }
}

template <typename T>
int launch_rows(const SlideOp &o, hipStream_t s) {
  if (o.i[0] <= 0) return 0;  // no samples / rows / points: nothing to launch (a zero-sized grid is a launch error)
  switch (o.kind) {
    case SLIDE_OP_ROWS_FROM_NCX: {  // i: B, C, P, ld
      const int B = o.i[0], C = o.i[1], P = o.i[2], ld = o.i[3];
      if (P <= 0) return 0;
      hipLaunchKernelGGL(rows_from_ncx_kernel<T>, dim3(ld / 32, (P + 31) / 32, B), dim3(256), 0, s, C, P, ld,
                         (const float *)o.p[0], (T *)o.p[1]);
      break;
    }
    case SLIDE_OP_ROWS_TO_NCX: {
      const int B = o.i[0], C = o.i[1], P = o.i[2], ld = o.i[3];
      if (P <= 0 || C <= 0) return 0;
      hipLaunchKernelGGL(rows_to_ncx_kernel<T>, dim3((C + 31) / 32, (P + 31) / 32, B), dim3(256), 0, s, C, P, ld,
                         (const T *)o.p[0], (float *)o.p[1]);
      break;
    }
    case SLIDE_OP_ROWS_GROUP: {  // i: B, N, np, K, C, ldf, ldg, flags
      const int B = o.i[0], N = o.i[1], np = o.i[2], K = o.i[3], C = o.i[4], ldf = o.i[5], ldg = o.i[6], flags = o.i[7];
      if (ldg % 8 || (C > 0 && ldf % 8) || C + ((flags & 1) ? 11 : (flags & 8) ? 0 : 3 + ((flags & 2) ? 3 : 0) + ((flags & 4) ? 3 : 0)) > ldg) return -3;
      const size_t total = (size_t)B * np * K * (ldg / 8);
      if (total == 0) return 0;
      hipLaunchKernelGGL(rows_group_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, N, np, K, C, ldf, ldg,
                         flags, (const float *)o.p[0], (const float *)o.p[1], (const T *)o.p[2], (const void *)o.p[3],
                         (const float *)o.p[4], (const int *)o.p[6], (T *)o.p[5], total);
      break;
    }
    case SLIDE_OP_ROWS_GN: {  // i: B, S, ld, G, n_norm, flags, addvec_ld, res_ld   p: x, gamma, beta, addvec, residual, part, y, [10] optional mean / rstd out [B][64][2]
      const int B = o.i[0], S = o.i[1], ld = o.i[2], G = o.i[3], n_norm = o.i[4], flags = o.i[5];
      if (ld % 32 || ld > 1024 || G > 64 || (G > 0 && n_norm % G)) return -3;
      if (S <= 0) return 0;
      const int rt = 256 / (ld / (16 / (int)sizeof(T)));
      int nchunk = (S + rt * 8 - 1) / (rt * 8);
      nchunk = nchunk < 1 ? 1 : nchunk > 64 ? 64 : nchunk;
      const int rpc = (S + nchunk - 1) / nchunk;
      nchunk = (S + rpc - 1) / rpc;
      // scratch: [B][nchunk <= 64][ld][2] partial sums, then [B][2][ld] scale / shift (or the caller's p[9]).
      // flags & 4: statistics + finalisation only (the consumer GEMM applies scale / shift while it loads: deferred
      // normalisation); flags & 8: apply only, with the scale / shift of an earlier finalise-only call in p[9]
      float *ssp = G > 0 ? (o.p[9] ? (float *)o.p[9] : (float *)o.p[5] + (size_t)B * 64 * ld * 2) : nullptr;
      if (G > 0 && !o.p[7] && !(flags & 8))
        hipLaunchKernelGGL(rows_gn_stats_kernel<T>, dim3(nchunk, B), dim3(256), 0, s, S, ld, rpc, flags & 1, (const T *)o.p[0],
                           (float *)o.p[5]);
      if (G > 0 && !(flags & 8))
        hipLaunchKernelGGL(rows_gn_finalize_kernel, dim3(B), dim3(256), 0, s, S, ld, nchunk, G, n_norm, (const float *)o.p[5],
                           (const float *)o.p[1], (const float *)o.p[2], (const float *)o.p[7], (const float *)o.p[8], o.i[8],
                           ssp, (float *)o.p[10]);
      if (!(flags & 4))
        hipLaunchKernelGGL(rows_gn_apply_kernel<T>, dim3(nchunk, B), dim3(256), 0, s, S, ld, rpc, flags, (const T *)o.p[0], ssp,
                           (const float *)o.p[3], o.i[6], (const T *)o.p[4], o.i[7], (T *)o.p[6]);
      break;
    }
    case SLIDE_OP_ROWS_CONCAT_QK: {  // i: rows, K, C1, ldq, C2, ldk, ldo   p: q, k, out
      const size_t total = (size_t)o.i[0] * (o.i[6] / (16 / (int)sizeof(T)));
      hipLaunchKernelGGL(rows_concat_qk_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, o.i[1], o.i[2],
                         o.i[3], o.i[4], o.i[5], o.i[6], (const T *)o.p[0], (const T *)o.p[1], (T *)o.p[2], total);
      break;
    }
    case SLIDE_OP_ROWS_ATTN: {  // i: points, K, C, lds, ldv, ldo   p: S, V, out
      const size_t total = (size_t)o.i[0] * (o.i[5] / (16 / (int)sizeof(T)));
      hipLaunchKernelGGL(rows_attn_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, o.i[1], o.i[2], o.i[3],
                         o.i[4], o.i[5], (const T *)o.p[0], (const T *)o.p[1], (const int *)o.p[3], (T *)o.p[2], total,
                         (const float *)o.p[4], o.i[6] > 0 ? o.i[6] : 1, o.i[7]);
      break;
    }
    case SLIDE_OP_ROWS_GN_JOINT: {  // i: B, C1, ldq, tq, K, C2, ldk, tk, G (n_norm in f[1])   f: 1 / rows per sample, n_norm   p: qsum, qsq, ksum, ksq, gamma, beta, ssq, ssk
      const int B = o.i[0], C1 = o.i[1], ldq = o.i[2], tq = o.i[3], K = o.i[4], C2 = o.i[5], ldk = o.i[6], tk = o.i[7], G = o.i[8];
      const int n_norm = (int)o.f[1];
      if (C1 + C2 > 1024 || G <= 0 || G > 64 || n_norm % G || n_norm > C1 + C2 || tq <= 0 || tk <= 0 || C1 > ldq || C2 > ldk) return -3;
      for (int k = 0; k < 8; ++k)
        if (!o.p[k]) return -3;
      hipLaunchKernelGGL(rows_gn_joint_kernel, dim3(B), dim3(256), 0, s, C1, ldq, tq, (float)K, C2, ldk, tk, G, n_norm, o.f[0],
                         (const float *)o.p[0], (const float *)o.p[1], (const float *)o.p[2], (const float *)o.p[3],
                         (const float *)o.p[4], (const float *)o.p[5], (float *)o.p[6], (float *)o.p[7]);
      break;
    }
    case SLIDE_OP_ROWS_PAIR_EXPAND: {  // i: B, N, np, K, ld, ldA, flags   p: A, bias, coef, xyz, new_xyz, idx, d2, out, st_sum, st_sq
      const int B = o.i[0], N = o.i[1], np = o.i[2], K = o.i[3], ld = o.i[4], ldA = o.i[5], flags = o.i[6];
      if (ld % 32 || ld <= 0 || ld > 1024 || !o.p[0] || ldA % 4 || ldA < ld || !o.p[1] || !o.p[2] || !o.p[3] || !o.p[4] || !o.p[5] || !o.p[7] ||
          ((flags & 2) && !o.p[6]) || (!o.p[8]) != (!o.p[9]) || N <= 0 || K <= 0)
        return -3;
      if (std::is_same<T, float>::value) return -3;  // fp16 rows only
      const size_t rows = (size_t)B * np * K;
      if (rows == 0) return 0;
      const int S = np * K, tps = S % 256 == 0 ? S / 256 : 0;  // 256-row tiles per sample (0: samples are not whole tiles -> linear map)
      const unsigned grid = tps ? (unsigned)(8 * ((B + 7) / 8) * tps) : (unsigned)((rows + 255) / 256);
      if (ld >= 256)
        hipLaunchKernelGGL((rows_pair_expand_kernel<T, 8>), dim3(grid), dim3(256), 0, s, N, np, K, ld, ldA, flags, tps, rows,
                           (const float *)o.p[0], (const float *)o.p[1], (const float *)o.p[2], (const float *)o.p[3], (const float *)o.p[4],
                           (const void *)o.p[5], (const float *)o.p[6], (T *)o.p[7], (float *)o.p[8], (float *)o.p[9]);
      else
        hipLaunchKernelGGL((rows_pair_expand_kernel<T, 4>), dim3(grid), dim3(256), 0, s, N, np, K, ld, ldA, flags, tps, rows,
                           (const float *)o.p[0], (const float *)o.p[1], (const float *)o.p[2], (const float *)o.p[3], (const float *)o.p[4],
                           (const void *)o.p[5], (const float *)o.p[6], (T *)o.p[7], (float *)o.p[8], (float *)o.p[9]);
      break;
    }
    case SLIDE_OP_ROWS_POOL: {  // i: points, K, C, ldx, ldo, mode   p: x, out, counts
      const size_t total = (size_t)o.i[0] * (o.i[4] / (16 / (int)sizeof(T)));
      hipLaunchKernelGGL(rows_pool_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, o.i[1], o.i[2], o.i[3],
                         o.i[4], o.i[5], (const T *)o.p[0], (const int *)o.p[2], (T *)o.p[1], total);
      break;
    }
    default:
      return -1;
  }
  return (int)hipGetLastError();
}

}  // namespace

// called from engine.hip's op dispatcher; i[9] = 1: fp16 activations
int slide_launch_rows_op(const SlideOp &o, hipStream_t s) {
  return o.i[9] ? launch_rows<half_t>(o, s) : launch_rows<float>(o, s);
}
