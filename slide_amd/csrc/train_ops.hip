// train_ops.hip -- gfx950 BACKWARD kernels of the row-major layers (the training step).
//
// The forward kernels are the module path's (rows_ops.hip: grouping, GroupNorm, relu([q | k]), softmax-weighted sum; engine.hip:
// the MFMA GEMM); this file holds what torch.autograd needs from each of them, on the same [rows][ld] fp32 matrices (ld = channels
// rounded up to 32, pad columns zero).  Reference semantics being differentiated:
//   MyGroupNorm + ReLU            pointnet2_ops_lib/pointnet2_ops/pointnet2_modules.py:24-69
//   relu(cat([q.expand, k]))      pointnet2_ops_lib/pointnet2_ops/attention.py:78-88
//   softmax over K + weighted sum attention.py:89-95
//   grouping_operation / knn_gather on features   pointnet2_utils.py:222-268, :506-507  (the reference's own backward:
//                                                  _ext-src/src/group_points_gpu.cu:30-60 -- an atomicAdd scatter, as here)
// The GEMM's data gradient is the forward GEMM on the transposed weights (slide_amd/train/functions.py); its weight gradient is a
// plain [O x rows] x [rows x I] library GEMM.  HBM-bound kernels: one thread per 4 consecutive channels of a row where rows are
// independent, one workgroup per (sample, group) where GroupNorm couples them.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/slide_train.h"

namespace {

constexpr float GN_EPS = 1e-5f;

__device__ __forceinline__ float block_sum(float v, float *red) {  // sum over the 256 threads of a workgroup (all get it)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// ---------------------------------------------------------------------------------------------------- GroupNorm (+ ReLUs)
// forward (rows_ops.hip): z = pre_relu ? relu(x) : x;  n = (z - mean_g) rstd_g over (gs channels x S rows) of sample b;
//                         g = n gamma + beta;  y = post_relu ? relu(g) : g;   channels >= n_norm: y = relu?(relu?(x)).
// backward, one workgroup per (sample, group): statistics recomputed from x (fp32, two passes over the 256-row x gs-channel
// slab, which stays in the caches), then  dg = dy [g > 0];  dgamma += sum dg n;  dbeta += sum dg;  dn = dg gamma;
//   dz = rstd (dn - mean(dn) - n mean(dn n));  dx = dz [x > 0 if pre_relu].
// dgamma / dbeta: per-sample partials [B][ld] (summed over the batch by the caller: deterministic).
__global__ __launch_bounds__(256) void gn_rows_bwd_kernel(int S, int ld, int G, int n_norm, int flags, const float *__restrict__ x,
                                                          const float *__restrict__ gamma, const float *__restrict__ beta,
                                                          const float *__restrict__ dy, float *__restrict__ dx,
                                                          float *__restrict__ dgamma, float *__restrict__ dbeta) {
  __shared__ float red[4];
  const bool pre_relu = flags & 1, post_relu = flags & 2;
  const int b = blockIdx.y, g = blockIdx.x;
  const int gs = G > 0 ? n_norm / G : 0;
  const size_t base = (size_t)b * S * ld;
  if (g == G) {  // pass-through channels [n_norm, ld): y = post_relu(pre_relu(x))
    const int w = ld - n_norm;
    for (int e = threadIdx.x; e < S * w; e += 256) {
      const int r = e / w, c = n_norm + e - r * w;
      const float xv = x[base + (size_t)r * ld + c];
      float d = dy[base + (size_t)r * ld + c];
      if ((pre_relu || post_relu) && !(xv > 0.f)) d = 0.f;
      dx[base + (size_t)r * ld + c] = d;
    }
    return;
  }
  const int c0 = g * gs, n = S * gs;
  const float inv = 1.0f / (float)n;
  float s = 0.f;
  for (int e = threadIdx.x; e < n; e += 256) {
    const int r = e / gs, c = c0 + e - r * gs;
    float z = x[base + (size_t)r * ld + c];
    if (pre_relu) z = fmaxf(z, 0.f);
    s += z;
  }
  const float mean = block_sum(s, red) * inv;
  float q = 0.f;
  for (int e = threadIdx.x; e < n; e += 256) {
    const int r = e / gs, c = c0 + e - r * gs;
    float z = x[base + (size_t)r * ld + c];
    if (pre_relu) z = fmaxf(z, 0.f);
    q += (z - mean) * (z - mean);
  }
  const float rstd = 1.0f / sqrtf(block_sum(q, red) * inv + GN_EPS);
  // sums of dn and dn * n over the slab; per-channel dgamma / dbeta (a thread visits the channels c0 + (tid + 256 k) % gs: when
  // gs divides 256 always the same one -- accumulated in registers, combined through LDS below)
  float s1 = 0.f, s2 = 0.f;
  for (int e = threadIdx.x; e < n; e += 256) {
    const int r = e / gs, c = c0 + e - r * gs;
    float z = x[base + (size_t)r * ld + c];
    if (pre_relu) z = fmaxf(z, 0.f);
    const float nv = (z - mean) * rstd, gm = gamma[c];
    float dg = dy[base + (size_t)r * ld + c];
    if (post_relu && !(nv * gm + beta[c] > 0.f)) dg = 0.f;
    const float dn = dg * gm;
    s1 += dn;
    s2 += dn * nv;
  }
  const float m1 = block_sum(s1, red) * inv;
  const float m2 = block_sum(s2, red) * inv;
  for (int e = threadIdx.x; e < n; e += 256) {
    const int r = e / gs, c = c0 + e - r * gs;
    const float xv = x[base + (size_t)r * ld + c];
    const float z = pre_relu ? fmaxf(xv, 0.f) : xv;
    const float nv = (z - mean) * rstd, gm = gamma[c];
    float dg = dy[base + (size_t)r * ld + c];
    if (post_relu && !(nv * gm + beta[c] > 0.f)) dg = 0.f;
    float d = rstd * (dg * gm - m1 - nv * m2);
    if (pre_relu && !(xv > 0.f)) d = 0.f;
    dx[base + (size_t)r * ld + c] = d;
  }
  // parameter gradients of this (sample, group): one channel at a time, the workgroup reduces over the sample's rows
  for (int j = 0; j < gs; ++j) {
    const int c = c0 + j;
    const float gm = gamma[c], bt = beta[c];
    float a = 0.f, bsum = 0.f;
    for (int r = threadIdx.x; r < S; r += 256) {
      float z = x[base + (size_t)r * ld + c];
      if (pre_relu) z = fmaxf(z, 0.f);
      const float nv = (z - mean) * rstd;
      float dg = dy[base + (size_t)r * ld + c];
      if (post_relu && !(nv * gm + bt > 0.f)) dg = 0.f;
      a += dg * nv;
      bsum += dg;
    }
    a = block_sum(a, red);
    bsum = block_sum(bsum, red);
    if (threadIdx.x == 0) {
      dgamma[(size_t)b * ld + c] = a;
      dbeta[(size_t)b * ld + c] = bsum;
    }
  }
}

// ---------------------------------------------------------------------------------------------------- grouping (features)
// forward: out[(b, p, k)][c] = feat[b][idx[b][p][k]][c] for c < C (a centre with an empty ball: zero features);
// backward: dfeat[b][idx][c] += dout[row][c].  One thread per (row, 4 channels); fp32 atomics (the reference's backward does the same).
__global__ __launch_bounds__(256) void group_rows_bwd_kernel(int N, int np, int K, int C, int ldf, int ldg,
                                                             const int64_t *__restrict__ idx, const int *__restrict__ counts,
                                                             const float *__restrict__ dout, float *__restrict__ dfeat, size_t total) {
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int ppr = (C + 3) >> 2;
  const size_t row = gid / ppr;
  const int c0 = (int)(gid - row * ppr) * 4;
  const size_t pt = row / K;
  if (counts && counts[pt] == 0) return;
  const int b = (int)(pt / np);
  const int nb = (int)idx[row];
  const float *d = dout + row * ldg + c0;
  float *f = dfeat + ((size_t)b * N + nb) * ldf + c0;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (c0 + j < C) atomicAdd(f + j, d[j]);
}

// ---------------------------------------------------------------------------------------------------- relu([q | k])
// forward: out[row][0..C1) = relu(q[row / K]), out[row][C1..C1+C2) = relu(k[row]).  backward with the forward OUTPUT as the mask:
// dk[row][c] = dout[row][C1 + c] [out > 0];  dq[pt][c] = sum_k dout[(pt, k)][c] [out > 0].
__global__ __launch_bounds__(256) void concat_qk_bwd_kernel(int K, int C1, int ldq, int C2, int ldk, int ldo, const float *__restrict__ out,
                                                            const float *__restrict__ dout, float *__restrict__ dq,
                                                            float *__restrict__ dk, size_t pts) {
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int cw = C1 + C2;
  if (gid >= pts * cw) return;
  const size_t pt = gid / cw;
  const int c = (int)(gid - pt * cw);
  if (c < C1) {
    float a = 0.f;
    for (int k = 0; k < K; ++k) {
      const size_t o = (pt * K + k) * ldo + c;
      a += out[o] > 0.f ? dout[o] : 0.f;
    }
    dq[pt * ldq + c] = a;
  } else {
    for (int k = 0; k < K; ++k) {
      const size_t o = (pt * K + k) * ldo + c;
      dk[(pt * K + k) * ldk + (c - C1)] = out[o] > 0.f ? dout[o] : 0.f;
    }
  }
}

// ---------------------------------------------------------------------------------------------------- softmax + weighted sum
// forward: w = softmax over the first max(1, count) of the K neighbour rows of a point, out[pt][c] = sum_k w[k][c] v[(pt, k)][c].
// backward: dv[k] = w[k] dout;  ds[k] = w[k] (v[k] - out) dout   (rows beyond the count: 0).
__global__ __launch_bounds__(256) void attn_rows_bwd_kernel(int K, int C, int lds, int ldv, int ldo, const float *__restrict__ s,
                                                            const float *__restrict__ v, const int *__restrict__ counts,
                                                            const float *__restrict__ dout, float *__restrict__ ds,
                                                            float *__restrict__ dv, size_t pts) {
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= pts * C) return;
  const size_t pt = gid / C;
  const int c = (int)(gid - pt * C);
  int cnt = K;
  if (counts) {
    cnt = counts[pt];
    cnt = cnt < 1 ? 1 : (cnt > K ? K : cnt);
  }
  float m = -INFINITY;
  for (int k = 0; k < cnt; ++k) m = fmaxf(m, s[(pt * K + k) * lds + c]);
  float den = 0.f, num = 0.f;
  for (int k = 0; k < cnt; ++k) {
    const float e = expf(s[(pt * K + k) * lds + c] - m);
    den += e;
    num += e * v[(pt * K + k) * ldv + c];
  }
  const float o = num / den, d = dout[pt * ldo + c];
  for (int k = 0; k < K; ++k) {
    float w = 0.f, vv = 0.f;
    if (k < cnt) {
      w = expf(s[(pt * K + k) * lds + c] - m) / den;
      vv = v[(pt * K + k) * ldv + c];
    }
    dv[(pt * K + k) * ldv + c] = w * d;
    ds[(pt * K + k) * lds + c] = w * (vv - o) * d;
  }
}

#define LAUNCH_STATUS() ((int)hipGetLastError())

}  // namespace

extern "C" {

int slide_gn_rows_bwd(int B, int S, int ld, int G, int n_norm, int flags, const float *x, const float *gamma, const float *beta,
                      const float *dy, float *dx, float *dgamma, float *dbeta, slide_stream_t stream) {
  if (B <= 0 || S <= 0) return 0;
  if (ld % 4 || G < 0 || n_norm < 0 || n_norm > ld || (G > 0 && n_norm % G) || (G == 0 && n_norm != 0)) return -3;
  const int extra = n_norm < ld ? 1 : 0;
  hipLaunchKernelGGL(gn_rows_bwd_kernel, dim3(G + extra, B), dim3(256), 0, (hipStream_t)stream, S, ld, G, n_norm, flags, x, gamma, beta,
                     dy, dx, dgamma, dbeta);
  return LAUNCH_STATUS();
}

int slide_group_rows_bwd(int B, int N, int np, int K, int C, int ldf, int ldg, const int64_t *idx, const int *counts,
                         const float *dout, float *dfeat, slide_stream_t stream) {
  const size_t total = (size_t)B * np * K * ((C + 3) / 4);
  if (total == 0) return 0;
  hipLaunchKernelGGL(group_rows_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, N, np, K, C, ldf,
                     ldg, idx, counts, dout, dfeat, total);
  return LAUNCH_STATUS();
}

int slide_concat_qk_bwd(long long pts, int K, int C1, int ldq, int C2, int ldk, int ldo, const float *out, const float *dout, float *dq,
                        float *dk, slide_stream_t stream) {
  const size_t total = (size_t)pts * (C1 + C2);
  if (total == 0) return 0;
  hipLaunchKernelGGL(concat_qk_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, K, C1, ldq, C2, ldk,
                     ldo, out, dout, dq, dk, (size_t)pts);
  return LAUNCH_STATUS();
}

int slide_attn_rows_bwd(long long pts, int K, int C, int lds, int ldv, int ldo, const float *s, const float *v, const int *counts,
                        const float *dout, float *ds, float *dv, slide_stream_t stream) {
  const size_t total = (size_t)pts * C;
  if (total == 0) return 0;
  hipLaunchKernelGGL(attn_rows_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, K, C, lds, ldv, ldo,
                     s, v, counts, dout, ds, dv, (size_t)pts);
  return LAUNCH_STATUS();
}

}  // extern "C"
