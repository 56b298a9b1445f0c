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
// independent; GroupNorm's backward as column reductions + an elementwise pass on the forward's thread layout.
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
// backward:  dg = dy [g > 0];  dgamma_c = sum_rows dg n;  dbeta_c = sum_rows dg;  dn = dg gamma;
//            dz = rstd (dn - mean_g(dn) - n mean_g(dn n));  dx = dz [x > 0 if pre_relu],
// and the group means follow from the per-channel sums: mean_g(dn) = sum_{c in g} gamma_c dbeta_c / n, mean_g(dn n) likewise from
// dgamma_c.  So the backward is the forward's shape again -- a per-channel column reduction over the sample's rows, a tiny
// per-sample finalisation, an elementwise pass -- on the forward's thread layout: a thread owns 4 consecutive channels, cn = ld / 4
// threads cover a row, rt = 256 / cn rows are in flight per workgroup step; every access is a full coalesced row.  mean / rstd
// come from the forward (SLIDE_OP_ROWS_GN p[10]).  HBM traffic: x and dy twice, dx once (5 passes; the first version's one
// workgroup per (sample, group) read 16-byte slivers of every row and ran at 3 % of the HBM rate).
// dgamma / dbeta: per-sample partials [B][ld] (summed over the batch by the caller: deterministic).
struct GnCh {  // per-thread constants of its 4 channels
  float gam[4], bet[4], mean[4], rstd[4];
  bool norm[4];
};

__device__ __forceinline__ GnCh gn_channels(int b, int c0, int gs, int n_norm, const float *__restrict__ gamma,
                                            const float *__restrict__ beta, const float *__restrict__ mr) {
  GnCh k;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = c0 + j;
    k.norm[j] = c < n_norm;
    const int g = k.norm[j] ? c / gs : 0;
    k.gam[j] = k.norm[j] ? gamma[c] : 1.f;
    k.bet[j] = k.norm[j] ? beta[c] : 0.f;
    k.mean[j] = k.norm[j] ? mr[((size_t)b * 64 + g) * 2 + 0] : 0.f;
    k.rstd[j] = k.norm[j] ? mr[((size_t)b * 64 + g) * 2 + 1] : 1.f;
  }
  return k;
}

// pass 1: part[b][chunk][ld][2] = per-channel (sum dg, sum dg n) over the chunk's rows
__global__ __launch_bounds__(256) void gn_bwd_sums_kernel(int S, int ld, int rpc, int gs, int n_norm, int flags, const float *__restrict__ x,
                                                          const float *__restrict__ dy, const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, const float *__restrict__ mr,
                                                          float *__restrict__ part) {
  __shared__ float red[256 * 4 * 2];
  const bool pre_relu = flags & 1, post_relu = flags & 2;
  const int cn = ld / 4, rt = 256 / cn;
  const int pr = threadIdx.x / cn, pc = threadIdx.x - pr * cn;
  const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
  const int r0 = chunk * rpc, r1 = min(S, r0 + rpc);
  if (pr < rt) {
    const GnCh k = gn_channels(b, pc * 4, gs, n_norm, gamma, beta, mr);
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const size_t base = ((size_t)b * S) * ld + pc * 4;
    for (int r = r0 + pr; r < r1; r += rt) {
      const float4 xv = *reinterpret_cast<const float4 *>(x + base + (size_t)r * ld);
      const float4 dv = *reinterpret_cast<const float4 *>(dy + base + (size_t)r * ld);
      const float xa[4] = {xv.x, xv.y, xv.z, xv.w}, da[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float z = pre_relu ? fmaxf(xa[j], 0.f) : xa[j];
        const float nv = (z - k.mean[j]) * k.rstd[j];
        float d = da[j];
        if (post_relu && !(nv * k.gam[j] + k.bet[j] > 0.f)) d = 0.f;
        s1[j] += d;
        s2[j] += d * nv;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      red[(pr * ld + pc * 4 + j) * 2 + 0] = s1[j];
      red[(pr * ld + pc * 4 + j) * 2 + 1] = s2[j];
    }
  }
  __syncthreads();
  float *po = part + (((size_t)b * nchunk + chunk) * ld) * 2;
  for (int c = threadIdx.x; c < ld; c += 256) {
    float a = 0.f, q = 0.f;
    for (int r = 0; r < rt; ++r) {
      a += red[(r * ld + c) * 2 + 0];
      q += red[(r * ld + c) * 2 + 1];
    }
    po[c * 2 + 0] = a;
    po[c * 2 + 1] = q;
  }
}

// between the passes, once per sample: dgamma / dbeta of the sample and the two group means -> coef[b][g][2]
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(int S, int ld, int nchunk, int G, int n_norm, const float *__restrict__ part,
                                                              const float *__restrict__ gamma, float *__restrict__ dgamma,
                                                              float *__restrict__ dbeta, float *__restrict__ coef) {
  __shared__ float l1[1024], l2[1024];
  const int b = blockIdx.x;
  const float *pp = part + ((size_t)b * nchunk * ld) * 2;
  for (int c = threadIdx.x; c < ld; c += 256) {
    float a = 0.f, q = 0.f;
    if (c < n_norm) {
      for (int k = 0; k < nchunk; ++k) {
        a += pp[((size_t)k * ld + c) * 2 + 0];
        q += pp[((size_t)k * ld + c) * 2 + 1];
      }
      l1[c] = gamma[c] * a;
      l2[c] = gamma[c] * q;
    }
    dbeta[(size_t)b * ld + c] = a;
    dgamma[(size_t)b * ld + c] = q;
  }
  __syncthreads();
  const int gs = n_norm / G;
  const float inv = 1.0f / ((float)gs * (float)S);
  for (int g = threadIdx.x; g < G; g += 256) {
    float a = 0.f, q = 0.f;
    for (int j = 0; j < gs; ++j) {
      a += l1[g * gs + j];
      q += l2[g * gs + j];
    }
    coef[((size_t)b * 64 + g) * 2 + 0] = a * inv;
    coef[((size_t)b * 64 + g) * 2 + 1] = q * inv;
  }
}

// pass 2: dx.  gs == 0: no normalisation anywhere (ReLUs only; mr / coef unused)
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(int S, int ld, int rpc, int gs, int n_norm, int flags, const float *__restrict__ x,
                                                           const float *__restrict__ dy, const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, const float *__restrict__ mr,
                                                           const float *__restrict__ coef, float *__restrict__ dx) {
  const bool pre_relu = flags & 1, post_relu = flags & 2;
  const int cn = ld / 4, rt = 256 / cn;
  const int pr = threadIdx.x / cn, pc = threadIdx.x - pr * cn;
  const int b = blockIdx.y, chunk = blockIdx.x;
  if (pr >= rt) return;
  const GnCh k = gn_channels(b, pc * 4, gs, n_norm, gamma, beta, mr);
  float m1[4], m2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int g = k.norm[j] ? (pc * 4 + j) / gs : 0;
    m1[j] = k.norm[j] ? coef[((size_t)b * 64 + g) * 2 + 0] : 0.f;
    m2[j] = k.norm[j] ? coef[((size_t)b * 64 + g) * 2 + 1] : 0.f;
  }
  const int r0 = chunk * rpc, r1 = min(S, r0 + rpc);
  const size_t base = ((size_t)b * S) * ld + pc * 4;
  for (int r = r0 + pr; r < r1; r += rt) {
    const float4 xv = *reinterpret_cast<const float4 *>(x + base + (size_t)r * ld);
    const float4 dv = *reinterpret_cast<const float4 *>(dy + base + (size_t)r * ld);
    const float xa[4] = {xv.x, xv.y, xv.z, xv.w}, da[4] = {dv.x, dv.y, dv.z, dv.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float d = da[j];
      if (k.norm[j]) {
        const float z = pre_relu ? fmaxf(xa[j], 0.f) : xa[j];
        const float nv = (z - k.mean[j]) * k.rstd[j];
        if (post_relu && !(nv * k.gam[j] + k.bet[j] > 0.f)) d = 0.f;
        d = k.rstd[j] * (d * k.gam[j] - m1[j] - nv * m2[j]);
        if (pre_relu && !(xa[j] > 0.f)) d = 0.f;
      } else if ((pre_relu || post_relu) && !(xa[j] > 0.f)) {  // pass-through channel: y = relu?(relu?(x))
        d = 0.f;
      }
      o[j] = d;
    }
    *reinterpret_cast<float4 *>(dx + base + (size_t)r * ld) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// ---------------------------------------------------------------------------------------------------- column sums (bias gradient)
// part[chunk][c] = sum over the chunk's rows of x[row][c] for the `width` columns of stripe blockIdx.y (same thread layout over the
// stripe: width / 4 threads per row).  Two launches make a full column sum: full-width row chunks that fill the chip, then one
// 32-column stripe per workgroup over the <= 1024 partial rows (deterministic; no atomics).
__global__ __launch_bounds__(256) void col_sums_kernel(long long rows, int ld, int width, int rpc, const float *__restrict__ x,
                                                       float *__restrict__ part) {
  __shared__ float red[256 * 4];
  const int cn = width / 4, rt = 256 / cn;
  const int pr = threadIdx.x / cn, pc = threadIdx.x - pr * cn;
  const int c0 = blockIdx.y * width;
  const long long r0 = (long long)blockIdx.x * rpc, r1 = min(rows, r0 + rpc);
  if (pr < rt) {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (long long r = r0 + pr; r < r1; r += rt) {
      const float4 v = *reinterpret_cast<const float4 *>(x + (size_t)r * ld + c0 + pc * 4);
      s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[pr * width + pc * 4 + j] = s[j];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < width; c += 256) {
    float a = 0.f;
    for (int r = 0; r < rt; ++r) a += red[r * width + c];
    part[(size_t)blockIdx.x * ld + c0 + c] = a;
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
                      const float *mean_rstd, const float *dy, float *dx, float *dgamma, float *dbeta, float *scratch,
                      slide_stream_t stream) {
  if (B <= 0 || S <= 0) return 0;
  if (ld % 32 || ld > 1024 || G < 0 || G > 64 || n_norm < 0 || n_norm > ld || (G > 0 && n_norm % G) || (G == 0 && n_norm != 0)) return -3;
  if (G > 0 && (!mean_rstd || !scratch || !gamma || !beta || !dgamma || !dbeta)) return -3;
  const int rt = 256 / (ld / 4);
  int nchunk = (S + rt * 4 - 1) / (rt * 4);
  nchunk = nchunk < 1 ? 1 : nchunk > 64 ? 64 : nchunk;
  const int rpc = (S + nchunk - 1) / nchunk;
  nchunk = (S + rpc - 1) / rpc;
  const int gs = G > 0 ? n_norm / G : 0;
  float *coef = scratch ? scratch + (size_t)B * 64 * ld * 2 : nullptr;
  hipStream_t st = (hipStream_t)stream;
  if (G > 0) {
    hipLaunchKernelGGL(gn_bwd_sums_kernel, dim3(nchunk, B), dim3(256), 0, st, S, ld, rpc, gs, n_norm, flags, x, dy, gamma, beta, mean_rstd,
                       scratch);
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(B), dim3(256), 0, st, S, ld, nchunk, G, n_norm, scratch, gamma, dgamma, dbeta, coef);
  }
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(nchunk, B), dim3(256), 0, st, S, ld, rpc, gs, G > 0 ? n_norm : 0, flags, x, dy, gamma, beta,
                     mean_rstd, coef, dx);
  return LAUNCH_STATUS();
}

int slide_col_sums(long long rows, int ld, const float *x, float *out, float *scratch, slide_stream_t stream) {
  if (ld % 32 || ld > 1024 || rows < 0) return -3;
  long long nchunk = rows / 64;
  nchunk = nchunk < 1 ? 1 : nchunk > 1024 ? 1024 : nchunk;
  const int rpc = (int)((rows + nchunk - 1) / nchunk);
  if (nchunk == 1) {
    hipLaunchKernelGGL(col_sums_kernel, dim3(1, ld / 32), dim3(256), 0, (hipStream_t)stream, rows, ld, 32, rpc < 1 ? 1 : rpc, x, out);
    return LAUNCH_STATUS();
  }
  if (!scratch) return -3;
  hipLaunchKernelGGL(col_sums_kernel, dim3((unsigned)nchunk), dim3(256), 0, (hipStream_t)stream, rows, ld, ld, rpc, x, scratch);
  hipLaunchKernelGGL(col_sums_kernel, dim3(1, ld / 32), dim3(256), 0, (hipStream_t)stream, nchunk, ld, 32, (int)nchunk, scratch, out);
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
