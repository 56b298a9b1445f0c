// pair_norm.h -- the pair-table pass of the pair decomposition (SLIDE_OP_PAIR_NORM version 2) as a device body
// shared by gemm_gx.hip (its own launch) and gemm_gxs.hip (a step of the per-point stage kernel of the split plans).
#pragma once
#include "gemm_common.h"

namespace {

// Version 2 (the default): ONE workgroup of 1024 threads per sample covers every channel, so the sample's key statistics
// are complete inside the workgroup and the joint GroupNorm over the attention's [query | key] concatenation
// (weight_conv.1, attention.py:45-47; SlideGnFin in include/slide_engine.h) is finalised here as well -- per-channel scale /
// shift for the query GEMM and the generated-X key GEMM -- instead of inside the next launch's prologue.
// TT = float (round 5): the tables of the SPLIT-arithmetic plans (fp32-grade position DDPM) -- same pass, tables kept in fp32
// (device body: sample b, nthr threads of the workgroup, dyn_l = [16][NT + 1] floats of LDS for the FP form; called by
//  pair_norm2_kernel and by the per-point stage kernel of the split plans, gemm_gxs.hip)
template <bool FP, typename TT, int NT>
__device__ __forceinline__ void pair_norm2_body(int ld, const float *__restrict__ y, const float *__restrict__ xyz,
                                                const float *__restrict__ wa, const float *__restrict__ wb,
                                                const SlideEpi *__restrict__ epi, TT *__restrict__ ta,
                                                TT *__restrict__ tb, const int *__restrict__ nbr,
                                                const float *__restrict__ d2t, const float *__restrict__ wt,
                                                const float *__restrict__ vv_in, float *__restrict__ vv_out,
                                                const SlideGnFin *__restrict__ finp, const int b, const int nthr, float *dyn_l) {
  __shared__ float sx[48];
  __shared__ int sq[16 * 8];
  __shared__ float sd[16 * 8], sw[16 * 8];
  __shared__ float csum[2048], csq[2048];  // per-channel sums in the [query | key] channel space
  __shared__ float grp[32][2];
  const int tid = threadIdx.x;
  if (tid < 48) sx[tid] = xyz[(size_t)b * 48 + tid];
  if (FP && tid < 128) {
    const int slot = (b * 16 + (tid >> 3)) * 16 + (tid & 7);
    sq[tid] = nbr[slot]; sd[tid] = d2t[slot]; sw[tid] = wt[slot];
  }
  SlideGnFin f = {};
  if (finp) {
    f = *finp;
    for (int c = tid; c < f.C; c += nthr) {  // query half: published by the query GEMM; key half: overwritten below
      csum[c] = f.sum[(size_t)b * f.bs + c];
      csq[c] = f.sq[(size_t)b * f.bs + c];
    }
  }
  __syncthreads();
  for (int c = tid; c < ld; c += nthr) {  // (whole waves: ld is a multiple of 32 and group sizes divide 32)
    const SlideEpi e = epi[c >> 5];
    const int cl = c & 31;
    float av[16], bv[16];
    const float4 ca = *reinterpret_cast<const float4 *>(wa + (size_t)c * 4), cb = *reinterpret_cast<const float4 *>(wb + (size_t)c * 4);
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const float x0 = sx[p * 3], x1 = sx[p * 3 + 1], x2 = sx[p * 3 + 2];
      av[p] = y[((size_t)b * 16 + p) * ld + c] + (ca.x * x0 + ca.y * x1 + ca.z * x2);
      bv[p] = cb.x * x0 + cb.y * x1 + cb.z * x2;
    }
    float vd = 0.f, vw = 0.f;
    if (FP) { vd = vv_in[c]; vw = vv_in[ld + c]; }
    float g = 1.f, sh = 0.f;
    if (e.mode != SLIDE_EPI_RAW) {
      const bool pre_relu = (e.flags & SLIDE_F_PRE_RELU) != 0;
      float s = 0.f, ss = 0.f;
      if (!FP && !pre_relu) {
        // all 16 x 16 pairs of a[q] + b[p] in closed form: sum = 16 (A + B), sum of squares = 16 (A2 + B2) + 2 A B
        float A = 0.f, Bs = 0.f, A2 = 0.f, B2 = 0.f;
#pragma unroll
        for (int p = 0; p < 16; ++p) { A += av[p]; Bs += bv[p]; A2 = fmaf(av[p], av[p], A2); B2 = fmaf(bv[p], bv[p], B2); }
        s = 16.f * (A + Bs);
        ss = fmaf(2.f * A, Bs, 16.f * (A2 + B2));
      } else if (!FP) {
#pragma unroll
        for (int p = 0; p < 16; ++p)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const float v = fmaxf(av[q] + bv[p], 0.f);
            s += v; ss = fmaf(v, v, ss);
          }
      } else {
#pragma unroll
        for (int p = 0; p < 16; ++p) dyn_l[p * (NT + 1) + tid] = av[p];  // (a thread reads back only its own column: no barrier)
#pragma unroll
        for (int p = 0; p < 16; ++p)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int sl = p * 8 + j;
            float v = dyn_l[sq[sl] * (NT + 1) + tid] + bv[p] + sd[sl] * vd + sw[sl] * vw;
            if (pre_relu) v = fmaxf(v, 0.f);
            s += v; ss = fmaf(v, v, ss);
          }
      }
      if (e.mode == SLIDE_EPI_STATS) {
        if (finp) {  // (the statistics tensors ARE the joint GroupNorm's sum / sq rows)
          const int idx = (int)(e.stats_sum - f.sum) + cl;
          csum[idx] = s * e.stats_scale;
          csq[idx] = ss * e.stats_scale;
        } else {
          e.stats_sum[(size_t)b * e.stats_bs + cl] = s * e.stats_scale;
          e.stats_sq[(size_t)b * e.stats_bs + cl] = ss * e.stats_scale;
        }
      } else {  // NORM: groups of e.gs physical channels (a power of two <= 32: lanes of one wave)
        for (int m = 1; m < e.gs; m <<= 1) {
          s += __shfl_xor(s, m, 64);
          ss += __shfl_xor(ss, m, 64);
        }
        const float mean = s * e.inv_count;
        const float var = fmaxf(ss * e.inv_count - mean * mean, 0.f);
        g = e.gamma[cl] * __builtin_amdgcn_rsqf(var + GN_EPS);
        sh = e.beta[cl] - mean * g;
        if (cl >= e.n_norm) { g = 1.f; sh = 0.f; }  // MyGroupNorm leaves the last C % G channels as they are
      }
    }
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      ta[((size_t)b * 16 + p) * ld + c] = (TT)(av[p] * g + sh);
      tb[((size_t)b * 16 + p) * ld + c] = (TT)(bv[p] * g);
    }
    if (FP) {
      vv_out[(size_t)b * 2 * ld + c] = vd * g;
      vv_out[(size_t)b * 2 * ld + ld + c] = vw * g;
    }
  }
  if (!finp) return;
  __syncthreads();
  if (tid < 32) {  // one group per thread, fixed summation order (deterministic)
    float mean = 0.f, rstd = 0.f;
    if (tid < f.G) {
      float S = 0.f, SS = 0.f;
      const int c_end = f.gend[tid];
      for (int c = f.gstart[tid]; c < c_end; ++c) { S += csum[c]; SS += csq[c]; }
      mean = S * f.inv_count;
      const float var = fmaxf(SS * f.inv_count - mean * mean, 0.f);
      rstd = 1.0f / sqrtf(var + GN_EPS);
    }
    grp[tid][0] = mean; grp[tid][1] = rstd;
  }
  __syncthreads();
  for (int c = tid; c < f.C; c += nthr) {
    const int g = f.gid[c];
    float sc = 1.f, sh = 0.f;
    if (g >= 0) {
      sc = f.gamma[c] * grp[g][1];
      sh = f.beta[c] - grp[g][0] * sc;
    }
    f.scale[(size_t)b * f.bs + c] = sc;
    f.shift[(size_t)b * f.bs + c] = sh;
  }
}

}  // namespace
