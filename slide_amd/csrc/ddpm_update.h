// ddpm_update.h -- the DDPM update arithmetic shared by the update kernels (engine.hip) and the launches that carry an update in
// their epilogue (engine.hip head_update_kernel, point_chain.hip): Philox4x32-10 + Box-Muller noise, the device-side timestep
// advance, one element of the feature DDPM's denoising step.  Anonymous namespace: every translation unit gets its own copy.
#pragma once
#include "gemm_common.h"

namespace {

// Philox4x32-10 counter RNG + Box-Muller (used when no explicit noise tensor is supplied)
__device__ __forceinline__ void philox_round(uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3, uint32_t k0,
                                             uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1,
                 n3 = (uint32_t)p0;
  c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
// counter = (element, step, chain nonce, constant), key = seed: every chain a sampler starts (t_dev[3], bumped by the host
// side's begin()) draws its own noise trajectory, like the reference's fresh torch.randn per batch
__device__ __forceinline__ float philox_normal(uint32_t seed_lo, uint32_t seed_hi, uint32_t step, uint32_t elem,
                                               uint32_t nonce) {
  uint32_t c0 = elem, c1 = step, c2 = 0x243F6A88u ^ nonce, c3 = 0x85A308D3u, k0 = seed_lo, k1 = seed_hi;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    philox_round(c0, c1, c2, c3, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  const float u1 = ((float)(c0 >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float u2 = ((float)(c1 >> 8) + 0.5f) * (1.0f / 16777216.0f);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

// The update kernel also advances the device-side timestep: every block read t / step at its start, so the block that
// finishes LAST (a counter in t_dev[2]) may write t - 1 / step + 1 for the next replay -- one launch less per step.
__device__ __forceinline__ void advance_t_last_block(int *t_dev, int t, int step) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(&t_dev[2], 1) == (int)gridDim.x - 1) {
      t_dev[2] = 0;
      t_dev[0] = t - 1;
      t_dev[1] = step + 1;
    }
  }
}

__device__ __forceinline__ void update_feat_element(int e, int npts, int C, int kdim, int eps_ld, float clamp,
                                                    uint32_t seed_lo, uint32_t seed_hi, float *__restrict__ x,
                                                    const float *__restrict__ eps, const float *__restrict__ noise, int t,
                                                    int step, uint32_t nonce, uint32_t eoff, const float *__restrict__ complete_x0,
                                                    const float *__restrict__ kmask, const float *__restrict__ keypoint,
                                                    const float *__restrict__ rc, const float *__restrict__ rm1,
                                                    const float *__restrict__ c1, const float *__restrict__ c2,
                                                    const float *__restrict__ stdv, void *__restrict__ feat0 = nullptr,
                                                    int ldf = 0, int half_out = 0,
                                                    const SlidePrepCopy *__restrict__ copies = nullptr, int n_copies = 0,
                                                    float eps_val = 0.f,  // (eps == nullptr: the prediction is eps_val)
                                                    const float *z_pre = nullptr) {  // (the element's noise, drawn earlier)
#pragma clang fp contract(off)
  if (e >= npts * C) return;
  const int p = e / C, c = e - p * C;
  if (c < kdim) {
    x[e] = keypoint[(size_t)p * kdim + c];
    return;
  }
  const float xv = x[e];
  float x0 = rc[t] * xv - rm1[t] * (eps ? eps[eps_ld ? (size_t)p * eps_ld + c : (size_t)e] : eps_val);
  if (clamp > 0.f) x0 = fminf(fmaxf(x0, -clamp), clamp);
  if (complete_x0) {  // local re-sampling (diffusion.py:76-79): pred_xstart*mask + complete_x0*(1-mask), mask per point
    const float m = kmask[p];
    x0 = x0 * m + complete_x0[e] * (1.f - m);
  }
  float v = c1[t] * x0 + c2[t] * xv;
  if (t > 0) {
    const float z = z_pre ? *z_pre
                          : noise ? noise[(size_t)step * npts * C + e]
                                  : philox_normal(seed_lo, seed_hi, (uint32_t)step, (uint32_t)e + eoff, nonce);
    v = v + stdv[t] * z;
  }
  x[e] = v;
  // fixed key points (kdim == 3): what the NEXT step's point preparation would derive from this element -- the feature
  // column of the per-point table and of the concatenation buffers that carry the input features -- is written here, and
  // SLIDE_OP_PREP_POINTS (coordinates, neighbour tables: constant over the chain) leaves the step plan
  if (feat0) {
    const int cf = c - kdim;
    if (half_out) reinterpret_cast<_Float16 *>(feat0)[(size_t)p * ldf + cf] = (_Float16)v;
    else reinterpret_cast<float *>(feat0)[(size_t)p * ldf + cf] = v;
    for (int q = 0; q < n_copies; ++q) {
      const SlidePrepCopy cp = copies[q];
      if (cp.kind == 0 && cf < cp.n) {
        if (half_out) reinterpret_cast<_Float16 *>(cp.dst)[(size_t)p * cp.ld + cf] = (_Float16)v;
        else reinterpret_cast<float *>(cp.dst)[(size_t)p * cp.ld + cf] = v;
      }
    }
  }
}


}  // namespace
