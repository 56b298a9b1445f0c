// point_chain.hip -- the per-point END of a denoiser step as ONE launch (round 5): the last FP block's second Mlp
// (Mlp_plus_t_emb: first_mlp + res_connect, second_mlp + fc_condition + residual; pointnet2_modules.py:119-176, :842-855) and the
// output head fc_lyaer (conv -> GroupNorm(32, 128) -> ReLU -> conv, pointnet2_with_pcld_condition.py:480-483) -- four dependent
// 16-rows-per-sample GEMMs whose launches are pure latency on every chain's critical path (timing ablation, tools/ab/r05_ablate.sh: the
// twelve 16-row launches of a feature step cost 29 % of it with 0.3 % of its FLOPs).
//
// One workgroup owns 32 rows (two samples), four waves.  Every weight of the four layers (<= 98 + 32 + 40 + 16 KB) is loaded at kernel
// start straight into MFMA A-fragment registers by the wave that uses it -- wave w owns channel block w of every layer -- together
// with the input rows and all vectors (one L2 round trip for the whole chain); activations cross the waves through LDS as fp16
// [row][channel]; accumulators are D[channel][row] (lane = row, register = channel), so a GroupNorm group (four consecutive channels)
// is four registers of a lane and its statistics over the sample's 16 rows are one 16-lane DPP reduction.  The residual stays in fp32
// registers; the block's output is also written to the head's input buffer (diagnostics read it), the prediction to eps [rows][eps_ld].
// WIDE (round 6, the default of fp16 plans: Wz_lo ... W1_lo set): these four layers are the END of the denoiser -- their operand
// rounding reaches the prediction unattenuated (numpy emulation, tools/prec_select_feat.py: wide operands here alone halve the
// feature net's forward error, 6.2e-4 -> 3.2e-4 relative L2) and they hold 0.3 % of its FLOPs -- so they run in the SPLIT arithmetic of
// the position plan: every weight as w = hi + 2^-11 lo' (two fp16 fragments), every activation that crosses the waves as two fp16
// planes x = xh + 2^-11 xl', three MFMAs per product (hi.xh into one accumulator set; hi.xl' + lo'.xh into a second, scaled by 2^-11
// at the end) -- fp32-grade products; the input rows Z are fp16 as stored (exact: two MFMAs).  The lo fragments of the head's
// weights are loaded after layer 1 (into the registers its weights leave).
// fuse_update (OPT-IN, SLIDE_POINT_CHAIN_UPDATE=1): the launch also applies the feature DDPM's update (update_feat_element,
// ddpm_update.h) and advances the device-side timestep -- the lane that holds eps[row][channel] updates x[row][channel]; all 256
// threads draw the workgroup's 32 x C normals into LDS right after issuing the chain's loads.  Measured slower than the separate
// update launch (the kernel 18.7 -> 45.8 us; 382.8 vs 389.8 shapes/s): Box-Muller with precise logf / cosf and the update's scattered
// stores of 1632 elements per workgroup cost more than the 71-workgroup launch they replace (round 4's head_update_kernel: -1.4 %).
#include "gemm_common.h"
#include "ddpm_update.h"

namespace {

typedef SlidePointChainArgs ChainArgs;  // include/slide_engine.h

template <int KZMAX, int K0MAX, bool WIDE>
__global__ __launch_bounds__(256, 1) void point_chain_kernel(ChainArgs a) {
#pragma clang fp contract(off)
  using T = _Float16;
  constexpr int LDZ = KZMAX + 8, LDH = 128 + 8, LDX = K0MAX + 8;
  __shared__ __attribute__((aligned(16))) T zs[32 * LDZ];
  __shared__ __attribute__((aligned(16))) T hs[32 * LDH];
  __shared__ __attribute__((aligned(16))) T xs[32 * LDX];
  __shared__ __attribute__((aligned(16))) T hs_lo[WIDE ? 32 * LDH : 8];  // WIDE: the 2^11-scaled low planes of hs / xs
  __shared__ __attribute__((aligned(16))) T xs_lo[WIDE ? 32 * LDX : 8];
  // vectors: [b1 | g1 | be1 | b_res][128], tvec [2 samples][128], [b2 | g2 | be2][128], cvec [2 samples][128], [b0 | g0 | be0][128], b_out [64]
  constexpr int NVL = 512 + 256 + 384 + 256 + 384 + 64;
  __shared__ __attribute__((aligned(16))) float vl[NVL];
  float *const vz_l = vl, *const tv_l = vl + 512, *const v2_l = vl + 768, *const cv_l = vl + 1152, *const v0_l = vl + 1408,
        *const bo_l = vl + 1792;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
  const int row0 = blockIdx.x * 32;
  const int nkz = a.kz >> 4, nk0 = a.k0 >> 4;
  // ---- every global read of the chain, issued together
  f16x8 wz[2][KZMAX / 16], w2[8], w0[K0MAX / 16], w1[8];
  f16x8 wzl[2][WIDE ? KZMAX / 16 : 1], w2l[WIDE ? 8 : 1];  // WIDE: the low fragments of layers 1 and 2 (the head's follow after layer 1)
  if constexpr (WIDE) {
    const GLOBAL_AS T *wp = gptr<const T>((uint64_t)a.Wz_lo) + (size_t)(wave * 32 + col) * a.kz + half * 8;
#pragma unroll
    for (int s2 = 0; s2 < KZMAX / 16; ++s2)
      if (s2 < nkz) {
        wzl[0][s2] = *(const GLOBAL_AS f16x8 *)(wp + s2 * 16);
        wzl[1][s2] = *(const GLOBAL_AS f16x8 *)(wp + (size_t)128 * a.kz + s2 * 16);
      }
    const GLOBAL_AS T *w2p = gptr<const T>((uint64_t)a.W2_lo) + (size_t)(wave * 32 + col) * 128 + half * 8;
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) w2l[s2] = *(const GLOBAL_AS f16x8 *)(w2p + s2 * 16);
  }
  {
    const GLOBAL_AS T *wp = gptr<const T>((uint64_t)a.Wz) + (size_t)(wave * 32 + col) * a.kz + half * 8;
#pragma unroll
    for (int s2 = 0; s2 < KZMAX / 16; ++s2)
      if (s2 < nkz) {
        wz[0][s2] = *(const GLOBAL_AS f16x8 *)(wp + s2 * 16);                          // first_mlp.0, channel block `wave`
        wz[1][s2] = *(const GLOBAL_AS f16x8 *)(wp + (size_t)128 * a.kz + s2 * 16);     // res_connect, same channels
      }
    const GLOBAL_AS T *w2p = gptr<const T>((uint64_t)a.W2) + (size_t)(wave * 32 + col) * 128 + half * 8;
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) w2[s2] = *(const GLOBAL_AS f16x8 *)(w2p + s2 * 16);
    const GLOBAL_AS T *w0p = gptr<const T>((uint64_t)a.W0) + (size_t)(wave * 32 + col) * a.k0 + half * 8;
#pragma unroll
    for (int s2 = 0; s2 < K0MAX / 16; ++s2)
      if (s2 < nk0) w0[s2] = *(const GLOBAL_AS f16x8 *)(w0p + s2 * 16);
    if (wave < a.n1c) {
      const GLOBAL_AS T *w1p = gptr<const T>((uint64_t)a.W1) + (size_t)(wave * 32 + col) * 128 + half * 8;
#pragma unroll
      for (int s2 = 0; s2 < 8; ++s2) w1[s2] = *(const GLOBAL_AS f16x8 *)(w1p + s2 * 16);
    }
  }
  {
    const int ppr = a.kz >> 3;
    for (int i = tid; i < 32 * ppr; i += 256) {
      const int r = i / ppr, pc = i - r * ppr;
      int grow = row0 + r;
      grow = grow < a.rows ? grow : a.rows - 1;
      *reinterpret_cast<u32x4 *>(zs + r * LDZ + pc * 8) = *(const GLOBAL_AS u32x4 *)(gptr<const T>((uint64_t)a.Z) + (size_t)grow * a.z_ld + pc * 8);
    }
    // the head's input beyond the block's 128 output channels: [xyz | zero pad], written once per chain by the point preparation
    const int ppx = (a.k0 - 128) >> 3;
    for (int i = tid; i < 32 * ppx; i += 256) {
      const int r = i / ppx, pc = i - r * ppx;
      int grow = row0 + r;
      grow = grow < a.rows ? grow : a.rows - 1;
      *reinterpret_cast<u32x4 *>(xs + r * LDX + 128 + pc * 8) =
          *(const GLOBAL_AS u32x4 *)(gptr<const T>((uint64_t)a.X) + (size_t)grow * a.x_ld + 128 + pc * 8);
      if constexpr (WIDE) *reinterpret_cast<u32x4 *>(xs_lo + r * LDX + 128 + pc * 8) = u32x4{0, 0, 0, 0};  // (fp16 as stored: exact)
    }
    const int nsm = a.rows >> 4, smp0 = row0 >> 4;
    for (int i = tid; i < NVL; i += 256) {
      float v = 0.f;
      if (i < 512) v = a.vz[i];
      else if (i < 768) {  // row t of the per-timestep table (t_bs == 0) or the sample's own t-embedding row (per-sample timesteps)
        int sm = smp0 + ((i - 512) >> 7);
        sm = sm < nsm ? sm : nsm - 1;
        if (a.tvec) v = a.tvec[(a.t_idx ? (size_t)a.t_idx[0] * a.t_stride : 0) + (size_t)sm * a.t_bs + ((i - 512) & 127)];
      } else if (i < 1152) v = a.v2[i - 768];
      else if (i < 1408) {
        int sm = smp0 + ((i - 1152) >> 7);
        sm = sm < nsm ? sm : nsm - 1;
        if (a.cvec) v = a.cvec[(size_t)sm * a.c_bs + ((i - 1152) & 127)];
      } else if (i < 1792) v = a.v0[i - 1408];
      else if (i - 1792 < a.n1c * 32) v = a.b1[i - 1792];
      vl[i] = v;
    }
  }
  // the update's noise, drawn while the loads above are in flight: z[row][channel] of the workgroup's rows (t > 0, channels past the key points)
  __shared__ float zl[32 * 64];
  int ut = 0, ustep = 0;
  uint32_t unonce = 0, ueoff = 0;
  if (a.fuse_update) {
    const SlideHeadArgs &u = a.upd;
    ut = u.t_dev[0]; ustep = u.t_dev[1];
    unonce = (uint32_t)u.t_dev[3];
    ueoff = (uint32_t)u.t_dev[4] * (uint32_t)(16 * u.C);
    if (ut > 0)
      for (int i = tid; i < 32 * u.C; i += 256) {
        const int r = i / u.C, c = i - r * u.C, p = row0 + r;
        if (c < u.kdim || p >= a.rows) continue;
        const int e = p * u.C + c;
        zl[r * 64 + c] = u.noise ? u.noise[(size_t)ustep * a.rows * u.C + e]
                                 : philox_normal(u.seed_lo, u.seed_hi, (uint32_t)ustep, (uint32_t)e + ueoff, unonce);
      }
  }
  __syncthreads();
  // lane's 16 channels of block `wave`: c(r) = wave * 32 + (r & 3) + 8 (r >> 2) + 4 half; quads q = r >> 2 are GroupNorm groups
  auto vec16 = [&](const float *base, float (&o)[16]) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 t4 = *reinterpret_cast<const float4 *>(base + wave * 32 + 8 * q + 4 * half);
      o[4 * q] = t4.x; o[4 * q + 1] = t4.y; o[4 * q + 2] = t4.z; o[4 * q + 3] = t4.w;
    }
  };
  // y = relu(GroupNorm(acc + bias)) over quads of channels x the sample's 16 rows (= the 16 lanes of the lane's row group)
  auto gn_relu = [&](const f32x16 &acc, const float *vec3, float (&y)[16]) __attribute__((always_inline)) {
    float bia[16], gam[16], bet[16];
    vec16(vec3, bia); vec16(vec3 + 128, gam); vec16(vec3 + 256, bet);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v[4], s = 0.f, ss = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = acc[4 * q + j] + bia[4 * q + j];
        s += v[j];
        ss = fmaf(v[j], v[j], ss);
      }
      s = lane_group_sum<16>(s);
      ss = lane_group_sum<16>(ss);
      const float mean = s * (1.0f / 64.0f);
      const float var = fmaxf(ss * (1.0f / 64.0f) - mean * mean, 0.f);
      const float rstd = __builtin_amdgcn_rsqf(var + GN_EPS);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float g = gam[4 * q + j] * rstd;
        y[4 * q + j] = fmaxf(fmaf(v[j], g, bet[4 * q + j] - mean * g), 0.f);
      }
    }
  };
  auto put16 = [&](T *dst_row, const float (&y)[16]) __attribute__((always_inline)) {  // the lane's row, its 16 channels, as fp16
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f16x4 h;
#pragma unroll
      for (int j = 0; j < 4; ++j) h[j] = (T)y[4 * q + j];
      *reinterpret_cast<f16x4 *>(dst_row + wave * 32 + 8 * q + 4 * half) = h;
    }
  };
  // WIDE: the lane's row as two fp16 planes, y = hi + 2^-11 lo'
  auto put16w = [&](T *hi_row, T *lo_row, const float (&y)[16]) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f16x4 h, l;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        h[j] = (T)y[4 * q + j];
        l[j] = (T)((y[4 * q + j] - (float)h[j]) * 2048.0f);
      }
      *reinterpret_cast<f16x4 *>(hi_row + wave * 32 + 8 * q + 4 * half) = h;
      *reinterpret_cast<f16x4 *>(lo_row + wave * 32 + 8 * q + 4 * half) = l;
    }
  };
  auto zero16 = [](f32x16 &v) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = 0.f;
  };
  auto fold = [](f32x16 &hi, const f32x16 &lo) __attribute__((always_inline)) {  // hi += 2^-11 lo
#pragma unroll
    for (int r = 0; r < 16; ++r) hi[r] = fmaf(lo[r], 1.0f / 2048.0f, hi[r]);
  };
  // ---- layer 1: first_mlp.0 -> h = relu(GN(.)) + t-embedding row;  res_connect -> r (raw, stays in registers)
  f32x16 ah, ar;
  zero16(ah); zero16(ar);
  {
    f32x16 al, arl;
    if constexpr (WIDE) { zero16(al); zero16(arl); }
#pragma unroll
    for (int s2 = 0; s2 < KZMAX / 16; ++s2)
      if (s2 < nkz) {
        const f16x8 xb = *reinterpret_cast<const f16x8 *>(zs + col * LDZ + s2 * 16 + half * 8);
        ah = __builtin_amdgcn_mfma_f32_32x32x16_f16(wz[0][s2], xb, ah, 0, 0, 0);
        ar = __builtin_amdgcn_mfma_f32_32x32x16_f16(wz[1][s2], xb, ar, 0, 0, 0);
        if constexpr (WIDE) {  // (the input rows are fp16 as stored: their low plane is zero)
          al = __builtin_amdgcn_mfma_f32_32x32x16_f16(wzl[0][s2], xb, al, 0, 0, 0);
          arl = __builtin_amdgcn_mfma_f32_32x32x16_f16(wzl[1][s2], xb, arl, 0, 0, 0);
        }
      }
    if constexpr (WIDE) { fold(ah, al); fold(ar, arl); }
  }
  // WIDE: the head's low fragments, loaded now (the registers of layer 1's weights are free); they land under layer 2
  f16x8 w0l[WIDE ? K0MAX / 16 : 1], w1l[WIDE ? 8 : 1];
  if constexpr (WIDE) {
    const GLOBAL_AS T *w0p = gptr<const T>((uint64_t)a.W0_lo) + (size_t)(wave * 32 + col) * a.k0 + half * 8;
#pragma unroll
    for (int s2 = 0; s2 < K0MAX / 16; ++s2)
      if (s2 < nk0) w0l[s2] = *(const GLOBAL_AS f16x8 *)(w0p + s2 * 16);
    if (wave < a.n1c) {
      const GLOBAL_AS T *w1p = gptr<const T>((uint64_t)a.W1_lo) + (size_t)(wave * 32 + col) * 128 + half * 8;
#pragma unroll
      for (int s2 = 0; s2 < 8; ++s2) w1l[s2] = *(const GLOBAL_AS f16x8 *)(w1p + s2 * 16);
    }
  }
  float res[16];
  {
    float y[16], tv[16], br[16];
    gn_relu(ah, vz_l, y);
    vec16(tv_l + (col >> 4) * 128, tv); vec16(vz_l + 384, br);
#pragma unroll
    for (int r = 0; r < 16; ++r) { y[r] += tv[r]; res[r] = ar[r] + br[r]; }
    if constexpr (WIDE) put16w(hs + col * LDH, hs_lo + col * LDH, y);
    else put16(hs + col * LDH, y);
  }
  __syncthreads();
  // ---- layer 2: second_mlp.0 -> out = relu(GN(.)) + class-embedding row + r
  {
    f32x16 a2, a2l;
    zero16(a2);
    if constexpr (WIDE) zero16(a2l);
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) {
      const f16x8 hb = *reinterpret_cast<const f16x8 *>(hs + col * LDH + s2 * 16 + half * 8);
      a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2[s2], hb, a2, 0, 0, 0);
      if constexpr (WIDE) {
        const f16x8 hl = *reinterpret_cast<const f16x8 *>(hs_lo + col * LDH + s2 * 16 + half * 8);
        a2l = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2[s2], hl, a2l, 0, 0, 0);
        a2l = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2l[s2], hb, a2l, 0, 0, 0);
      }
    }
    if constexpr (WIDE) fold(a2, a2l);
    float y[16], cv[16];
    gn_relu(a2, v2_l, y);
    vec16(cv_l + (col >> 4) * 128, cv);
#pragma unroll
    for (int r = 0; r < 16; ++r) y[r] = y[r] + cv[r] + res[r];
    if constexpr (WIDE) put16w(xs + col * LDX, xs_lo + col * LDX, y);
    else put16(xs + col * LDX, y);
    if (row0 + col < a.rows) put16(reinterpret_cast<T *>(a.X) + (size_t)(row0 + col) * a.x_ld, y);
  }
  __syncthreads();
  // ---- head layer 1: hh = relu(GN(W0 . [out | xyz] + b0))
  {
    f32x16 a3, a3l;
    zero16(a3);
    if constexpr (WIDE) zero16(a3l);
#pragma unroll
    for (int s2 = 0; s2 < K0MAX / 16; ++s2)
      if (s2 < nk0) {
        const f16x8 xb = *reinterpret_cast<const f16x8 *>(xs + col * LDX + s2 * 16 + half * 8);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0[s2], xb, a3, 0, 0, 0);
        if constexpr (WIDE) {
          const f16x8 xl = *reinterpret_cast<const f16x8 *>(xs_lo + col * LDX + s2 * 16 + half * 8);
          a3l = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0[s2], xl, a3l, 0, 0, 0);
          a3l = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0l[s2], xb, a3l, 0, 0, 0);
        }
      }
    if constexpr (WIDE) fold(a3, a3l);
    float y[16];
    gn_relu(a3, v0_l, y);
    // (every wave is past its layer-2 reads of hs: the barrier above)
    if constexpr (WIDE) put16w(hs + col * LDH, hs_lo + col * LDH, y);
    else put16(hs + col * LDH, y);
  }
  __syncthreads();
  // ---- head layer 2: eps = W1 . hh + b1
  if (wave < a.n1c) {
    f32x16 e2, e2l;
    zero16(e2);
    if constexpr (WIDE) zero16(e2l);
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) {
      const f16x8 hb = *reinterpret_cast<const f16x8 *>(hs + col * LDH + s2 * 16 + half * 8);
      e2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[s2], hb, e2, 0, 0, 0);
      if constexpr (WIDE) {
        const f16x8 hl = *reinterpret_cast<const f16x8 *>(hs_lo + col * LDH + s2 * 16 + half * 8);
        e2l = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[s2], hl, e2l, 0, 0, 0);
        e2l = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1l[s2], hb, e2l, 0, 0, 0);
      }
    }
    if constexpr (WIDE) fold(e2, e2l);
    float bo[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 t4 = *reinterpret_cast<const float4 *>(bo_l + wave * 32 + 8 * q + 4 * half);
      bo[4 * q] = t4.x; bo[4 * q + 1] = t4.y; bo[4 * q + 2] = t4.z; bo[4 * q + 3] = t4.w;
    }
    const int p = row0 + col;
    if (p < a.rows) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = wave * 32 + 8 * q + 4 * half;
        if (a.eps && c + 3 < a.eps_ld)
          *reinterpret_cast<float4 *>(a.eps + (size_t)p * a.eps_ld + c) =
              make_float4(e2[4 * q] + bo[4 * q], e2[4 * q + 1] + bo[4 * q + 1], e2[4 * q + 2] + bo[4 * q + 2], e2[4 * q + 3] + bo[4 * q + 3]);
      }
      if (a.fuse_update) {
        const SlideHeadArgs &u = a.upd;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (c >= u.C) continue;
          update_feat_element(p * u.C + c, a.rows, u.C, u.kdim, 0, u.clamp, u.seed_lo, u.seed_hi, u.x, nullptr, u.noise, ut, ustep, unonce,
                              ueoff, u.complete_x0, u.kmask, u.keypoint, u.t0, u.t1, u.t2, u.t3, u.t4, u.feat0, u.ldf, u.half_out,
                              u.copies, u.n_copies, e2[r] + bo[r], zl + col * 64 + c);
        }
      }
    }
  }
  if (a.fuse_update) advance_t_last_block(a.upd.t_dev, ut, ustep);
}

}  // namespace

// SLIDE_OP_POINT_CHAIN: p[0] = HOST pointer to a SlidePointChainArgs block (device pointers inside; kept alive by the plan)
int slide_launch_point_chain(const SlideOp &o, hipStream_t s) {
  const SlidePointChainArgs *h = (const SlidePointChainArgs *)o.p[0];
  if (!h || h->rows <= 0 || h->rows % 16 || h->kz % 32 || h->kz <= 0 || h->kz > 192 || h->z_ld < h->kz || h->z_ld % 8 || h->k0 % 32 ||
      h->k0 <= 128 || h->k0 > 160 || h->x_ld < h->k0 || h->x_ld % 8 || (h->n1c != 1 && h->n1c != 2) || h->eps_ld % 4 ||
      h->eps_ld > 32 * h->n1c || h->t_bs < 0 || !h->Z || !h->Wz || !h->W2 || !h->W0 || !h->W1 || !h->vz || !h->v2 || !h->v0 || !h->b1 || !h->X || (!h->eps && !h->fuse_update))
    return -3;
  if (h->fuse_update && (h->upd.kind != 1 || h->upd.C > 32 * h->n1c || h->upd.C > 64 || !h->upd.x || !h->upd.t_dev || !h->upd.t0 || !h->upd.t1 ||
                         !h->upd.t2 || !h->upd.t3 || !h->upd.t4 || (h->upd.kdim > 0 && !h->upd.keypoint)))
    return -3;
  const bool wide = h->Wz_lo || h->W2_lo || h->W0_lo || h->W1_lo;
  if (wide && !(h->Wz_lo && h->W2_lo && h->W0_lo && h->W1_lo)) return -3;
  if (wide) hipLaunchKernelGGL((point_chain_kernel<192, 160, true>), dim3((h->rows + 31) / 32), dim3(256), 0, s, *h);
  else hipLaunchKernelGGL((point_chain_kernel<192, 160, false>), dim3((h->rows + 31) / 32), dim3(256), 0, s, *h);
  return (int)hipGetLastError();
}
