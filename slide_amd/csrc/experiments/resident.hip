// resident.hip -- LDS-resident denoiser for gfx950 (C-ABI: include/experiments/slide_resident.h).
//
// One workgroup (4 waves, one per SIMD, the whole 512-register file each) owns ONE latent-point set and runs the whole
// PointNet2CloudCondition.forward (pointnet2/models/pointnet2_with_pcld_condition.py:286-489) for it, timestep after
// timestep, without leaving the compute unit:
//   * every activation lives in LDS as a row-major fp16 matrix [row][channel] whose row stride is an ODD multiple of 16
//     bytes (conflict-free ds_read_b128 of MFMA B fragments: lane -> row);
//   * a 1x1 convolution is D[channel][row] = sum_k W[channel][k] X[row][k] on v_mfma_f32_32x32x16_f16; a wave owns one
//     32-channel strip over all (or a power-of-two part of) the sample's rows, so GroupNorm statistics are register
//     sums + one DPP reduction; the weights are used by exactly one wave each and therefore go L2 -> VGPR directly,
//     pre-packed on the host in MFMA A-fragment order (1 KB per wave instruction, fully coalesced), prefetched for the
//     next layer while the current one is in its epilogue;
//   * the grouped input of a set-abstraction / feature-propagation block is never built: its feature channels are B
//     fragments read from the NEIGHBOUR's row of the 16-row point-feature table (per-lane LDS addresses are free);
//   * res_connect is a second accumulation phase INTO the normalised registers, the attention tail keeps scores and
//     values in registers (softmax over a point's neighbours = DPP reductions inside 16- / 8-lane groups).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../../include/experiments/slide_resident.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int MAXF = 16;  // weight fragments (16-deep K steps) a unit may need: phase A + phase B
constexpr float GN_EPS = 1e-5f;

// workgroup barrier for LDS traffic only: plain global loads (the weight prefetch) stay in flight across it
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ float dpp_xor1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ float dpp_xor2(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true)); }
__device__ __forceinline__ float dpp_hmirror(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true)); }
__device__ __forceinline__ float dpp_mirror(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true)); }
__device__ __forceinline__ float swz_xor16(float v) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F)); }

// sum / max over aligned groups of W lanes (8, 16 or 32); every lane of the group gets the result
template <int W>
__device__ __forceinline__ float group_sum(float v) {
  v += dpp_xor1(v);
  v += dpp_xor2(v);
  v += dpp_hmirror(v);
  if (W >= 16) v += dpp_mirror(v);
  if (W >= 32) v += swz_xor16(v);
  return v;
}
template <int W>
__device__ __forceinline__ float group_max(float v) {
  v = fmaxf(v, dpp_xor1(v));
  v = fmaxf(v, dpp_xor2(v));
  v = fmaxf(v, dpp_hmirror(v));
  if (W >= 16) v = fmaxf(v, dpp_mirror(v));
  return v;
}

// Philox4x32-10 + Box-Muller, the same stream as engine.hip's update kernels: counter (element, step, nonce), key seed
__device__ __forceinline__ float philox_normal(uint32_t seed_lo, uint32_t seed_hi, uint32_t step, uint32_t elem, uint32_t nonce) {
  uint32_t c0 = elem, c1 = step, c2 = 0x243F6A88u ^ nonce, c3 = 0x85A308D3u, k0 = seed_lo, k1 = seed_hi;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  const float u1 = ((float)(c0 >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float u2 = ((float)(c1 >> 8) + 0.5f) * (1.0f / 16777216.0f);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

__device__ __forceinline__ float sqdist3(float ax, float ay, float az, float bx, float by, float bz) {
#pragma clang fp contract(off)
  const float dx = ax - bx, dy = ay - by, dz = az - bz;
  return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

// descriptor records are read through the CONSTANT address space: uniform index -> s_load into SGPRs (through the generic
// pointer the compiler issues per-lane vector loads and waits for each field)
#define CONST_AS __attribute__((address_space(4)))
typedef const CONST_AS ROp *OpPtr;
typedef const CONST_AS RStrip *StripPtr;

struct Ctx {
  unsigned char *smem;
  const RArgs *a;
  StripPtr strips;
  const float *trow, *crow;  // this step's t-embedding row, this sample's class-embedding row
  int lane, wave, tid;
};

// ---------------------------------------------------------------------------------------------- weights
// the unit's fragments: phase A then phase B, consumed in that order (static register indexing needs the full unroll)
__device__ __forceinline__ void load_wfrags(const Ctx &c, const CONST_AS RStrip &st, int nf, f16x8 (&wf)[MAXF]) {
  const f16x8 *src = reinterpret_cast<const f16x8 *>(c.a->wpool) + (size_t)st.wfrag * 64 + c.lane;
#pragma unroll
  for (int f = 0; f < MAXF; ++f)
    if (f < nf) wf[f] = __builtin_nontemporal_load(src + f * 64);
}

// ---------------------------------------------------------------------------------------------- MFMA phases
// CODE SIZE IS THE FIRST-ORDER COST HERE: a step runs ~50 ops, each once; straight-line code unrolled over the row blocks
// and K steps (340 KB) streamed through the 64 KB instruction cache at ~1.4 bytes per cycle and the kernel was fetch-bound
// (0.38 ms per step).  So every loop that can be a RUNTIME loop is one; register arrays (accumulator tiles, weight
// fragments) are touched with static indices only inside small `switch` selectors on a wave-uniform index.
constexpr int RBM = 8;  // row blocks of 32 per accumulator tile (256 rows)

__device__ __forceinline__ f16x8 pick_w(const f16x8 (&wf)[MAXF], int f) {
  switch (f) {
    case 0: return wf[0]; case 1: return wf[1]; case 2: return wf[2]; case 3: return wf[3];
    case 4: return wf[4]; case 5: return wf[5]; case 6: return wf[6]; case 7: return wf[7];
    case 8: return wf[8]; case 9: return wf[9]; case 10: return wf[10]; case 11: return wf[11];
    case 12: return wf[12]; case 13: return wf[13]; case 14: return wf[14]; default: return wf[15];
  }
}
__device__ __forceinline__ f32x16 pick_acc(const f32x16 (&acc)[RBM], int rb) {
  switch (rb) {
    case 0: return acc[0]; case 1: return acc[1]; case 2: return acc[2]; case 3: return acc[3];
    case 4: return acc[4]; case 5: return acc[5]; case 6: return acc[6]; default: return acc[7];
  }
}
__device__ __forceinline__ void put_acc(f32x16 (&acc)[RBM], int rb, const f32x16 v) {
  switch (rb) {
    case 0: acc[0] = v; break; case 1: acc[1] = v; break; case 2: acc[2] = v; break; case 3: acc[3] = v; break;
    case 4: acc[4] = v; break; case 5: acc[5] = v; break; case 6: acc[6] = v; break; default: acc[7] = v; break;
  }
}

// B-fragment byte addresses of this lane for the unit's row blocks rb0 .. rb0 + rbu: row -> (gathered table row, own row)
__device__ __forceinline__ void frag_addrs(const Ctx &c, const CONST_AS ROp &op, const CONST_AS RIn &in, int rb0, int rbu,
                                           int (&ga)[RBM], int (&xa)[RBM]) {
  const int half16 = (c.lane >> 5) * 16;
  const unsigned char *knn = c.smem + c.a->knn_off;
#pragma unroll
  for (int rb = 0; rb < RBM; ++rb) {
    const int row = (rb0 + (rb < rbu ? rb : 0)) * 32 + (c.lane & 31);
    xa[rb] = in.x_off + row * in.x_ld * 2 + half16;
    ga[rb] = 0;
    if (in.nks_gat > 0) {
      const int p = row >> op.kshift, j = row & ((1 << op.kshift) - 1);
      const int nb = knn[(p & 15) * 16 + j];
      ga[rb] = in.gat_off + nb * in.gat_ld * 2 + half16;
    }
  }
}

// acc[rb] += W(32 x 16 nf) . X(rb)^T for the unit's rbu row blocks; fragments f0 .. f0 + nf of wf
__device__ __forceinline__ void mfma_phase(const Ctx &c, const CONST_AS RIn &in, const int (&ga)[RBM], const int (&xa)[RBM],
                                           const f16x8 (&wf)[MAXF], int f0, int rbu, f32x16 (&acc)[RBM]) {
  const int ng = in.nks_gat, nf = in.nks_gat + in.nks_x;
#pragma nounroll
  for (int k = 0; k < nf; ++k) {
    const f16x8 w = pick_w(wf, f0 + k);
    const bool g = k < ng;
    const int koff = (g ? k : k - ng) * 32;
#pragma unroll
    for (int rb = 0; rb < RBM; ++rb) {
      if (rb < rbu) {
        const f16x8 xb = *reinterpret_cast<const f16x8 *>(c.smem + (g ? ga[rb] : xa[rb]) + koff);
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, xb, acc[rb], 0, 0, 0);
      }
    }
  }
}

__device__ __forceinline__ void load_vec16(const float *base, int half, float (&v)[16]) {
  // channels of this lane: 8 q + 4 half + i  <->  register 4 q + i
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 t = *reinterpret_cast<const float4 *>(base + 8 * q + 4 * half);
    v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
  }
}
__device__ __forceinline__ void zero16(float (&v)[16]) {
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = 0.f;
}

// GroupNorm scale / shift of this lane's 16 channels from per-channel totals S, Q (already complete over the sample's
// rows), groups of gs (1, 2 or 4) consecutive channels
__device__ __forceinline__ void gn_scale_shift(const float (&S)[16], const float (&Q)[16], int gs, float inv_count,
                                               const float (&gamma)[16], const float (&beta)[16], float (&sc)[16],
                                               float (&sh)[16]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float gsum[4], gsq[4];
    const float s01 = S[4 * q] + S[4 * q + 1], s23 = S[4 * q + 2] + S[4 * q + 3];
    const float q01 = Q[4 * q] + Q[4 * q + 1], q23 = Q[4 * q + 2] + Q[4 * q + 3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      gsum[i] = gs == 4 ? s01 + s23 : (gs == 2 ? (i < 2 ? s01 : s23) : S[4 * q + i]);
      gsq[i] = gs == 4 ? q01 + q23 : (gs == 2 ? (i < 2 ? q01 : q23) : Q[4 * q + i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float mean = gsum[i] * inv_count;
      const float var = fmaxf(gsq[i] * inv_count - mean * mean, 0.f);
      const float rstd = rsqrtf(var + GN_EPS);
      sc[4 * q + i] = gamma[4 * q + i] * rstd;
      sh[4 * q + i] = beta[4 * q + i] - mean * sc[4 * q + i];
    }
  }
}

// per-channel totals over the 32 rows (lanes) of each half and, for units that cover only a part of the rows, over the
// parts through LDS.  `xch`: scratch [strip_local][part][half][16][2] floats.  Contains a workgroup barrier when parts > 1
// (every wave of the workgroup calls it the same number of times: idle waves pass `active` = false).
__device__ __forceinline__ void complete_stats(const Ctx &c, bool active, int parts, int part, int sl, float *xch, float (&S)[16],
                                               float (&Q)[16]) {
  const int half = c.lane >> 5;
  if (active) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      S[r] = group_sum<32>(S[r]);
      Q[r] = group_sum<32>(Q[r]);
    }
  }
  if (parts > 1) {
    if (active && (c.lane & 31) == 0) {
      float *dst = xch + ((sl * parts + part) * 2 + half) * 32;
#pragma unroll
      for (int r = 0; r < 16; ++r) { dst[2 * r] = S[r]; dst[2 * r + 1] = Q[r]; }
    }
    lds_barrier();
    if (active) {
      zero16(S); zero16(Q);
#pragma nounroll
      for (int pp = 0; pp < parts; ++pp) {  // fixed order: every part computes bit-identical totals
        const float4 *src = reinterpret_cast<const float4 *>(xch + ((sl * parts + pp) * 2 + half) * 32);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const float4 t = src[r];
          S[2 * r] += t.x; Q[2 * r] += t.y; S[2 * r + 1] += t.z; Q[2 * r + 1] += t.w;
        }
      }
    }
  }
}

// stores this lane's 16 channels of one row (fp16 quads, or fp32 scalars for the network output)
__device__ __forceinline__ void store_row(const Ctx &c, const CONST_AS RStrip &st, int row, const f32x16 v) {
  const int half = c.lane >> 5;
  if (st.flags & SLIDE_RF_OUT_F32) {
    float *o = reinterpret_cast<float *>(c.smem + st.out_off) + row * st.out_ld + st.out_col;
#pragma nounroll
    for (int q = 0; q < 4; ++q)
      for (int i = 0; i < 4; ++i)
        if (8 * q + 4 * half + i < st.n_store) o[8 * q + 4 * half + i] = v[4 * q + i];
    return;
  }
  _Float16 *o = reinterpret_cast<_Float16 *>(c.smem + st.out_off) + row * st.out_ld + st.out_col + 4 * half;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (8 * q + 4 * half < st.n_store) {
      f16x4 h;
#pragma unroll
      for (int i = 0; i < 4; ++i) h[i] = (_Float16)v[4 * q + i];
      *reinterpret_cast<f16x4 *>(o + 8 * q) = h;
    }
  }
}

// ---------------------------------------------------------------------------------------------- GEMM op
// p[0] = LDS byte offset of the statistics exchange scratch, p[1] = some strip needs statistics
// One branch-free epilogue serves every strip kind: the mode / flag differences are CONSTANTS of the arithmetic
//   s  = max(acc + bias + pre, lo_stat)      statistics input   (lo = 0 where a ReLU precedes the statistics, else -inf)
//   v  = max(acc + bias + pre, lo_keep)      value carried on   (STATS strips keep the raw value)
//   y  = max(v * sc + sh, lo_post) + addvec  (RAW / STATS: sc = 1, sh = 0)
__device__ __forceinline__ void gemm_op(const Ctx c, OpPtr opp) {
  const CONST_AS ROp &op = *opp;
  const int half = c.lane >> 5, col = c.lane & 31;
  const int n_units = op.n_strips * op.parts;
  const int nfa = op.a.nks_gat + op.a.nks_x, nfb = op.b.nks_gat + op.b.nks_x;
  const bool r16 = op.rows_log2 == 4;
  const int rbu = r16 ? 1 : (1 << (op.rows_log2 - 5)) / op.parts;
  float *xch = reinterpret_cast<float *>(c.smem + op.p[0]);
  const int rounds = (n_units + 3) >> 2;
  const float NEG = -__builtin_inff();
#pragma nounroll
  for (int rd = 0; rd < rounds; ++rd) {
    const int u = rd * 4 + c.wave;
    const bool active = u < n_units;
    const int sl = active ? u / op.parts : 0, part = active ? u - sl * op.parts : 0;
    const CONST_AS RStrip &st = c.strips[op.strip0 + sl];
    f32x16 acc[RBM];
    f16x8 wf[MAXF];
    float bias[16], gamma[16], beta[16], av[16], S[16], Q[16], sc[16], sh[16];
    int ga[RBM], xa[RBM];
    const int rb0 = part * rbu;
    const bool rvalid = !(r16 && col >= 16);  // rows 16..31 of a 16-row tensor are whatever follows it in LDS
    const int mode = st.mode, flags = st.flags;
    const float lo_stat = (mode == SLIDE_RS_STATS || (flags & SLIDE_RF_PRE_RELU)) ? 0.f : NEG;
    const float lo_keep = (mode != SLIDE_RS_STATS && (flags & SLIDE_RF_PRE_RELU)) ? 0.f : NEG;
    const float lo_post = (mode != SLIDE_RS_STATS && (flags & SLIDE_RF_POST_RELU)) ? 0.f : NEG;
    const bool has_pre = st.preadd_off >= 0;
    const _Float16 *pre_base = reinterpret_cast<const _Float16 *>(c.smem + (has_pre ? st.preadd_off : 0)) + 4 * half;
    zero16(S); zero16(Q); zero16(av);
    if (active) {
      // every global load of the unit is issued here, ahead of the MFMA phase
      load_wfrags(c, st, nfa + nfb, wf);
      load_vec16(c.a->vpool + st.vec_off, half, bias);
      if (mode == SLIDE_RS_NORM) {
        load_vec16(c.a->vpool + st.vec_off + 32, half, gamma);
        load_vec16(c.a->vpool + st.vec_off + 64, half, beta);
      }
      if (st.addvec_kind) load_vec16((st.addvec_kind == 1 ? c.trow : c.crow) + st.addvec_off, half, av);
#pragma unroll
      for (int rb = 0; rb < RBM; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
      frag_addrs(c, op, op.a, rb0, rbu, ga, xa);
      mfma_phase(c, op.a, ga, xa, wf, 0, rbu, acc);
      if (mode != SLIDE_RS_RAW) {
#pragma nounroll
        for (int rb = 0; rb < rbu; ++rb) {
          const f32x16 t = pick_acc(acc, rb);
          const int row = (rb0 + rb) * 32 + col;
          const _Float16 *pp = pre_base + ((row >> op.kshift) & 15) * st.preadd_ld;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f16x4 pv = {0, 0, 0, 0};
            if (has_pre) pv = *reinterpret_cast<const f16x4 *>(pp + 8 * q);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float sv0 = fmaxf(t[4 * q + i] + bias[4 * q + i] + (float)pv[i], lo_stat);
              const float sv = rvalid ? sv0 : 0.f;
              S[4 * q + i] += sv; Q[4 * q + i] += sv * sv;
            }
          }
        }
      }
    }
    if (op.p[1]) complete_stats(c, active && mode != SLIDE_RS_RAW, op.parts, part, sl, xch, S, Q);
    if (active) {
      if (mode == SLIDE_RS_STATS) {
        // per-channel sums of relu(x) for the later concatenated GroupNorm (every part holds the complete totals)
        if (col == 0 && part == 0) {
          float *dst = reinterpret_cast<float *>(c.smem + st.stats_off);
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              dst[2 * (8 * q + 4 * half + i)] = S[4 * q + i];
              dst[2 * (8 * q + 4 * half + i) + 1] = Q[4 * q + i];
            }
        }
      }
      if (mode == SLIDE_RS_NORM) gn_scale_shift(S, Q, st.gs, st.inv_count, gamma, beta, sc, sh);
      else {
#pragma unroll
        for (int r = 0; r < 16; ++r) { sc[r] = 1.f; sh[r] = 0.f; }
      }
    }
    // an output that overwrites an input of this op may only be stored once every wave has read that input
    if (nfb == 0 && (op.flags & SLIDE_RO_BARRIER_BEFORE_STORE)) lds_barrier();
    if (active) {
#pragma nounroll
      for (int rb = 0; rb < rbu; ++rb) {
        const f32x16 t = pick_acc(acc, rb);
        const int row = (rb0 + rb) * 32 + col;
        const _Float16 *pp = pre_base + ((row >> op.kshift) & 15) * st.preadd_ld;
        f32x16 y;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f16x4 pv = {0, 0, 0, 0};
          if (has_pre) pv = *reinterpret_cast<const f16x4 *>(pp + 8 * q);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = 4 * q + i;
            const float v = fmaxf(t[r] + bias[r] + (float)pv[i], lo_keep);
            y[r] = fmaxf(v * sc[r] + sh[r], lo_post) + av[r];
          }
        }
        if (nfb > 0) put_acc(acc, rb, y);
        else if (rvalid) store_row(c, st, row, y);
      }
      if (nfb > 0) {  // second accumulation phase into the finished registers (res_connect)
        frag_addrs(c, op, op.b, rb0, rbu, ga, xa);
        mfma_phase(c, op.b, ga, xa, wf, nfa, rbu, acc);
      }
    }
    if (nfb > 0) {
      if (op.flags & SLIDE_RO_BARRIER_BEFORE_STORE) lds_barrier();
      if (active) {
        float bias2[16];
        load_vec16(c.a->vpool + st.vec_off + 96, half, bias2);
#pragma nounroll
        for (int rb = 0; rb < rbu; ++rb) {
          f32x16 t = pick_acc(acc, rb);
#pragma unroll
          for (int r = 0; r < 16; ++r) t[r] += bias2[r];
          if (rvalid) store_row(c, st, (rb0 + rb) * 32 + col, t);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- attention tail
// AttentionModule tail (attention.py:86-95): scores = W5 u + b5 (phase a), values = relu(GN(Wv mo + bv)) (phase b),
// weights = softmax over the 2^kshift neighbours of a point, out[point][c] = sum_j weights * values.
// vec: [bias_s | gamma_v | beta_v | bias_v].  p[0] = statistics exchange scratch.  A unit holds two accumulator tiles of
// at most RBM / 2 row blocks (scores: as[0..3], values: as[4..7]).
__device__ __forceinline__ float nb_max(float v, bool k16) {
  v = fmaxf(v, dpp_xor1(v)); v = fmaxf(v, dpp_xor2(v)); v = fmaxf(v, dpp_hmirror(v));
  return k16 ? fmaxf(v, dpp_mirror(v)) : v;
}
__device__ __forceinline__ float nb_sum(float v, bool k16) {
  v += dpp_xor1(v); v += dpp_xor2(v); v += dpp_hmirror(v);
  return k16 ? v + dpp_mirror(v) : v;
}

__device__ __forceinline__ void tail_op(const Ctx c, OpPtr opp) {
  const CONST_AS ROp &op = *opp;
  constexpr int HB = RBM / 2;
  const int half = c.lane >> 5, col = c.lane & 31;
  const int n_units = op.n_strips * op.parts;
  const int nfa = op.a.nks_gat + op.a.nks_x, nfb = op.b.nks_gat + op.b.nks_x;
  const int rbu = (1 << (op.rows_log2 - 5)) / op.parts;  // <= 4
  const bool k16 = op.kshift == 4;
  const int KN = 1 << op.kshift;
  float *xch = reinterpret_cast<float *>(c.smem + op.p[0]);
  const int rounds = (n_units + 3) >> 2;
#pragma nounroll
  for (int rd = 0; rd < rounds; ++rd) {
    const int u = rd * 4 + c.wave;
    const bool active = u < n_units;
    const int sl = active ? u / op.parts : 0, part = active ? u - sl * op.parts : 0;
    const CONST_AS RStrip &st = c.strips[op.strip0 + sl];
    f32x16 acc[RBM];
    f16x8 wf[MAXF];
    float bs[16], bv[16], gamma[16], beta[16], S[16], Q[16], sc[16], sh[16];
    int ga[RBM], xa[RBM];
    const int rb0 = part * rbu;
    zero16(S); zero16(Q);
    if (active) {
      load_wfrags(c, st, nfa + nfb, wf);
      load_vec16(c.a->vpool + st.vec_off, half, bs);
      load_vec16(c.a->vpool + st.vec_off + 32, half, gamma);
      load_vec16(c.a->vpool + st.vec_off + 64, half, beta);
      load_vec16(c.a->vpool + st.vec_off + 96, half, bv);
#pragma unroll
      for (int rb = 0; rb < RBM; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
      frag_addrs(c, op, op.a, rb0, rbu, ga, xa);
      mfma_phase(c, op.a, ga, xa, wf, 0, rbu, acc);
      // values into the upper half of the tile: shift the address arrays so that tile block HB + rb sees row block rb
      int gb[RBM], xb[RBM];
      frag_addrs(c, op, op.b, rb0, rbu, ga, xa);
#pragma unroll
      for (int rb = 0; rb < RBM; ++rb) { gb[rb] = ga[rb & (HB - 1)]; xb[rb] = xa[rb & (HB - 1)]; }
      {
        const int ng = op.b.nks_gat;
#pragma nounroll
        for (int k = 0; k < nfb; ++k) {
          const f16x8 w = pick_w(wf, nfa + k);
          const bool g = k < ng;
          const int koff = (g ? k : k - ng) * 32;
#pragma unroll
          for (int rb = HB; rb < RBM; ++rb) {
            if (rb - HB < rbu) {
              const f16x8 xv = *reinterpret_cast<const f16x8 *>(c.smem + (g ? gb[rb] : xb[rb]) + koff);
              acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, xv, acc[rb], 0, 0, 0);
            }
          }
        }
      }
#pragma nounroll
      for (int rb = 0; rb < rbu; ++rb) {
        const f32x16 t = pick_acc(acc, HB + rb);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = t[r] + bv[r];
          S[r] += v; Q[r] += v * v;
        }
      }
    }
    complete_stats(c, active, op.parts, part, sl, xch, S, Q);
    if (active) {
      gn_scale_shift(S, Q, st.gs, st.inv_count, gamma, beta, sc, sh);
#pragma nounroll
      for (int rb = 0; rb < rbu; ++rb) {
        const f32x16 ts = pick_acc(acc, rb), tv = pick_acc(acc, HB + rb);
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = fmaxf((tv[r] + bv[r]) * sc[r] + sh[r], 0.f);
          const float s = ts[r] + bs[r];
          const float e = __expf(s - nb_max(s, k16));
          o[r] = nb_sum(e * v, k16) / nb_sum(e, k16);
        }
        if ((col & (KN - 1)) == 0) store_row(c, st, ((rb0 + rb) * 32 + col) >> op.kshift, o);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- small ops
// PREP: p[0] feature-table offset, p[1] its ld, p[2] its padded width.  features = [x[:, 3:], xyz]
// (attach_position_to_input_feature, pointnet2_with_pcld_condition.py:332-334); neighbour table sorted by
// (distance, index) like knn_points / the engine's prep_points_kernel.
__device__ __forceinline__ void prep_op(const Ctx c, OpPtr opp) {
  const CONST_AS ROp &op = *opp;
  const RArgs &a = *c.a;
  const float *xs = reinterpret_cast<const float *>(c.smem + a.xstate_off);
  float *xyz = reinterpret_cast<float *>(c.smem + a.xyz_off);
  unsigned char *knn = c.smem + a.knn_off;
  float *kd2 = reinterpret_cast<float *>(c.smem + a.kd2_off);
  _Float16 *ft = reinterpret_cast<_Float16 *>(c.smem + op.p[0]);
  const int cx = a.cx, nf = cx - 3, tid = c.tid;
  if (tid < 48) xyz[tid] = xs[(tid / 3) * cx + tid % 3];
  for (int e = tid; e < 16 * op.p[2]; e += 256) {
    const int p = e / op.p[2], ch = e - p * op.p[2];
    float v = 0.f;
    if (ch < nf) v = xs[p * cx + 3 + ch];
    else if (ch < cx) v = xs[p * cx + (ch - nf)];
    ft[p * op.p[1] + ch] = (_Float16)v;
  }
  const int i = tid >> 4, j = tid & 15;
  const float d = sqdist3(xs[i * cx], xs[i * cx + 1], xs[i * cx + 2], xs[j * cx], xs[j * cx + 1], xs[j * cx + 2]);
  // rank of j among the 16 candidates of point i: ties -> lower index (16 lanes of one DPP row hold one point's distances)
  int rank = 0;
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) {
    const float o = __shfl(d, (c.lane & 48) + jj, 64);
    rank += (o < d || (o == d && jj < j)) ? 1 : 0;
  }
  knn[i * 16 + rank] = (unsigned char)j;
  kd2[i * 16 + rank] = d;
}

// ASSEMBLE: coordinate channels of the grouped input, row (p, k):  p[0] out offset, p[1] out ld, p[2] = 1 for the
// group_knn layout, p[3] feature-table offset, p[4] its ld, p[5] feature channels copied in front (small C only)
//   SA (pointnet2_utils.py:397-408): [feat(C) | xyz[nbr]-xyz[p] | xyz[nbr] | xyz[p]]
//   FP (pointnet2_utils.py:506-523): [feat(C) | d2 | w | xyz[nbr] | xyz[nbr]-xyz[p] | xyz[p]], w from squared distances
__device__ __forceinline__ void assemble_op(const Ctx c, OpPtr opp) {
#pragma clang fp contract(off)
  const CONST_AS ROp &op = *opp;
  const RArgs &a = *c.a;
  const float *xyz = reinterpret_cast<const float *>(c.smem + a.xyz_off);
  const unsigned char *knn = c.smem + a.knn_off;
  const float *kd2 = reinterpret_cast<const float *>(c.smem + a.kd2_off);
  _Float16 *out = reinterpret_cast<_Float16 *>(c.smem + op.p[0]);
  const _Float16 *ft = reinterpret_cast<const _Float16 *>(c.smem + op.p[3]);
  const int K = 1 << op.kshift, rows = 16 * K, nfc = op.p[5];
  for (int row = c.tid; row < rows; row += 256) {
    const int p = row >> op.kshift, k = row & (K - 1);
    const int nb = knn[p * 16 + k];
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = 0.f;
    int o = 0;
    for (int i = 0; i < nfc; ++i) v[o++] = (float)ft[nb * op.p[4] + i];
    if (op.p[2]) {
      float norm = 0.f;
      for (int kk = 0; kk < K; ++kk) norm += 1.0f / (kd2[p * 16 + kk] + 1e-8f);
      v[o++] = kd2[p * 16 + k];
      v[o++] = (1.0f / (kd2[p * 16 + k] + 1e-8f)) / norm;
      for (int d = 0; d < 3; ++d) v[o++] = xyz[nb * 3 + d];
      for (int d = 0; d < 3; ++d) v[o++] = xyz[nb * 3 + d] - xyz[p * 3 + d];
      for (int d = 0; d < 3; ++d) v[o++] = xyz[p * 3 + d];
    } else {
      for (int d = 0; d < 3; ++d) v[o++] = xyz[nb * 3 + d] - xyz[p * 3 + d];
      for (int d = 0; d < 3; ++d) v[o++] = xyz[nb * 3 + d];
      for (int d = 0; d < 3; ++d) v[o++] = xyz[p * 3 + d];
    }
    f16x8 h0, h1;
#pragma unroll
    for (int i = 0; i < 8; ++i) { h0[i] = (_Float16)v[i]; h1[i] = (_Float16)v[8 + i]; }
    *reinterpret_cast<f16x8 *>(out + row * op.p[1]) = h0;
    *reinterpret_cast<f16x8 *>(out + row * op.p[1] + 8) = h1;
  }
}

// FINALIZE: GroupNorm over the virtual concatenation [query (C1, per point) | keys (C2, per neighbour)]
// (attention.py weight_conv.1; groups may straddle the two producers).
//   p[0] query sums [C1p][2] (x f[1] = neighbours per point)   p[1] key sums [parts][C2p][2]   p[2] C1  p[3] C2
//   p[4] = vpool offset of [gamma (C1+C2) | beta (C1+C2)] in concatenation order   p[5] query scale/shift [C1p][2] out
//   p[6] key scale/shift [C2p][2] out   p[7] = C1p | C2p << 10 | key parts << 20    f[0] = 1 / (gs * rows)
__device__ __forceinline__ void finalize_op(const Ctx c, OpPtr opp) {
  const CONST_AS ROp &op = *opp;
  const RArgs &a = *c.a;
  const float *qs = reinterpret_cast<const float *>(c.smem + op.p[0]);
  const float *ks = reinterpret_cast<const float *>(c.smem + op.p[1]);
  float *qo = reinterpret_cast<float *>(c.smem + op.p[5]);
  float *ko = reinterpret_cast<float *>(c.smem + op.p[6]);
  const int C1 = op.p[2], C2 = op.p[3], Ct = C1 + C2;
  const int C1p = op.p[7] & 1023, C2p = (op.p[7] >> 10) & 1023, kparts = op.p[7] >> 20;
  const int G = Ct < 32 ? Ct : 32, n_norm = Ct - Ct % G, gs = n_norm / G;
  const float *gamma = a.vpool + op.p[4], *beta = gamma + Ct;
  // pad channels: scale = shift = 0 (their columns stay zero through AFFINE)
  for (int ch = c.tid; ch < C1p; ch += 256)
    if (ch >= C1) { qo[2 * ch] = 0.f; qo[2 * ch + 1] = 0.f; }
  for (int ch = c.tid; ch < C2p; ch += 256)
    if (ch >= C2) { ko[2 * ch] = 0.f; ko[2 * ch + 1] = 0.f; }
  // pass-through channels beyond n_norm: y = relu(x)
  for (int ch = n_norm + c.tid; ch < Ct; ch += 256) {
    float *o = ch < C1 ? qo + 2 * ch : ko + 2 * (ch - C1);
    o[0] = 1.f; o[1] = 0.f;
  }
  if (c.tid < G) {
    const int g = c.tid;
    float S = 0.f, Q = 0.f;
    for (int i = 0; i < gs; ++i) {
      const int ch = g * gs + i;
      if (ch < C1) { S += qs[2 * ch] * op.f[1]; Q += qs[2 * ch + 1] * op.f[1]; }
      else
        for (int pp = 0; pp < kparts; ++pp) { S += ks[(pp * C2p + (ch - C1)) * 2]; Q += ks[(pp * C2p + (ch - C1)) * 2 + 1]; }
    }
    const float mean = S * op.f[0];
    const float var = fmaxf(Q * op.f[0] - mean * mean, 0.f);
    const float rstd = rsqrtf(var + GN_EPS);
    for (int i = 0; i < gs; ++i) {
      const int ch = g * gs + i;
      const float sc = gamma[ch] * rstd, sh = beta[ch] - mean * sc;
      float *o = ch < C1 ? qo + 2 * ch : ko + 2 * (ch - C1);
      o[0] = sc; o[1] = sh;
    }
  }
}

// AFFINE: x <- relu(x) * scale[ch] + shift[ch] in place.  p[0] buffer, p[1] ld, p[2] rows, p[3] padded channels (x8),
// p[4] scale/shift [ch][2]
__device__ __forceinline__ void affine_op(const Ctx c, OpPtr opp) {
  const CONST_AS ROp &op = *opp;
  _Float16 *x = reinterpret_cast<_Float16 *>(c.smem + op.p[0]);
  const float *ss = reinterpret_cast<const float *>(c.smem + op.p[4]);
  const int nch8 = op.p[3] >> 3, n = op.p[2] * nch8;
  for (int e = c.tid; e < n; e += 256) {
    const int row = e / nch8, c8 = (e - row * nch8) * 8;
    f16x8 h = *reinterpret_cast<f16x8 *>(x + row * op.p[1] + c8);
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = (_Float16)(fmaxf((float)h[i], 0.f) * ss[2 * (c8 + i)] + ss[2 * (c8 + i) + 1]);
    *reinterpret_cast<f16x8 *>(x + row * op.p[1] + c8) = h;
  }
}

// ZFILL: columns of a 16-row concatenation buffer that no GEMM writes: dst[:, col0 : col0+n] = src[:, 0:n] (fp16 table),
// then xyz (3), then zeros up to p[6].   p[0] dst, p[1] dst ld, p[2] col0, p[3] src (-1: none), p[4] src ld, p[5] n,
// p[6] end column (padded K)
__device__ __forceinline__ void zfill_op(const Ctx c, OpPtr opp) {
  const CONST_AS ROp &op = *opp;
  const float *xyz = reinterpret_cast<const float *>(c.smem + c.a->xyz_off);
  _Float16 *dst = reinterpret_cast<_Float16 *>(c.smem + op.p[0]);
  const _Float16 *src = reinterpret_cast<const _Float16 *>(c.smem + (op.p[3] >= 0 ? op.p[3] : 0));
  const int w = op.p[6] - op.p[2], n = op.p[5];
  for (int e = c.tid; e < 16 * w; e += 256) {
    const int row = e / w, j = e - row * w;
    _Float16 v = (_Float16)0.f;
    if (j < n) v = src[row * op.p[4] + j];
    else if (j < n + 3) v = (_Float16)xyz[row * 3 + (j - n)];
    dst[row * op.p[1] + op.p[2] + j] = v;
  }
}

// ---------------------------------------------------------------------------------------------- the kernel
__global__ __launch_bounds__(256, 1) void resident_kernel(RArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Ctx c;
  c.smem = smem; c.a = &a;
  c.strips = (StripPtr)(uintptr_t)a.strips;
  c.tid = threadIdx.x; c.lane = threadIdx.x & 63; c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x;
  float *xs = reinterpret_cast<float *>(smem + a.xstate_off);
  const int nx = 16 * a.cx;
  for (int e = c.tid; e < nx; e += 256) xs[e] = a.x[(size_t)b * nx + e];
  int t = 0, step = 0;
  uint32_t nonce = 0;
  if (a.t_dev) { t = a.t_dev[0]; step = a.t_dev[1]; nonce = (uint32_t)a.t_dev[3]; }
  c.crow = a.cvec + (size_t)b * a.cvec_ld;
  lds_barrier();
  for (int s = 0; s < a.n_steps; ++s) {
    c.trow = a.tvec + (size_t)(a.per_sample_t ? b : t) * a.tvec_ld;
    if (a.timeline && s == a.n_steps - 1 && b == 0 && c.tid == 0) a.timeline[0] = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < a.n_ops; ++i) {
      const OpPtr opp = (OpPtr)(uintptr_t)a.ops + i;
      const CONST_AS ROp &op = *opp;
      switch (op.type) {
        case SLIDE_R_PREP: prep_op(c, opp); break;
        case SLIDE_R_ASSEMBLE: assemble_op(c, opp); break;
        case SLIDE_R_FINALIZE: finalize_op(c, opp); break;
        case SLIDE_R_AFFINE: affine_op(c, opp); break;
        case SLIDE_R_ZFILL: zfill_op(c, opp); break;
        case SLIDE_R_GEMM: gemm_op(c, opp); break;
        case SLIDE_R_TAIL: tail_op(c, opp); break;
        default: break;
      }
      lds_barrier();
      if (a.timeline && s == a.n_steps - 1 && b == 0 && c.tid == 0) a.timeline[i + 1] = __builtin_amdgcn_s_memtime();
    }
    const float *eps = reinterpret_cast<const float *>(smem + a.eps_off);  // [16][4] fp32, written by the head
    if (a.t_dev) {
#pragma clang fp contract(off)
      // sampling() update (pointnet2/util.py:247-253): x = (x - c_eps[t] eps) / sqrt_alpha[t]; t > 0: x += sigma[t] z
      if (c.tid < nx) {
        const int p = c.tid / a.cx, d = c.tid - p * a.cx;
        float v = (xs[c.tid] - a.c_eps[t] * eps[p * 4 + d]) / a.sqrt_alpha[t];
        if (t > 0) {
          const uint32_t e = (uint32_t)(b * nx + c.tid);
          const float z = a.noise ? a.noise[(size_t)(step + s) * a.B * nx + e]
                                  : philox_normal(a.seed_lo, a.seed_hi, (uint32_t)(step + s), e, nonce);
          v = v + a.sigma[t] * z;
        }
        xs[c.tid] = v;
      }
      --t;
      lds_barrier();
    }
    if (a.eps_out && s == a.n_steps - 1 && c.tid < 16 * a.out_dim)
      a.eps_out[(size_t)b * 16 * a.out_dim + c.tid] = eps[(c.tid / a.out_dim) * 4 + c.tid % a.out_dim];
  }
  for (int e = c.tid; e < nx; e += 256) a.x[(size_t)b * nx + e] = xs[e];
  if (a.dbg) {
    uint32_t *d = reinterpret_cast<uint32_t *>(a.dbg) + (size_t)b * (a.lds_bytes / 4);
    for (int e = c.tid; e < a.lds_bytes / 4; e += 256) d[e] = reinterpret_cast<const uint32_t *>(smem)[e];
  }
  // the last workgroup to finish advances the device-side timestep for the next launch
  if (a.t_dev) {
    __syncthreads();
    if (c.tid == 0) {
      __threadfence();
      if (atomicAdd(&a.t_dev[2], 1) == (int)gridDim.x - 1) {
        a.t_dev[2] = 0;
        a.t_dev[0] = t;
        a.t_dev[1] = step + a.n_steps;
      }
    }
  }
}

}  // namespace

extern "C" {

int slide_resident_run(const RArgs *args, slide_stream_t stream) {
  if (!args || args->B <= 0 || args->n_steps <= 0 || args->n_ops <= 0) return -3;
  if (args->lds_bytes > 160 * 1024 || args->cx * 16 > 256 || args->out_dim > 4) return -8;
  static bool attr_done[64] = {};
  int d = 0;
  (void)hipGetDevice(&d);
  bool &set = attr_done[d >= 0 && d < 64 ? d : 0];
  if (!set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&resident_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    set = true;
  }
  hipLaunchKernelGGL(resident_kernel, dim3(args->B), dim3(256), (size_t)args->lds_bytes, (hipStream_t)stream, *args);
  return (int)hipGetLastError();
}
int slide_sizeof_rop(void) { return (int)sizeof(ROp); }
int slide_sizeof_rstrip(void) { return (int)sizeof(RStrip); }
int slide_sizeof_rargs(void) { return (int)sizeof(RArgs); }

}  // extern "C"
