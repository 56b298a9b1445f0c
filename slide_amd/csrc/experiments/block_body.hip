// block_body.hip -- the whole K-expanded body of an SA / FP block in ONE launch (round 3).
//
// After the pair decomposition (gemm_gx.hip) the 256- / 128-row part of a block is
//   mo = Mlp tail over h1 (second_mlp [-> rest_mlp], pointnet2_modules.py:119-176), h1 generated from the pair tables, + pair residual
//   u  = GN4(relu(W2k . GN(relu(keys)) + P[point] + b))      (weight_conv.2 .. .4, attention.py:70-96; keys generated)
//   out[point] = sum_j softmax_j(W5 . u + b5) * relu(GN(Wv . mo + bv))       (weight_conv.5, feat_out_conv, softmax over the neighbours)
// For blocks whose widths are at most 256 channels (FP1, FP0, SA0 of the feature denoiser) a workgroup that owns ONE sample
// (32 rows per wave: 4 waves for 16 x 8 rows, 8 for 16 x 16) keeps mo and u in REGISTERS as MFMA operand fragments
// (64 + 64 registers at 256 channels): a wave owns all channels of its 32 rows, so after a GroupNorm (statistics across the
// sample's waves through LDS) its accumulators, converted to fp16 and re-paired with v_permlane32_swap, are the operand
// fragments of the next contraction -- the B operand of the next layer (D[channel][row]) and, transposed roles, the A operand
// of the attention tail (D[row][channel]: a lane owns a channel, the softmax over a point's neighbours is a register loop).
// Nothing K-expanded touches memory: inputs are the per-point pair tables, outputs 16 rows per sample.  All weights stream
// through one LDS-DMA ring of 16 KB slots whose (host-built) descriptor list runs ahead across every stage.
#include "../gemm_common.h"

namespace {

struct BodySlot {        // one 16 KB ring slot: sub-images of [rows][64 B] (eight-wave form: kind 0 = 2 x 128 rows, kind 1 = 4 x 64 rows; four-wave form: 1 x 256 rows)
  const void *src;       // element (k chunk kc0, first row) of a chunk-major weight matrix [k/32][n][32]
  int32_t chunk_stride;  // elements between consecutive k chunks (n * 32)
  int32_t nrows;         // valid rows of a sub-image (the rest is clamped: never read)
  int32_t kind;          // 0 | 1
  int32_t nvalid;        // valid sub-images (k chunks)
};

struct BodyArgs {
  const BodySlot *slots;
  int n_slots;
  const void *ta, *tb;        // fp16 pair tables [B*16][t_ld]
  int t_ld, off1, k1, offr, offk, kk;
  const float *vv; int vbs;   // 8-neighbour samples: per-sample (vd | vw) fp32 [b*vbs + {0, vbs/2} + table column]
  const float *rv;            // ... and the res_connect coefficients fp32 [2][n_mo] (unscaled)
  const int *nbr; const float *d2, *w;
  const float *add0; const int *add0_idx; int add0_stride, add0_bs;  // h1 add vector (t-embedding) or NULL
  const float *sc, *sh; int aff_bs;                                  // joint GroupNorm of the keys: [b*aff_bs + k]
  const void *P; int p_ld;                                           // fp16 [B*16][p_ld]: per-point pre-activation of u
  const float *vec1; int n1, gs1; float inv1; const float *add1; int add1_bs;      // (REST) h1 -> h2
  const float *vecm; int n_mo, gsm; float invm; const float *addm; int addm_bs;    // -> mo (+ pair residual)
  const float *vecu; int n_u, gsu, nnu; float invu;                                // keys -> u (ReLU BEFORE the norm)
  const float *vect; int n_out, gsv, nnv; float invv;                              // tail [bias_s | bias_v | gamma | beta][n_out]
  void *out; int out_ld; void *out2; int out2_ld, out2_n;
  int B;
  unsigned long long *dbg;
};

constexpr int SLOT_B = 16384, BB_NST = 3;

__device__ __forceinline__ float hw_sum_hi(float v) {  // 32-lane sum, valid in the half's upper 16 lanes (pure DPP)
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false));
  return v;
}
__device__ __forceinline__ void lds_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ float oth_half(float x) {  // value of lane ^ 32
  uint32_t a = __float_as_uint(x), b = a;
  lane32_swap(a, b);
  return __uint_as_float((threadIdx.x & 32) ? a : b);
}

// NPXL 7: 16 x 8 rows (4 waves), neighbour table + the two per-slot scalars of group_knn; 8: 16 x 16 rows (8 waves), natural order.
// REST: the Mlp has a rest_mlp (h1 -> h2 [NB1 blocks] -> mo) else h1 -> mo.  NBM / NBU: 32-channel blocks of mo / u.
template <int NPXL, bool REST, int NB1, int NBM, int NBU>
__global__ __launch_bounds__(64 << (NPXL - 5), NPXL == 7 ? 1 : 2) void block_body_kernel(BodyArgs a) {
  // (16 x 8-row samples: four waves, one per SIMD -- the workgroup's LDS footprint admits one per CU anyway -- so a wave may use
  //  the whole 512-entry register file; 16 x 16-row samples: eight waves, 256 registers each)
  using T = _Float16;
  constexpr bool FP = NPXL == 7;
  constexpr int NW = 1 << (NPXL - 5), NT = 64 * NW;  // waves, threads
  constexpr int KN = 1 << (NPXL - 4), GPB = 32 / KN, RPG = 16 / GPB;  // neighbours per point, points per 32-row block, regs per point
  constexpr int LPW = 16 / NW;                          // DMA instructions per wave and slot
  // blocks per slab of the D[channel][row] stages / per column block of the tail: the four-wave form has the whole register
  // file per wave and one wave per SIMD, so it takes everything at once (8 MFMAs per 16-deep step cover the LDS reads of the
  // next step, whose count does not depend on the width); the eight-wave form has 256 registers per wave
  constexpr int SLB = FP ? 8 : 4, TCB = FP ? 4 : 2;
  constexpr int CPS0 = 8 / SLB, CPS1 = 8 / TCB;         // 32-deep chunks per ring slot (slab stages / tail)
  constexpr int NR = 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, col = lane & 31;
#ifdef SLIDE_TIMELINE
#define BB_STAMP(k) do { if (a.dbg && tid == 0) a.dbg[(size_t)b * 16 + (k)] = wall_clock64(); } while (0)
#else
#define BB_STAMP(k) do { } while (0)
#endif
  BB_STAMP(0);
  // ---- LDS map
  unsigned char *const ring = smem_raw;
  float *fp = reinterpret_cast<float *>(smem_raw + BB_NST * SLOT_B);
  float *const vec1_l = fp; fp += REST ? 3 * a.n1 : 0;
  float *const vecm_l = fp; fp += 3 * a.n_mo;
  float *const vecu_l = fp; fp += 3 * a.n_u;
  float *const vect_l = fp; fp += 4 * a.n_out;
  float *const add1_l = fp; fp += REST ? a.n1 : 0;
  float *const addm_l = fp; fp += a.n_mo;
  float *const rv_l = fp; fp += FP ? 2 * a.n_mo : 0;
  float *const red = fp; fp += NW * SLB * 2 * 4 * 2;
  float *const gsh = fp; fp += SLB * 2 * 32;
  float *const redt = fp; fp += NW * TCB * 32 * 2;
  T *hp = reinterpret_cast<T *>(fp);
  T *const P_l = hp; hp += 16 * a.n_u;
  T *const add0_l = hp; hp += a.k1;
  T *const sc_l = hp; hp += a.kk;
  T *const sh_l = hp; hp += a.kk;
  T *const vd1_l = hp; hp += FP ? a.k1 : 0;
  T *const vw1_l = hp; hp += FP ? a.k1 : 0;
  T *const vdk_l = hp; hp += FP ? a.kk : 0;
  T *const vwk_l = hp; hp += FP ? a.kk : 0;
  unsigned char *tp = reinterpret_cast<unsigned char *>(hp);
  unsigned char *const t1a = tp; tp += (a.k1 >> 3) * NR * 16;
  unsigned char *const t1b = tp; tp += (a.k1 >> 3) * NR * 16;
  unsigned char *const tra = tp; tp += (a.n_mo >> 3) * NR * 16;
  unsigned char *const trb = tp; tp += (a.n_mo >> 3) * NR * 16;
  unsigned char *const tka = tp; tp += (a.kk >> 3) * NR * 16;
  unsigned char *const tkb = tp; tp += (a.kk >> 3) * NR * 16;

  // ---- pair tables by LDS-DMA: image [piece][16 rows][16 B]; one instruction = 4 pieces x 16 rows
  {
    const int r = lane & 15, pl = lane >> 4;
    const size_t grow = ((size_t)b * 16 + r) * a.t_ld;
    auto stage_tab = [&](const void *tab, int coff, int width, unsigned char *dst, int w0) {
      const int nins = (width >> 3) / 4;
      for (int i = (wave + NW - w0 % NW) % NW; i < nins; i += NW)
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)(reinterpret_cast<const T *>(tab) + grow + coff + (i * 4 + pl) * 8),
                                         (__attribute__((address_space(3))) void *)(dst + i * 1024), 16, 0, 0);
    };
    stage_tab(a.ta, a.off1, a.k1, t1a, 0); stage_tab(a.tb, a.off1, a.k1, t1b, 1);
    stage_tab(a.ta, a.offr, a.n_mo, tra, 2); stage_tab(a.tb, a.offr, a.n_mo, trb, 3);
    stage_tab(a.ta, a.offk, a.kk, tka, 0); stage_tab(a.tb, a.offk, a.kk, tkb, 1);
  }
  // ---- weight ring
  auto issue = [&](int g) __attribute__((always_inline)) {
    const BodySlot sl = a.slots[g];
    unsigned char *dst = ring + (size_t)(g % BB_NST) * SLOT_B;
    const int rps_log = sl.kind ? (TCB == 8 ? 8 : TCB == 4 ? 7 : 6) : (SLB == 8 ? 8 : 7);  // rows per sub-image
#pragma unroll
    for (int j = 0; j < LPW; ++j) {
      const int irow = 16 * (j * NW + wave) + (lane >> 2);  // image row 0 .. 255
      int sub = irow >> rps_log, r = irow & ((1 << rps_log) - 1);
      const int piece = (lane & 3) ^ ((irow >> 2) & 3);
      sub = sub < sl.nvalid ? sub : 0;
      r = r < sl.nrows ? r : sl.nrows - 1;
      __builtin_amdgcn_global_load_lds(
          (const GLOBAL_AS void *)(reinterpret_cast<const T *>(sl.src) + (size_t)sub * sl.chunk_stride + r * 32 + piece * 8),
          (__attribute__((address_space(3))) void *)(dst + (j * NW + wave) * 1024), 16, 0, 0);
    }
  };
  int gslot = 0;  // next slot to consume (uniform)
  // slot g must have landed; the younger ones stay in flight
  auto slot_ready = [&]() __attribute__((always_inline)) {
    const int g = gslot;
    if (g + BB_NST - 2 < a.n_slots) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((BB_NST - 2) * LPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (g + BB_NST - 1 < a.n_slots) issue(g + BB_NST - 1);
    ++gslot;
    return ring + (size_t)(g % BB_NST) * SLOT_B;
  };
#pragma unroll
  for (int s0 = 0; s0 < BB_NST - 1; ++s0)
    if (s0 < a.n_slots) issue(s0);
  // (the DMAs above are in flight while the vectors below make their own round trips)
  // ---- vectors (plain loads, converted on the way)
  if (REST) for (int i = tid; i < 3 * a.n1; i += NT) vec1_l[i] = a.vec1[i];
  for (int i = tid; i < 3 * a.n_mo; i += NT) vecm_l[i] = a.vecm[i];
  for (int i = tid; i < 3 * a.n_u; i += NT) vecu_l[i] = a.vecu[i];
  for (int i = tid; i < 4 * a.n_out; i += NT) vect_l[i] = a.vect[i];
  if (REST) for (int i = tid; i < a.n1; i += NT) add1_l[i] = a.add1 ? a.add1[(size_t)b * a.add1_bs + i] : 0.f;
  for (int i = tid; i < a.n_mo; i += NT) addm_l[i] = a.addm ? a.addm[(size_t)b * a.addm_bs + i] : 0.f;
  if (FP) for (int i = tid; i < 2 * a.n_mo; i += NT) rv_l[i] = a.rv[i];
  {
    const float *addp = a.add0;
    if (addp && a.add0_idx) addp += (size_t)a.add0_idx[0] * a.add0_stride;
    for (int i = tid; i < a.k1; i += NT) {
      add0_l[i] = (T)(addp ? addp[(size_t)b * a.add0_bs + i] : 0.f);
      if (FP) {
        vd1_l[i] = (T)a.vv[(size_t)b * a.vbs + a.off1 + i];
        vw1_l[i] = (T)a.vv[(size_t)b * a.vbs + (a.vbs >> 1) + a.off1 + i];
      }
    }
    for (int i = tid; i < a.kk; i += NT) {
      sc_l[i] = (T)a.sc[(size_t)b * a.aff_bs + i];
      sh_l[i] = (T)a.sh[(size_t)b * a.aff_bs + i];
      if (FP) {
        vdk_l[i] = (T)a.vv[(size_t)b * a.vbs + a.offk + i];
        vwk_l[i] = (T)a.vv[(size_t)b * a.vbs + (a.vbs >> 1) + a.offk + i];
      }
    }
    for (int i = tid; i < 16 * (a.n_u >> 3); i += NT) {  // P rows, 16 bytes per thread
      const int r = i / (a.n_u >> 3), pc = i - r * (a.n_u >> 3);
      *reinterpret_cast<u32x4 *>(P_l + r * a.n_u + pc * 8) =
          *reinterpret_cast<const u32x4 *>(reinterpret_cast<const T *>(a.P) + ((size_t)b * 16 + r) * a.p_ld + pc * 8);
    }
  }
  BB_STAMP(1);

  // ---- this lane's row: point p, neighbour q, per-slot scalars
  const int pxl = wave * 32 + col;
  int p_pt, q_pt;
  f16x2 d2s = {0, 0}, ws = {0, 0};
  if (!FP) { p_pt = pxl >> 4; q_pt = pxl & 15; }
  else {
    p_pt = pxl >> 3;
    const int slot = (b * 16 + p_pt) * 16 + (pxl & 7);
    q_pt = a.nbr[slot];
    const T dh = (T)a.d2[slot], wh = (T)a.w[slot];
    d2s = f16x2{dh, dh}; ws = f16x2{wh, wh};
  }
  const int aoff = (half * NR + q_pt) * 16, boff = (half * NR + p_pt) * 16;  // piece-relative offsets inside a table image
  const int raoff = q_pt * 16, rboff = p_pt * 16;
  int wrow[8], wkey[8];
#pragma unroll
  for (int cb = 0; cb < 8; ++cb) {
    const int trow = cb * 32 + col;
    wrow[cb] = trow * 64; wkey[cb] = (trow >> 2) & 3;
  }

  // generated B fragment of K step (chunk kc, 16-deep step st2) from a table pair: the two table rows are FETCHED one step
  // ahead (GenIn), the fragment is made right before its MFMAs (the per-channel vectors are uniform reads)
  struct GenIn { f16x8 av, bv; int ke; };
  auto gen_fetch = [&](GenIn &o, const unsigned char *ta_l, const unsigned char *tb_l, int kc, int st2) __attribute__((always_inline)) {
    const int pb = (kc * 4 + st2 * 2) * NR * 16;
    o.ke = kc * 32 + st2 * 16 + half * 8;
    o.av = *reinterpret_cast<const f16x8 *>(ta_l + pb + aoff);
    o.bv = *reinterpret_cast<const f16x8 *>(tb_l + pb + boff);
  };
  auto gen_make = [&](auto mode_tag, const GenIn &in) __attribute__((always_inline)) {
    constexpr int MODE = decltype(mode_tag)::value;  // 0: max(.,0) + add0;  1: max(.,0) * sc + sh
    const f16x8 v0 = *reinterpret_cast<const f16x8 *>((MODE ? sc_l : add0_l) + in.ke);
    f16x8 v1, vd, vw;
    if (MODE) v1 = *reinterpret_cast<const f16x8 *>(sh_l + in.ke);
    if (FP) {
      vd = *reinterpret_cast<const f16x8 *>((MODE ? vdk_l : vd1_l) + in.ke);
      vw = *reinterpret_cast<const f16x8 *>((MODE ? vwk_l : vw1_l) + in.ke);
    }
    const f16x2 zero2 = {0, 0};
    f16x8 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f16x2 y = f16x2{in.av[2 * i], in.av[2 * i + 1]} + f16x2{in.bv[2 * i], in.bv[2 * i + 1]};
      if (FP) {
        y = __builtin_elementwise_fma(d2s, f16x2{vd[2 * i], vd[2 * i + 1]}, y);
        y = __builtin_elementwise_fma(ws, f16x2{vw[2 * i], vw[2 * i + 1]}, y);
      }
      y = __builtin_elementwise_max(y, zero2);
      if (MODE) y = __builtin_elementwise_fma(y, f16x2{v0[2 * i], v0[2 * i + 1]}, f16x2{v1[2 * i], v1[2 * i + 1]});
      else y = y + f16x2{v0[2 * i], v0[2 * i + 1]};
      o[2 * i] = y[0]; o[2 * i + 1] = y[1];
    }
    return o;
  };
  // accumulators of a 128-channel slab start from the bias
  auto init_acc = [&](f32x16 (&v)[SLB], const float *bias_l, int nb) __attribute__((always_inline)) {
#pragma unroll
    for (int cb = 0; cb < SLB; ++cb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 bia = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cb < nb) bia = *reinterpret_cast<const float4 *>(bias_l + cb * 32 + 8 * q + 4 * half);
        v[cb][4 * q] = bia.x; v[cb][4 * q + 1] = bia.y; v[cb][4 * q + 2] = bia.z; v[cb][4 * q + 3] = bia.w;
      }
  };
  auto wfrag = [&](const unsigned char *sb, int sub_off, int cb, int st2) __attribute__((always_inline)) {
    return *reinterpret_cast<const f16x8 *>(sb + sub_off + wrow[cb] + (((st2 * 2 + half) ^ wkey[cb]) << 4));
  };
  // one slab (<= 4 blocks) of a D[channel][row] GEMM over nkc 32-deep chunks, B fragments generated from the table pair
  // (ta_l, tb_l) in mode MODE.  Software-pipelined over the 16-deep steps: the weight fragments and table rows of step s + 1
  // are read before the MFMAs of step s; the next ring slot's barrier sits one step early.
  struct SlabStep { f16x8 wf[SLB]; GenIn g; };
  auto run_slab = [&](f32x16 (&acc)[SLB], int nb, int nkc, auto mode_tag, const unsigned char *ta_l, const unsigned char *tb_l) __attribute__((always_inline)) {
    auto ld = [&](SlabStep &o, const unsigned char *sb, int c2, int st2, int kc) __attribute__((always_inline)) {
#pragma unroll
      for (int cb = 0; cb < SLB; ++cb)
        if (cb < nb) o.wf[cb] = wfrag(sb, c2 * (SLB * 2048), cb, st2);
      gen_fetch(o.g, ta_l, tb_l, kc, st2);
    };
    auto mm = [&](const SlabStep &o) __attribute__((always_inline)) {
      const f16x8 bf = gen_make(mode_tag, o.g);
#pragma unroll
      for (int cb = 0; cb < SLB; ++cb)
        if (cb < nb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.wf[cb], bf, acc[cb], 0, 0, 0);
    };
    SlabStep st[2];
    const unsigned char *sb = slot_ready();
    ld(st[0], sb, 0, 0, 0);
    for (int kc0 = 0; kc0 < nkc; kc0 += CPS0) {
#pragma unroll
      for (int t = 0; t < 2 * CPS0; ++t) {  // the slot's 16-deep steps (an even count: the ping-pong parity restarts per slot)
        if (kc0 + (t >> 1) >= nkc) break;   // (odd chunk count: the last slot is half empty)
        const int tn = t + 1;
        if (tn < 2 * CPS0 && kc0 + (tn >> 1) < nkc) ld(st[tn & 1], sb, tn >> 1, tn & 1, kc0 + (tn >> 1));
        else if (kc0 + CPS0 < nkc) {
          sb = slot_ready();
          ld(st[tn & 1], sb, 0, 0, kc0 + CPS0);
        }
        mm(st[t & 1]);
      }
    }
  };
  // the same with the B fragments taken from REGISTERS (bq[2 * kc + st2], fully unrolled: NKC chunks) -- or, ROWS = true,
  // operands swapped: the register fragments are the A side (rows), the weight fragments the B side, D[row][channel], NBW
  // blocks of channels per weight image whose sub-images are SUBB bytes apart and hold SUBS chunks per slot
  auto run_reg = [&](auto nkc_tag, auto nbw_tag, auto rows_tag, f32x16 *acc, int nb, const f16x8 *bq) __attribute__((always_inline)) {
    constexpr int NKC = decltype(nkc_tag)::value, NBW = decltype(nbw_tag)::value;
    constexpr bool ROWS = decltype(rows_tag)::value;
    constexpr int SUBS = ROWS ? CPS1 : CPS0, SUBB = (ROWS ? TCB : SLB) * 2048;  // chunks per slot, bytes per sub-image
    constexpr int NSTEP = 2 * NKC;
    constexpr int PD = 3;  // fragment reads run PD steps ahead of their MFMAs (2 .. 4 MFMAs per step do not cover an LDS round trip)
    f16x8 w[PD + 1][NBW];
    const unsigned char *sb = nullptr;
    auto fetch = [&](int st) __attribute__((always_inline)) {  // (st is a compile-time constant at every call)
      const int kn = st >> 1, s2n = st & 1;
      if (kn % SUBS == 0 && s2n == 0) sb = slot_ready();
#pragma unroll
      for (int cb = 0; cb < NBW; ++cb)
        if (cb < nb) w[st % (PD + 1)][cb] = wfrag(sb, (kn % SUBS) * SUBB, cb, s2n);
    };
#pragma unroll
    for (int st = 0; st < PD; ++st)
      if (st < NSTEP) fetch(st);
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
      const int kc = st >> 1, st2 = st & 1;
      if (st + PD < NSTEP) fetch(st + PD);
#pragma unroll
      for (int cb = 0; cb < NBW; ++cb)
        if (cb < nb) {
          if (ROWS) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bq[2 * kc + st2], w[st % (PD + 1)][cb], acc[cb], 0, 0, 0);
          else acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[st % (PD + 1)][cb], bq[2 * kc + st2], acc[cb], 0, 0, 0);
        }
    }
  };
  // GroupNorm statistics of a slab (D layout; bias included): per-channel scale / shift into gsh.  nn: channels >= nn pass through
  auto slab_stats = [&](f32x16 (&v)[SLB], int nb, const float *vec_l, int cb_base, int n, int gs, int nn, float inv_count) __attribute__((always_inline)) {
#pragma unroll
    for (int cb = 0; cb < SLB; ++cb) {
      if (cb >= nb) break;
      float s[4], ss[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x2 lo = {v[cb][4 * q], v[cb][4 * q + 1]}, hi = {v[cb][4 * q + 2], v[cb][4 * q + 3]};
        const f32x2 t = lo + hi;
        const f32x2 tt = __builtin_elementwise_fma(hi, hi, lo * lo);
        s[q] = t[0] + t[1];
        ss[q] = tt[0] + tt[1];
      }
      if (gs == 16) { s[0] += s[1]; ss[0] += ss[1]; s[2] += s[3]; ss[2] += ss[3]; }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (gs == 16 && (q & 1)) continue;
        s[q] = hw_sum_hi(s[q]);
        ss[q] = hw_sum_hi(ss[q]);
      }
      if (col == 31) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<f32x2 *>(red + ((((wave * SLB + cb) * 2 + half) * 4 + q) * 2)) = f32x2{s[q], ss[q]};
      }
    }
    lds_bar();
    // one half wave per block, one channel per lane (blocks wave, wave + NW)
    if (wave + NW * half < nb) {
      const int cb = wave + NW * half, c = col, q = c >> 3, hh = (c >> 2) & 1;
      f32x2 t = {0.f, 0.f};
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const float *pw = red + (((w * SLB + cb) * 2) * 4) * 2;
        if (gs == 4) t += *reinterpret_cast<const f32x2 *>(pw + ((hh * 4 + q) * 2));
        else if (gs == 8) t += *reinterpret_cast<const f32x2 *>(pw + (q * 2)) + *reinterpret_cast<const f32x2 *>(pw + ((4 + q) * 2));
        else t += *reinterpret_cast<const f32x2 *>(pw + ((q & 2) * 2)) + *reinterpret_cast<const f32x2 *>(pw + ((4 + (q & 2)) * 2));
      }
      const float mean = t[0] * inv_count;
      const float var = fmaxf(t[1] * inv_count - mean * mean, 0.f);
      const int cg = (cb_base + cb) * 32 + c;
      float g = vec_l[n + cg] * __builtin_amdgcn_rsqf(var + GN_EPS);
      float sh = vec_l[2 * n + cg] - mean * g;
      if (cg >= nn) { g = 1.f; sh = 0.f; }
      gsh[(cb * 2 + 0) * 32 + c] = g;
      gsh[(cb * 2 + 1) * 32 + c] = sh;
    }
    lds_bar();
  };
  // normalise block cb of a slab, [ReLU], [+ addp (16 fp32 in the D layout's channel order)], fp16, re-paired between the lane
  // halves: o[p] = the 8 consecutive channels 16 p + 8 half of the lane's row
  auto norm_pack = [&](const f32x16 &v, int cb, bool relu, const float *addp, f16x8 (&o)[2]) __attribute__((always_inline)) {
    uint32_t u[8];
    const f16x2 zero2 = {0, 0};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 g = *reinterpret_cast<const float4 *>(gsh + (cb * 2 + 0) * 32 + 8 * q + 4 * half);
      const float4 sh = *reinterpret_cast<const float4 *>(gsh + (cb * 2 + 1) * 32 + 8 * q + 4 * half);
      const f32x2 lo = __builtin_elementwise_fma(f32x2{v[4 * q], v[4 * q + 1]}, f32x2{g.x, g.y}, f32x2{sh.x, sh.y});
      const f32x2 hi = __builtin_elementwise_fma(f32x2{v[4 * q + 2], v[4 * q + 3]}, f32x2{g.z, g.w}, f32x2{sh.z, sh.w});
      f16x2 l2 = __builtin_convertvector(lo, f16x2), h2 = __builtin_convertvector(hi, f16x2);
      if (relu) { l2 = __builtin_elementwise_max(l2, zero2); h2 = __builtin_elementwise_max(h2, zero2); }
      if (addp) {
        const float4 ad = *reinterpret_cast<const float4 *>(addp + 8 * q + 4 * half);
        l2 += __builtin_convertvector(f32x2{ad.x, ad.y}, f16x2);
        h2 += __builtin_convertvector(f32x2{ad.z, ad.w}, f16x2);
      }
      u[2 * q] = __builtin_bit_cast(uint32_t, l2);
      u[2 * q + 1] = __builtin_bit_cast(uint32_t, h2);
    }
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\t"
                 "v_permlane32_swap_b32 %4, %6\n\tv_permlane32_swap_b32 %5, %7\n\ts_nop 1"
                 : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]));
    o[0] = __builtin_bit_cast(f16x8, u32x4{u[0], u[1], u[2], u[3]});
    o[1] = __builtin_bit_cast(f16x8, u32x4{u[4], u[5], u[6], u[7]});
  };

  // ================================================================================================ Mlp tail -> mo
  f16x8 mof[2 * NBM];  // operand fragments of mo: K block kb = 16 channels, this lane's row
  {
    f16x8 h2f[REST ? 2 * NB1 : 1];
    if (REST) {  // h2 = relu(GN(W1 . h1 + b1)) + add1
#pragma unroll
      for (int sl = 0; sl < (NB1 + SLB - 1) / SLB; ++sl) {
        const int nb = NB1 - SLB * sl < SLB ? NB1 - SLB * sl : SLB;
        f32x16 acc[SLB];
        init_acc(acc, vec1_l + sl * SLB * 32, nb);
        run_slab(acc, nb, a.k1 >> 5, std::integral_constant<int, 0>(), t1a, t1b);
        slab_stats(acc, nb, vec1_l, sl * SLB, a.n1, a.gs1, a.n1, a.inv1);
#pragma unroll
        for (int cb = 0; cb < SLB; ++cb) {
          if (cb >= nb) break;
          f16x8 o[2];
          norm_pack(acc[cb], cb, true, add1_l + (sl * SLB + cb) * 32, o);
          h2f[REST ? 2 * (sl * SLB + cb) : 0] = o[0]; h2f[REST ? 2 * (sl * SLB + cb) + 1 : 0] = o[1];
        }
      }
    }
    BB_STAMP(2);
#pragma unroll
    for (int sl = 0; sl < (NBM + SLB - 1) / SLB; ++sl) {
      const int nb = NBM - SLB * sl < SLB ? NBM - SLB * sl : SLB;
      f32x16 acc[SLB];
      init_acc(acc, vecm_l + sl * SLB * 32, nb);
      if (REST) {
        run_reg(std::integral_constant<int, REST ? NB1 : 1>(), std::integral_constant<int, SLB>(), std::false_type(), acc, nb, h2f);
      } else {
        run_slab(acc, nb, a.k1 >> 5, std::integral_constant<int, 0>(), t1a, t1b);
      }
      slab_stats(acc, nb, vecm_l, sl * SLB, a.n_mo, a.gsm, a.n_mo, a.invm);
#pragma unroll
      for (int cb = 0; cb < SLB; ++cb) {
        if (cb >= nb) break;
        const int cg = (sl * SLB + cb) * 32;
        f16x8 o[2];
        norm_pack(acc[cb], cb, true, addm_l + cg, o);
#pragma unroll
        for (int p = 0; p < 2; ++p) {  // pair residual: res_connect(p, j) = ra[q] + rb[p] (+ d2 rvd + w rvw)
          const int pc = ((cg + 16 * p) >> 3) + half;
          f16x8 r8 = *reinterpret_cast<const f16x8 *>(tra + pc * NR * 16 + raoff) + *reinterpret_cast<const f16x8 *>(trb + pc * NR * 16 + rboff);
          if (FP) {
            const float *vd = rv_l + cg + 16 * p + 8 * half, *vw = rv_l + a.n_mo + cg + 16 * p + 8 * half;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              f16x2 y = {r8[2 * i], r8[2 * i + 1]};
              y = __builtin_elementwise_fma(d2s, f16x2{(T)vd[2 * i], (T)vd[2 * i + 1]}, y);
              y = __builtin_elementwise_fma(ws, f16x2{(T)vw[2 * i], (T)vw[2 * i + 1]}, y);
              r8[2 * i] = y[0]; r8[2 * i + 1] = y[1];
            }
          }
          mof[2 * (sl * SLB + cb) + p] = o[p] + r8;
        }
      }
    }
  }
  BB_STAMP(3);
  // ================================================================================================ keys -> u
  f16x8 uf[2 * NBU];
#pragma unroll
  for (int sl = 0; sl < (NBU + SLB - 1) / SLB; ++sl) {
    const int nb = NBU - SLB * sl < SLB ? NBU - SLB * sl : SLB;
    f32x16 acc[SLB];
    init_acc(acc, vecu_l + sl * SLB * 32, nb);
    run_slab(acc, nb, a.kk >> 5, std::integral_constant<int, 1>(), tka, tkb);
    // + P[point], ReLU (before the norm)
#pragma unroll
    for (int cb = 0; cb < SLB; ++cb) {
      if (cb >= nb) break;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f16x4 pv = *reinterpret_cast<const f16x4 *>(P_l + p_pt * a.n_u + (sl * SLB + cb) * 32 + 8 * q + 4 * half);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[cb][4 * q + i] = fmaxf(acc[cb][4 * q + i] + (float)pv[i], 0.f);
      }
    }
    slab_stats(acc, nb, vecu_l, sl * SLB, a.n_u, a.gsu, a.nnu, a.invu);
#pragma unroll
    for (int cb = 0; cb < SLB; ++cb) {
      if (cb >= nb) break;
      f16x8 o[2];
      norm_pack(acc[cb], cb, false, nullptr, o);
      uf[2 * (sl * SLB + cb)] = o[0]; uf[2 * (sl * SLB + cb) + 1] = o[1];
    }
  }
  BB_STAMP(4);
  // ================================================================================================ attention tail
  // per 64-channel column block: scores = u . W5^T, values = mo . Wv^T as D[row][channel] (operands swapped: the register
  // fragments are the A side); slots carry 4 chunk images of 64 rows: all score chunks of the block, then all value chunks
  const float *b_s = vect_l, *b_v = vect_l + a.n_out, *gam = vect_l + 2 * a.n_out, *bet = vect_l + 3 * a.n_out;
  for (int cbk = 0; cbk < a.n_out / (TCB * 32); ++cbk) {
    f32x16 sacc[TCB], vacc[TCB];
#pragma unroll
    for (int i = 0; i < TCB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) { sacc[i][r] = 0.f; vacc[i][r] = 0.f; }
    run_reg(std::integral_constant<int, NBU>(), std::integral_constant<int, TCB>(), std::true_type(), sacc, TCB, uf);
    run_reg(std::integral_constant<int, NBM>(), std::integral_constant<int, TCB>(), std::true_type(), vacc, TCB, mof);
    // values: bias, GroupNorm over the sample (this lane's channel: rows in registers, then the sample's waves through LDS)
#pragma unroll
    for (int cb = 0; cb < TCB; ++cb) {
      const float bv = b_v[(cbk * TCB + cb) * 32 + col];
      float s = 0.f, ss = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float x = vacc[cb][r] + bv;
        vacc[cb][r] = x;
        s += x;
        ss = fmaf(x, x, ss);
      }
      s += oth_half(s);
      ss += oth_half(ss);
      if (half == 0) *reinterpret_cast<f32x2 *>(redt + ((wave * TCB + cb) * 32 + col) * 2) = f32x2{s, ss};
    }
    lds_bar();
#pragma unroll
    for (int cb = 0; cb < TCB; ++cb) {
      f32x2 t = {0.f, 0.f};
#pragma unroll
      for (int w = 0; w < NW; ++w) t += *reinterpret_cast<const f32x2 *>(redt + ((w * TCB + cb) * 32 + col) * 2);
      float s = t[0], ss = t[1];
      if (a.gsv >= 2) { s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0xB1, 0xF, 0xF, true));
                        ss += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ss), 0xB1, 0xF, 0xF, true)); }
      if (a.gsv >= 4) { s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x4E, 0xF, 0xF, true));
                        ss += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ss), 0x4E, 0xF, 0xF, true)); }
      if (a.gsv >= 8) { s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x141, 0xF, 0xF, true));
                        ss += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ss), 0x141, 0xF, 0xF, true)); }
      if (a.gsv >= 16) { s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x140, 0xF, 0xF, true));
                         ss += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ss), 0x140, 0xF, 0xF, true)); }
      if (a.gsv >= 32) { s += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(s), 0x401F));
                         ss += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(ss), 0x401F)); }
      const int ch = (cbk * TCB + cb) * 32 + col;
      const float mean = s * a.invv;
      const float var = fmaxf(ss * a.invv - mean * mean, 0.f);
      float g = gam[ch] * __builtin_amdgcn_rsqf(var + GN_EPS);
      float bt = bet[ch] - mean * g;
      if (ch >= a.nnv) { g = 1.f; bt = 0.f; }
      const float bs = b_s[ch];
      // softmax over the KN neighbour rows of every point of this wave's 32-row block, weighted sum of the values
#pragma unroll
      for (int pg = 0; pg < GPB; ++pg) {
        float sc[RPG], vv[RPG];
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < RPG; ++j) {
          sc[j] = sacc[cb][pg * RPG + j] + bs;
          vv[j] = fmaxf(fmaf(vacc[cb][pg * RPG + j], g, bt), 0.f);
          m = fmaxf(m, sc[j]);
        }
        m = fmaxf(m, oth_half(m));
        float den = 0.f, num = 0.f;
#pragma unroll
        for (int j = 0; j < RPG; ++j) {
          const float e = __expf(sc[j] - m);
          den += e;
          num = fmaf(e, vv[j], num);
        }
        den += oth_half(den);
        num += oth_half(num);
        if (half == 0) {
          const T v = (T)(num / den);
          const size_t prow = (size_t)b * 16 + wave * GPB + pg;
          reinterpret_cast<T *>(a.out)[prow * a.out_ld + ch] = v;
          if (a.out2 && ch < a.out2_n) reinterpret_cast<T *>(a.out2)[prow * a.out2_ld + ch] = v;
        }
      }
    }
    lds_bar();  // redt is rewritten by the next column block
  }
  BB_STAMP(5);
#ifdef SLIDE_TIMELINE
  if (a.dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); BB_STAMP(6); }
#endif
}

template <int NPXL, bool REST, int NB1, int NBM, int NBU>
int launch_body(const BodyArgs &a, size_t shm, hipStream_t s) {
  static bool attr_done[64] = {};
  int d = 0;
  (void)hipGetDevice(&d);
  d = d >= 0 && d < 64 ? d : 0;
  if (!attr_done[d]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&block_body_kernel<NPXL, REST, NB1, NBM, NBU>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done[d] = true;
  }
  hipLaunchKernelGGL((block_body_kernel<NPXL, REST, NB1, NBM, NBU>), dim3(a.B), dim3(64 << (NPXL - 5)), shm, s, a);
  return (int)hipGetLastError();
}

}  // namespace

// SLIDE_OP_BLOCK_BODY (include/slide_engine.h)
int slide_launch_block_body(const SlideOp &o, hipStream_t s) {
  const BodyArgs *host = (const BodyArgs *)o.p[0];  // HOST pointer to the argument block (kept alive by the plan)
  if (!host) return -3;
  BodyArgs a = *host;
  a.dbg = (unsigned long long *)o.p[1];
  const int npxl = o.i[0], rest = o.i[1];
  const bool fp = npxl == 7;
  const int nw = 1 << (npxl - 5);
  auto okgs = [](int g) { return g == 4 || g == 8 || g == 16; };
  if (a.B <= 0 || a.k1 % 32 || a.kk % 32 || a.n_mo % 32 || a.n_u % 32 || a.n_out % (fp ? 128 : 64) || a.t_ld % 8 || a.p_ld % 8) return -3;
  if (!okgs(a.gsm) || !okgs(a.gsu) || (rest && !okgs(a.gs1)) || a.gsv > 32) return -3;
  if (fp && (!a.nbr || !a.d2 || !a.w || !a.vv || !a.rv)) return -3;
  const size_t fl = (size_t)(rest ? 4 * a.n1 : 0) + 4 * a.n_mo + 3 * a.n_u + 4 * a.n_out + (fp ? 2 * a.n_mo : 0) +
                    nw * (fp ? 8 : 4) * 2 * 4 * 2 + (fp ? 8 : 4) * 2 * 32 + nw * (fp ? 4 : 2) * 32 * 2;
  const size_t hl = (size_t)16 * a.n_u + a.k1 + 2 * a.kk + (fp ? 2 * a.k1 + 2 * a.kk : 0);
  const size_t tl = (size_t)2 * 256 * ((a.k1 >> 3) + (a.n_mo >> 3) + (a.kk >> 3));
  const size_t shm = (size_t)BB_NST * SLOT_B + fl * 4 + hl * 2 + tl + 64;
  if (shm > 160 * 1024) return -8;
  const int nb1 = a.n1 >> 5, nbm = a.n_mo >> 5, nbu = a.n_u >> 5;
#define BODY(L, R, N1, NM, NU) if (npxl == L && (rest != 0) == R && (!R || nb1 == N1) && nbm == NM && nbu == NU) return launch_body<L, R, N1, NM, NU>(a, shm, s)
  BODY(7, false, 0, 4, 4);   // FP0
#ifdef SLIDE_EXPERIMENTS
  BODY(8, true, 4, 8, 5);    // SA0 (spills 96 registers; SLIDE_BODY=2)
#else
  if (npxl == 8) return -20;  // experiments build only
#endif
#undef BODY
  return -4;
}
