// gemm_gxs.hip -- the K-expanded kernels of the SPLIT-arithmetic pair decomposition (round 5).
//
// The position DDPM has to be fp32-grade (north_star: generated latents within 1e-3; its fp16 plan is 1.5e-3 .. 3.2e-3 off on
// single forwards), and the fp32-structured split plan of round 4 moved ~0.8 GB of K-expanded fp32
// activations per step (batch 256).  This file carries the pair decomposition of gemm_gx.hip into the split arithmetic:
//   * the per-point tables ta / tb stay FLOAT (pair_norm2_kernel<., float>);
//   * gemm_gxs_kernel is the consumer GEMM of a block's first layer: x(p, j) = max(ta[q] + tb[p] (+ d2 vd + w vw), 0) (+ add |
//     * scale + shift) is GENERATED in fp32 from the two 16-row tables (L1 / L2 resident: 2 x 16 x k_pad x 4 B per sample) while the
//     chunk goes to LDS, split there into two fp16 terms x = hi + 2^-11 lo like the weights, three fp16 MFMAs per product
//     (hi hi + 2^-11 (hi lo + lo hi)) into two fp32 accumulator sets -- the 256- / 128-row first-layer outputs (h1, r, keys:
//     288 channels x 65536 rows x 4 B at the position net's SA1) are never written or read;
//   * attn_tail_split_kernel is the tail of an AttentionModule (attention.py:86-95) on float rows: scores S = W5 u + b5 and values
//     V = relu(GN(Wv mo + bv)) as split contractions with SWAPPED operand roles (a lane owns a channel, its registers run over
//     the rows), soft-max over a point's neighbours and the weighted sum in registers -- S and V never reach memory.
// Reference: pointnet2_ops/pointnet2_utils.py:383-430, :497-524 (grouped inputs), pointnet2_ops/pointnet2_modules.py:119-176
// (Mlp_plus_t_emb), pointnet2_ops/attention.py:70-96.
#include "gemm_common.h"
#include "pair_norm.h"

namespace {

constexpr int GXS_MAX_DEVICES = 64;

__device__ __forceinline__ void gxs_split4(const float4 v, f16x4 &hi, f16x4 &lo) {
  // x = hi + 2^-11 lo: the scaling keeps lo a normal fp16 number at any magnitude (engine.hip, gemm_kernel<SPLIT>)
  hi = f16x4{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
  lo = f16x4{(_Float16)((v.x - (float)hi[0]) * 2048.f), (_Float16)((v.y - (float)hi[1]) * 2048.f),
             (_Float16)((v.z - (float)hi[2]) * 2048.f), (_Float16)((v.w - (float)hi[3]) * 2048.f)};
}
// ONE accumulator set for the three products (round 5): the WEIGHT's high term also enters scaled, hs = 2^11 hi (exact in fp16 while
// |w| < 32 -- the plan checks its weights), so   2^11 (w x) = hs xh + hi xl' + wl' xh   (xl', wl' the scaled low terms) accumulates
// in one fp32 register set and a final exact 2^-11 restores the scale: the cross products sit 2^-11 below the leading one and keep 13
// of their own bits in the fp32 sum -- 2^-24 of the total, as with two sets -- at half the accumulator registers.
__device__ __forceinline__ void gxs_split4w(const float4 v, f16x4 &hi, f16x4 &hs, f16x4 &lo) {
  gxs_split4(v, hi, lo);
  hs = hi * f16x4{(_Float16)2048.f, (_Float16)2048.f, (_Float16)2048.f, (_Float16)2048.f};
}

// ------------------------------------------------------------------------------------------------ generated-X split GEMM
// Tile 256 rows (one 16 x 16 sample or two 16 x 8 samples) x 64 channels, four waves x 64 rows, K chunks of 32.  One LDS stage of
// five fp16 planes [X hi | X lo | W hi | 2^11 W hi | W lo] (rows of LDK = 40 halves: 56 KB); the next chunk's table rows and weights wait in
// registers.  Thread (l_row = tid / 8, l_c = 4 (tid % 8)) generates columns l_c .. l_c + 3 of tile rows l_row + 32 p, p = 0 .. 7:
// 16 x 16-row samples run in NATURAL neighbour order, so its eight rows share ONE neighbour row (q = l_row % 16) and touch eight
// centre rows; 16 x 8-row samples look their neighbours up once (nbr table) into a per-tile-row LDS table (offset, d2, w).
// The per-sample vectors (add | scale, shift | vd, vw) are staged once per workgroup in LDS.
// CHAIN (16 x 16-row samples, mode 0; round 5): the layer's output -- h2 = relu(GN(second_mlp(h1))) + class embedding of an SA block,
// at most 64 channels: one column tile -- is NOT stored: the common epilogue leaves it in the accumulators (KEEP), each 32-channel
// block of them IS one K chunk of the next layer (rest_mlp), written as split planes into the dead X stage, and the workgroup runs that
// layer's column tiles with its own epilogue (GroupNorm, ReLU, PAIR residual) -- one launch and one K-expanded round trip less per block.
template <int NPXL, int MODE, bool CHAIN = false>
__device__ __forceinline__ void gemm_gxs_body(const GemmArgs &a, const int bid) {
  constexpr int NPX = 1 << NPXL;
  constexpr bool FP = NPXL == 7;
  constexpr int NSAMP = TM >> NPXL;  // 1 or 2
  constexpr int CBW = 2, TN = 64;
  constexpr int LDK = TileT<SLIDE_PREC_SPLIT>::LDK;
  constexpr int XP = 8, WP = 2;                        // tile rows / weight rows per thread and chunk (32 rows per pass)
  constexpr int NVEC = (MODE ? 2 : 1) + (FP ? 2 : 0);  // float vectors per sample in LDS: [add | scale, shift][vd, vw][k_pad]
  constexpr int STAGE = (2 * TM + 3 * TN) * LDK;       // halves: planes [X hi | X lo | W hi | W hi 2^11 | W lo]
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  _Float16 *const Xh = reinterpret_cast<_Float16 *>(smem_raw);
  _Float16 *const Xl = Xh + TM * LDK, *const Wh = Xl + TM * LDK, *const Ws = Wh + TN * LDK, *const Wl = Ws + TN * LDK;
  const int ntc = (a.n_cob + CBW - 1) / CBW;
  const int xcd = bid & 7, q0 = bid >> 3;
  const int tc = q0 % ntc, tr = (q0 / ntc) * 8 + xcd;  // the column tiles of a row tile share one XCD's L2 (tables + weights)
  if (tr * TM >= a.rows) return;
  const int row0 = tr * TM, cob0 = tc * CBW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, col = lane & 31;
  const int l_row = tid >> 3, l_c = (tid & 7) * 4;
  const float *ta = reinterpret_cast<const float *>(a.gx_ta), *tb = reinterpret_cast<const float *>(a.gx_tb);
  const float *W = reinterpret_cast<const float *>(a.W);
  const int nsm = a.rows >> NPXL, smp0 = row0 >> NPXL;

  uint32_t *const epi_lds = reinterpret_cast<uint32_t *>(smem_raw + (size_t)STAGE * 2);
  float *const vec_lds = reinterpret_cast<float *>(epi_lds + CBW * EPI_DW + (CBW * EPI_DW) % 4);
  float *const vv_l = vec_lds + CBW * 96;  // [sample][NVEC][k_pad]
  SLIDE_STAMP(a, 0);
  stage_epilogue_tables<CBW>(a, cob0, tid, epi_lds, vec_lds);
  // this thread's eight tile rows: neighbour (a) and centre (b) table rows (element offsets: the tables are far below 2^31
  // elements).  16 x 8-row samples: the neighbour's table row and the two per-slot scalars of every tile row live in LDS
  // ([256] x (offset, d2, w): kept in registers they push the kernel over its 256)
  int aoff0 = 0, boff[XP];
  int *const arow_l = reinterpret_cast<int *>(vv_l + (size_t)NSAMP * NVEC * a.k_pad);
  float *const d2_l = reinterpret_cast<float *>(arow_l + TM), *const w_l = d2_l + TM;
#pragma unroll
  for (int p = 0; p < XP; ++p) {
    int row = row0 + p * 32 + l_row;
    row = row < a.rows ? row : a.rows - 1;
    const int smp = row >> NPXL, pxl = row & (NPX - 1);
    if (!FP) {
      if (p == 0) aoff0 = (smp * 16 + (pxl & 15)) * a.gx_ld + l_c;
      boff[p] = (smp * 16 + (pxl >> 4)) * a.gx_ld + l_c;
    } else {
      boff[p] = (smp * 16 + (pxl >> 3)) * a.gx_ld + l_c;
    }
  }
  if (FP) {
    int row = row0 + tid;
    row = row < a.rows ? row : a.rows - 1;
    const int smp = row >> NPXL, pxl = row & (NPX - 1);
    const int slot = (smp * 16 + (pxl >> 3)) * 16 + (pxl & 7);
    arow_l[tid] = (smp * 16 + a.gidx[slot]) * a.gx_ld;
    d2_l[tid] = a.gx_d2[slot];
    w_l[tid] = a.gx_w[slot];
  }

  f32x16 acc[CBW][2];
#pragma unroll
  for (int i = 0; i < CBW; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // raw table rows / weight rows of the chunk in flight.  16 x 8-row samples (a gathered neighbour row per tile row: twice the
  // registers) prefetch the first NPRE rows under the MFMAs and fetch the rest at the top of store_chunk, where no MFMA operand
  // is live -- all eight in flight across the MFMAs spills 36 registers
  constexpr int NPRE = FP ? 4 : XP;
  float4 ar[FP ? XP : 1], br[XP], wr[WP];
  auto load_rows = [&](int kc, auto lo_tag, auto hi_tag) __attribute__((always_inline)) {
    constexpr int P0 = decltype(lo_tag)::value, P1 = decltype(hi_tag)::value;
    const int ko = kc * BK;
#pragma unroll
    for (int p = P0; p < P1; ++p) {
      if (FP) ar[FP ? p : 0] = *reinterpret_cast<const float4 *>(ta + arow_l[p * 32 + l_row] + l_c + ko);
      else if (p == 0) ar[0] = *reinterpret_cast<const float4 *>(ta + aoff0 + ko);
      br[p] = *reinterpret_cast<const float4 *>(tb + boff[p] + ko);
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using IP = std::integral_constant<int, NPRE>;
  using IX = std::integral_constant<int, XP>;
  auto load_chunk = [&](int kc) __attribute__((always_inline)) {
    const int ko = kc * BK;
    load_rows(kc, I0(), IP());
#pragma unroll
    for (int p = 0; p < WP; ++p) {
      const int gco = cob0 * 32 + p * 32 + l_row;
      wr[p] = gco < a.n_cob * 32 ? *reinterpret_cast<const float4 *>(W + (size_t)gco * a.k_pad + ko + l_c)
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_chunk = [&](int kc) __attribute__((always_inline)) {
    const int ko = kc * BK + l_c;
    if constexpr (NPRE < XP) load_rows(kc, IP(), IX());
#pragma unroll
    for (int p = 0; p < XP; ++p) {
      const float *vp = vv_l + (size_t)(FP ? (p >> 2) : 0) * NVEC * a.k_pad + ko;  // (two samples: tile rows 0..127 | 128..255)
      const float4 av = ar[FP ? p : 0], bv = br[p];
      float4 x = make_float4(av.x + bv.x, av.y + bv.y, av.z + bv.z, av.w + bv.w);
      if (FP) {
        const float4 vd = *reinterpret_cast<const float4 *>(vp + (MODE ? 2 : 1) * a.k_pad);
        const float4 vw = *reinterpret_cast<const float4 *>(vp + (MODE ? 3 : 2) * a.k_pad);
        const float d2 = d2_l[p * 32 + l_row], w_ = w_l[p * 32 + l_row];
        x.x = fmaf(w_, vw.x, fmaf(d2, vd.x, x.x)); x.y = fmaf(w_, vw.y, fmaf(d2, vd.y, x.y));
        x.z = fmaf(w_, vw.z, fmaf(d2, vd.z, x.z)); x.w = fmaf(w_, vw.w, fmaf(d2, vd.w, x.w));
      }
      x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f);
      const float4 v0 = *reinterpret_cast<const float4 *>(vp);
      if (MODE) {
        const float4 v1 = *reinterpret_cast<const float4 *>(vp + a.k_pad);
        x.x = fmaf(x.x, v0.x, v1.x); x.y = fmaf(x.y, v0.y, v1.y); x.z = fmaf(x.z, v0.z, v1.z); x.w = fmaf(x.w, v0.w, v1.w);
      } else {
        x.x += v0.x; x.y += v0.y; x.z += v0.z; x.w += v0.w;
      }
      f16x4 hi, lo;
      gxs_split4(x, hi, lo);
      *reinterpret_cast<f16x4 *>(Xh + (p * 32 + l_row) * LDK + l_c) = hi;
      *reinterpret_cast<f16x4 *>(Xl + (p * 32 + l_row) * LDK + l_c) = lo;
    }
#pragma unroll
    for (int p = 0; p < WP; ++p) {
      f16x4 hi, hs, lo;
      gxs_split4w(wr[p], hi, hs, lo);
      *reinterpret_cast<f16x4 *>(Wh + (p * 32 + l_row) * LDK + l_c) = hi;
      *reinterpret_cast<f16x4 *>(Ws + (p * 32 + l_row) * LDK + l_c) = hs;
      *reinterpret_cast<f16x4 *>(Wl + (p * 32 + l_row) * LDK + l_c) = lo;
    }
  };
  auto compute = [&](f32x16 (&acc)[CBW][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      f16x8 ah[CBW], as[CBW], al[CBW], bh[2], bl[2];
#pragma unroll
      for (int cb = 0; cb < CBW; ++cb) {
        ah[cb] = *reinterpret_cast<const f16x8 *>(Wh + (cb * 32 + col) * LDK + st * 16 + half * 8);
        as[cb] = *reinterpret_cast<const f16x8 *>(Ws + (cb * 32 + col) * LDK + st * 16 + half * 8);
        al[cb] = *reinterpret_cast<const f16x8 *>(Wl + (cb * 32 + col) * LDK + st * 16 + half * 8);
      }
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        bh[rb] = *reinterpret_cast<const f16x8 *>(Xh + (wave * 64 + rb * 32 + col) * LDK + st * 16 + half * 8);
        bl[rb] = *reinterpret_cast<const f16x8 *>(Xl + (wave * 64 + rb * 32 + col) * LDK + st * 16 + half * 8);
      }
#pragma unroll
      for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          acc[cb][rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as[cb], bh[rb], acc[cb][rb], 0, 0, 0);
          acc[cb][rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cb], bl[rb], acc[cb][rb], 0, 0, 0);
          acc[cb][rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cb], bh[rb], acc[cb][rb], 0, 0, 0);
        }
    }
  };

  const int nk = a.k_pad / BK;
  // PROLOGUE ORDER (round 6): 16 x 16-row samples (natural neighbour order: no row table) request their first chunk BEFORE the
  // per-sample vectors are staged -- one memory round trip instead of two in front of the first MFMA
  if (!FP) load_chunk(0);
  {
    const float *addp = a.in_add;
    if (MODE == 0 && addp && a.gx_add_idx) addp += (size_t)a.gx_add_idx[0] * a.gx_add_idx_stride;  // row t of a per-timestep table
    for (int i = tid * 4; i < NSAMP * a.k_pad; i += 1024) {
      const int sl = i / a.k_pad, k = i - sl * a.k_pad;
      int smp = smp0 + sl;
      smp = smp < nsm ? smp : nsm - 1;
      float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0, vd = v0, vw = v0;
      if (MODE == 0) {
        if (addp) v0 = *reinterpret_cast<const float4 *>(addp + (size_t)smp * a.add_bs + k);
      } else {
        v0 = *reinterpret_cast<const float4 *>(a.in_scale + (size_t)smp * a.in_bs + k);
        v1 = *reinterpret_cast<const float4 *>(a.in_shift + (size_t)smp * a.in_bs + k);
      }
      if (FP && a.gx_vv) {
        vd = *reinterpret_cast<const float4 *>(a.gx_vv + (size_t)smp * a.gx_vbs + k);
        vw = *reinterpret_cast<const float4 *>(a.gx_vv + (size_t)smp * a.gx_vbs + (a.gx_vbs >> 1) + k);
      }
      float *dst = vv_l + (size_t)sl * NVEC * a.k_pad + k;
      *reinterpret_cast<float4 *>(dst) = v0;
      if (MODE) *reinterpret_cast<float4 *>(dst + a.k_pad) = v1;
      if (FP) {
        *reinterpret_cast<float4 *>(dst + (MODE ? 2 : 1) * a.k_pad) = vd;
        *reinterpret_cast<float4 *>(dst + (MODE ? 3 : 2) * a.k_pad) = vw;
      }
    }
  }
  SLIDE_STAMP(a, 7);
  if (FP) {
    __syncthreads();  // (the row table is read by load_chunk)
    load_chunk(0);
  }
  __syncthreads();  // the staged vectors and epilogue tables are visible
  store_chunk(0);
  __syncthreads();
  SLIDE_STAMP(a, 1);
  for (int kc = 0; kc < nk; ++kc) {
    if (kc + 1 < nk) load_chunk(kc + 1);
    compute(acc);
    __syncthreads();  // every wave is done reading the stage before it is overwritten
    if (kc + 1 < nk) store_chunk(kc + 1);
    __syncthreads();
  }
  SLIDE_STAMP(a, 2);
#pragma unroll
  for (int i = 0; i < CBW; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] *= 1.f / 2048.f;
  if constexpr (CHAIN) {
    // ---- this layer's epilogue with the result kept in `acc` (all its channels sit in this one column tile: cob0 == 0)
    gemm_epilogue<SLIDE_PREC_F32, NPXL, CBW, 2, false, true>(a, acc, row0, cob0, wave, half, col, epi_lds, vec_lds,
                                                             reinterpret_cast<float *>(smem_raw));
    GemmArgs b = a;
    b.W = a.ch_W; b.epi = a.ch_epi; b.vecs = a.ch_vecs; b.n_cob = a.ch_n_cob; b.k_pad = a.ch_k_pad;
    const float *W2 = reinterpret_cast<const float *>(a.ch_W);
    const int nk2 = a.ch_k_pad / BK;  // <= CBW: chunk kc2 of the next layer = channel block kc2 of this one
    for (int tc2 = 0; tc2 < (a.ch_n_cob + CBW - 1) / CBW; ++tc2) {
      __syncthreads();  // the previous epilogue is done with its tables and scratch
      stage_epilogue_tables<CBW>(b, tc2 * CBW, tid, epi_lds, vec_lds);
      f32x16 acc2[CBW][2];
#pragma unroll
      for (int i = 0; i < CBW; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;
#pragma unroll
      for (int kc2 = 0; kc2 < CBW; ++kc2) {
        if (kc2 >= nk2) break;
        if (kc2) __syncthreads();  // the stage is free again
        // X planes from the accumulators: lane (row, half) holds channels 8 q + 4 half + 0 .. 3 of block kc2
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          const int trow = wave * 64 + rb * 32 + col;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f16x4 hi, lo;
            gxs_split4(make_float4(acc[kc2][rb][4 * q], acc[kc2][rb][4 * q + 1], acc[kc2][rb][4 * q + 2], acc[kc2][rb][4 * q + 3]), hi, lo);
            *reinterpret_cast<f16x4 *>(Xh + trow * LDK + 8 * q + 4 * half) = hi;
            *reinterpret_cast<f16x4 *>(Xl + trow * LDK + 8 * q + 4 * half) = lo;
          }
        }
#pragma unroll
        for (int p = 0; p < WP; ++p) {
          const int gco = tc2 * CBW * 32 + p * 32 + l_row;
          const float4 w4 = gco < a.ch_n_cob * 32 ? *reinterpret_cast<const float4 *>(W2 + (size_t)gco * a.ch_k_pad + kc2 * BK + l_c)
                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
          f16x4 hi, hs, lo;
          gxs_split4w(w4, hi, hs, lo);
          *reinterpret_cast<f16x4 *>(Wh + (p * 32 + l_row) * LDK + l_c) = hi;
          *reinterpret_cast<f16x4 *>(Ws + (p * 32 + l_row) * LDK + l_c) = hs;
          *reinterpret_cast<f16x4 *>(Wl + (p * 32 + l_row) * LDK + l_c) = lo;
        }
        __syncthreads();
        compute(acc2);
      }
      __syncthreads();  // the stage turns into the epilogue's scratch
#pragma unroll
      for (int i = 0; i < CBW; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc2[i][j][r] *= 1.f / 2048.f;
      // (rest_mlp carries no add vector -- the class embedding is added to second_mlp's output: the NOADDV epilogue, whose 16 fewer
      //  registers keep h2 alive across this tile's epilogue without a spill)
      gemm_epilogue<SLIDE_PREC_F32, NPXL, CBW, 2, true, false, true>(b, acc2, row0, tc2 * CBW, wave, half, col, epi_lds, vec_lds,
                                                                     reinterpret_cast<float *>(smem_raw));
    }
    return;
  }
  // float rows: the fp32 epilogue (mode 0 = the Mlp layers: PAIR residual on the float tables)
  gemm_epilogue<SLIDE_PREC_F32, NPXL, CBW, 2, MODE == 0>(a, acc, row0, cob0, wave, half, col, epi_lds, vec_lds,
                                                         reinterpret_cast<float *>(smem_raw));
  SLIDE_STAMP(a, 5);
#ifdef SLIDE_TIMELINE
  if (a.dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); SLIDE_STAMP(a, 6); }
#endif
}

template <int NPXL, int MODE>
__global__ __launch_bounds__(256, 2) void gemm_gxs_kernel(GemmArgs a) {
  gemm_gxs_body<NPXL, MODE>(a, blockIdx.x);
}

// the two independent generated-X GEMMs of a block (keys -> u: mode 1; first Mlp layer: mode 0) in ONE launch: they read the same
// pair tables and nothing of each other -- one launch gap instead of two, and their workgroups fill the chip together
template <int NPXL, bool CHAIN = false>
__global__ __launch_bounds__(256, 2) void gemm_gxs_dual_kernel(GemmArgs a1, GemmArgs a0, int grid1) {
  if ((int)blockIdx.x < grid1) gemm_gxs_body<NPXL, 1>(a1, blockIdx.x);
  else gemm_gxs_body<NPXL, 0, CHAIN>(a0, blockIdx.x - grid1);
}

template <int NPXL>
__global__ __launch_bounds__(256, 2) void gemm_gxs_chain_kernel(GemmArgs a) {
  gemm_gxs_body<NPXL, 0, true>(a, blockIdx.x);
}

template <int NPXL, int MODE, bool CHAIN = false>
int launch_gxs(const GemmArgs &a, hipStream_t s) {
  constexpr int LDK = TileT<SLIDE_PREC_SPLIT>::LDK;
  constexpr int NSAMP = TM >> NPXL, NVEC = (MODE ? 2 : 1) + (NPXL == 7 ? 2 : 0);
  const size_t shm = (size_t)(2 * TM + 3 * 64) * LDK * 2 + (2 * EPI_DW + (2 * EPI_DW) % 4 + 2 * 96) * 4 + (size_t)NSAMP * NVEC * a.k_pad * 4 + (NPXL == 7 ? 3 * TM * 4 : 0) + 16;
  if (shm > 80 * 1024) return -8;
  const int ntc = (a.n_cob + 1) / 2, ntr = (a.rows + TM - 1) / TM;
  const int grid = ((ntr + 7) / 8) * 8 * ntc;
  static bool attr_done[GXS_MAX_DEVICES] = {};
  int d = 0;
  (void)hipGetDevice(&d);
  bool &attr_set = attr_done[d >= 0 && d < GXS_MAX_DEVICES ? d : 0];
  if constexpr (CHAIN) {
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_gxs_chain_kernel<NPXL>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                80 * 1024);
      attr_set = true;
    }
    hipLaunchKernelGGL((gemm_gxs_chain_kernel<NPXL>), dim3(grid), dim3(256), shm, s, a);
  } else {
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_gxs_kernel<NPXL, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                80 * 1024);
      attr_set = true;
    }
    hipLaunchKernelGGL((gemm_gxs_kernel<NPXL, MODE>), dim3(grid), dim3(256), shm, s, a);
  }
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ split attention tail
// The end of an AttentionModule (attention.py:86-95) on FLOAT rows in the split arithmetic, ONE launch: values V = relu(GN(Wv mo + bv))
// first (accumulators -> GroupNorm statistics over the sample -> normalised in place: 64 registers), then scores S = W5 u + b5 into a
// second accumulator set, soft-max over a point's K neighbours and the weighted sum -- S and V (2 x rows x C x 4 B, written and read
// back by the three-launch form: 134 MB per step at the position net's SA1) never leave the registers.  Operand roles are SWAPPED
// (A = X rows, B = W rows) as in attn_tail_kernel (engine.hip): a lane owns one channel and its registers run over the rows.
// Tile 256 rows x 64 channels, four waves x 64 rows; one LDS stage of five fp16 planes [X hi | X lo | W hi | 2^11 W hi | W lo].
struct TailSArgs {
  const float *X1, *W1, *X2, *W2;  // scores: u [rows][x1_ld] . W5 [n_cob*32][k1];  values: mo [rows][x2_ld] . Wv [n_cob*32][k2]
  const float *vec;                // [bias_s | bias_v | gamma | beta], n_cob * 32 floats each
  float *out;                      // [rows >> (NPXL - 4)][out_ld]
  float *out2;                     // optional copy of the first out2_n channels into another per-point buffer [..][out2_ld]
  int out2_ld, out2_n;
  int rows, x1_ld, k1, x2_ld, k2, n_cob, gs, n_norm, out_ld;
  float inv_count;
};

__device__ __forceinline__ float gxs_other_half(float x) {  // value of lane ^ 32
  uint32_t a = __float_as_uint(x), b = a;
  lane32_swap(a, b);
  return __uint_as_float((threadIdx.x & 32) ? a : b);
}

template <int NPXL>
__global__ __launch_bounds__(256, 2) void attn_tail_split_kernel(TailSArgs a) {
  constexpr int CBW = 2, TN = 64;
  constexpr int LDK = TileT<SLIDE_PREC_SPLIT>::LDK;
  constexpr int XP = 8, WP = 2;
  constexpr int KLOG = NPXL - 4, KN = 1 << KLOG, GPB = 32 / KN;  // neighbours per point, points per 32-row block
  constexpr int WPS = (1 << NPXL) / 64;                            // waves per sample
  constexpr int STAGE = (2 * TM + 3 * TN) * LDK;                   // halves
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  _Float16 *const Xh = reinterpret_cast<_Float16 *>(smem_raw);
  _Float16 *const Xl = Xh + TM * LDK, *const Wh = Xl + TM * LDK, *const Ws = Wh + TN * LDK, *const Wl = Ws + TN * LDK;
  const int ntc = (a.n_cob + CBW - 1) / CBW;
  const int xcd = blockIdx.x & 7, q0 = blockIdx.x >> 3;
  const int tc = q0 % ntc, tr = (q0 / ntc) * 8 + xcd;
  if (tr * TM >= a.rows) return;
  const int row0 = tr * TM, cob0 = tc * CBW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
  const int l_row = tid >> 3, l_c = (tid & 7) * 4;
  float *const vec_lds = reinterpret_cast<float *>(smem_raw + (size_t)STAGE * 2);  // [4 vectors][CBW*32]
  for (int i = tid; i < 4 * CBW * 32; i += 256) {
    const int which = i / (CBW * 32), c = i - which * (CBW * 32), gc = cob0 * 32 + c;
    vec_lds[i] = gc < a.n_cob * 32 ? a.vec[(size_t)which * a.n_cob * 32 + gc] : 0.f;
  }
  // one split contraction: acc[cb][rb] = D[row][channel] (lane: channel col of block cb; reg r: row (r&3) + 8 (r>>2) + 4 half)
  auto run = [&](const float *Xp, const float *Wp, int x_ld, int k_pad, f32x16 (&acc)[CBW][2]) __attribute__((always_inline)) {
    float4 xr[XP], wr[WP];
    auto load_chunk = [&](int kc) __attribute__((always_inline)) {
#pragma unroll
      for (int p = 0; p < XP; ++p) {
        int grow = row0 + p * 32 + l_row;
        grow = grow < a.rows ? grow : a.rows - 1;
        xr[p] = *reinterpret_cast<const float4 *>(Xp + (size_t)grow * x_ld + kc * BK + l_c);
      }
#pragma unroll
      for (int p = 0; p < WP; ++p) {
        int gco = cob0 * 32 + p * 32 + l_row;
        gco = gco < a.n_cob * 32 ? gco : a.n_cob * 32 - 1;  // (channels beyond the matrix are never stored)
        wr[p] = *reinterpret_cast<const float4 *>(Wp + (size_t)gco * k_pad + kc * BK + l_c);
      }
    };
    auto store_chunk = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int p = 0; p < XP; ++p) {
        f16x4 hi, lo;
        gxs_split4(xr[p], hi, lo);
        *reinterpret_cast<f16x4 *>(Xh + (p * 32 + l_row) * LDK + l_c) = hi;
        *reinterpret_cast<f16x4 *>(Xl + (p * 32 + l_row) * LDK + l_c) = lo;
      }
#pragma unroll
      for (int p = 0; p < WP; ++p) {
        f16x4 hi, hs, lo;
        gxs_split4w(wr[p], hi, hs, lo);
        *reinterpret_cast<f16x4 *>(Wh + (p * 32 + l_row) * LDK + l_c) = hi;
        *reinterpret_cast<f16x4 *>(Ws + (p * 32 + l_row) * LDK + l_c) = hs;
        *reinterpret_cast<f16x4 *>(Wl + (p * 32 + l_row) * LDK + l_c) = lo;
      }
    };
#pragma unroll
    for (int i = 0; i < CBW; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nk = k_pad / BK;
    load_chunk(0);
    store_chunk();
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
      if (kc + 1 < nk) load_chunk(kc + 1);
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        f16x8 xh[2], xl[2], wh[CBW], ws[CBW], wl[CBW];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          xh[rb] = *reinterpret_cast<const f16x8 *>(Xh + (wave * 64 + rb * 32 + col) * LDK + st * 16 + half * 8);
          xl[rb] = *reinterpret_cast<const f16x8 *>(Xl + (wave * 64 + rb * 32 + col) * LDK + st * 16 + half * 8);
        }
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb) {
          wh[cb] = *reinterpret_cast<const f16x8 *>(Wh + (cb * 32 + col) * LDK + st * 16 + half * 8);
          ws[cb] = *reinterpret_cast<const f16x8 *>(Ws + (cb * 32 + col) * LDK + st * 16 + half * 8);
          wl[cb] = *reinterpret_cast<const f16x8 *>(Wl + (cb * 32 + col) * LDK + st * 16 + half * 8);
        }
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
          for (int rb = 0; rb < 2; ++rb) {  // rows x channels
            acc[cb][rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[rb], ws[cb], acc[cb][rb], 0, 0, 0);
            acc[cb][rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl[rb], wh[cb], acc[cb][rb], 0, 0, 0);
            acc[cb][rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[rb], wl[cb], acc[cb][rb], 0, 0, 0);
          }
      }
      __syncthreads();  // every wave is done reading the stage
      if (kc + 1 < nk) {
        store_chunk();
        __syncthreads();
      }
    }
#pragma unroll
    for (int i = 0; i < CBW; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] *= 1.f / 2048.f;
  };
  f32x16 vacc[CBW][2], sacc[CBW][2];
  run(a.X2, a.W2, a.x2_ld, a.k2, vacc);  // values first

  // ---- values: bias, GroupNorm over the sample (rows of WPS waves x the gs adjacent channel lanes), ReLU -- in place
  float *const red = reinterpret_cast<float *>(smem_raw);  // [wave][cb][32 channels][sum, sumsq] (the stage is free: run() ends on a barrier)
  const float *b_s = vec_lds, *b_v = vec_lds + CBW * 32, *gam = vec_lds + 2 * CBW * 32, *bet = vec_lds + 3 * CBW * 32;
#pragma unroll
  for (int cb = 0; cb < CBW; ++cb) {
    const float bv = b_v[cb * 32 + col];
    const f32x2 bv2 = {bv, bv};
    f32x2 s2 = {0.f, 0.f}, ss2 = {0.f, 0.f};  // (register pairs: v_pk_add / v_pk_fma_f32)
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const f32x2 x = f32x2{vacc[cb][rb][2 * i], vacc[cb][rb][2 * i + 1]} + bv2;
        vacc[cb][rb][2 * i] = x[0]; vacc[cb][rb][2 * i + 1] = x[1];
        s2 += x;
        ss2 = __builtin_elementwise_fma(x, x, ss2);
      }
    float s = s2[0] + s2[1], ss = ss2[0] + ss2[1];
    s += gxs_other_half(s);
    ss += gxs_other_half(ss);
    if (half == 0) *reinterpret_cast<f32x2 *>(red + ((wave * CBW + cb) * 32 + col) * 2) = f32x2{s, ss};
  }
  __syncthreads();
  const int w0 = (wave / WPS) * WPS;
#pragma unroll
  for (int cb = 0; cb < CBW; ++cb) {
    f32x2 t = {0.f, 0.f};
#pragma unroll
    for (int w = 0; w < WPS; ++w) t += *reinterpret_cast<const f32x2 *>(red + (((w0 + w) * CBW + cb) * 32 + col) * 2);
    float s = t[0], ss = t[1];
    // the gs channels of a group sit in gs adjacent lanes (physical GroupNorm layout: power-of-two runs)
    if (a.gs >= 2) { s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0xB1, 0xF, 0xF, true));
                     ss += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ss), 0xB1, 0xF, 0xF, true)); }
    if (a.gs >= 4) { s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x4E, 0xF, 0xF, true));
                     ss += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ss), 0x4E, 0xF, 0xF, true)); }
    if (a.gs >= 8) { s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x141, 0xF, 0xF, true));
                     ss += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ss), 0x141, 0xF, 0xF, true)); }
    if (a.gs >= 16) { s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x140, 0xF, 0xF, true));
                      ss += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ss), 0x140, 0xF, 0xF, true)); }
    if (a.gs >= 32) { s += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(s), 0x401F));
                      ss += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(ss), 0x401F)); }
    const float mean = s * a.inv_count;
    const float var = fmaxf(ss * a.inv_count - mean * mean, 0.f);
    float g = gam[cb * 32 + col] * __builtin_amdgcn_rsqf(var + GN_EPS);
    float bt = bet[cb * 32 + col] - mean * g;
    if ((cob0 + cb) * 32 + col >= a.n_norm) { g = 1.f; bt = 0.f; }
    const f32x2 g2 = {g, g}, bt2 = {bt, bt};
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const f32x2 v = __builtin_elementwise_fma(f32x2{vacc[cb][rb][2 * i], vacc[cb][rb][2 * i + 1]}, g2, bt2);
        vacc[cb][rb][2 * i] = fmaxf(v[0], 0.f); vacc[cb][rb][2 * i + 1] = fmaxf(v[1], 0.f);
      }
  }
  __syncthreads();  // every wave has read the statistics: the stage is free for the score contraction
  run(a.X1, a.W1, a.x1_ld, a.k1, sacc);
  // ---- base-2 soft-max over the K neighbour rows of every point, weighted sum of the values, one row out per point.  Round 6: written
  // like attn_tail_finish (engine.hip) -- register pairs (v_pk_fma / v_pk_add_f32), log2 e folded into the bias step and v_exp_f32 on
  // the difference (libm's expf cost ~25 instructions per value here), v_rcp instead of an IEEE division (1 ulp), one lane-half
  // exchange for numerator and denominator together.  The scores' rounding (one fma at magnitude |s| log2 e) stays at a few 1e-6 of a
  // soft-max weight: fp32-grade (position forwards vs the fp32 mode: bench.py parity, tests/test_hip_engine.py).
  constexpr float LOG2E = 1.44269504088896340736f;
  auto pr = [](const f32x16 &v, int i) __attribute__((always_inline)) { return f32x2{v[2 * i], v[2 * i + 1]}; };
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const int pt0 = (row0 + wave_s * 64) >> KLOG, npts = a.rows >> KLOG;
#pragma unroll
  for (int cb = 0; cb < CBW; ++cb) {
    const float bsl = b_s[cb * 32 + col] * LOG2E;
    const f32x2 bs2 = {bsl, bsl}, l2 = {LOG2E, LOG2E};
    const bool cb_ok = cob0 + cb < a.n_cob;  // (uniform)
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int pg = 0; pg < GPB; ++pg) {
        // rows of point pg inside the 32-row block: 16 -> regs 8pg .. 8pg+7 (both halves); 8 -> regs 4pg .. 4pg+3
        constexpr int PPG = 8 / GPB;  // register pairs per point
        f32x2 sc[PPG];
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < PPG; ++j) {
          sc[j] = __builtin_elementwise_fma(pr(sacc[cb][rb], pg * PPG + j), l2, bs2);
          m = fmaxf(m, fmaxf(sc[j][0], sc[j][1]));
        }
        m = fmaxf(m, gxs_other_half(m));
        const f32x2 m2 = {m, m};
        f32x2 den2 = {0.f, 0.f}, num2 = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < PPG; ++j) {
          const f32x2 d = sc[j] - m2;
          const f32x2 e = {__builtin_amdgcn_exp2f(d[0]), __builtin_amdgcn_exp2f(d[1])};
          den2 += e;
          num2 = __builtin_elementwise_fma(e, pr(vacc[cb][rb], pg * PPG + j), num2);
        }
        uint32_t un = __float_as_uint(num2[0] + num2[1]), ud = __float_as_uint(den2[0] + den2[1]);
        lane32_swap(un, ud);
        const float tot = __uint_as_float(un) + __uint_as_float(ud);  // lower lanes: numerator, upper lanes: denominator
        uint32_t ua = __float_as_uint(tot), ub = ua;
        lane32_swap(ua, ub);  // ub (lower lanes) = the upper lanes' tot
        const int pidx = rb * GPB + pg;  // point of the wave
        if (half == 0 && pt0 + pidx < npts && cb_ok) {
          const float v = tot * __builtin_amdgcn_rcpf(__uint_as_float(ub));
          a.out[(size_t)(pt0 + pidx) * a.out_ld + (cob0 + cb) * 32 + col] = v;
          // second copy into the columns of a later concatenation buffer (the skip input of an FP block's second Mlp)
          if (a.out2 && (cob0 + cb) * 32 + col < a.out2_n)
            a.out2[(size_t)(pt0 + pidx) * a.out2_ld + (cob0 + cb) * 32 + col] = v;
        }
      }
  }
}


#ifdef SLIDE_EXPERIMENTS  // (opt-in variant that lost its A/B: experiments build only, SLIDE_PP=1)
// ------------------------------------------------------------------------------------------------ per-point stage
// Everything of a block that happens on the 16 points of a sample -- the per-point GEMM of the pair decomposition with the
// attention queries riding on it, the pair-table pass with the joint GroupNorm, the query half of weight_conv.2, an FP block's
// second Mlp_plus_t_emb, the output head -- is a chain of SMALL dense layers (16 rows, K <= 160, N <= 288 in the position net) whose
// every step depends on the previous one for the SAME sample only.  As launches of their own they cost the chain a launch latency
// each (17 of the split position plan's 30); here ONE workgroup per sample walks the whole chain (MEASURED SLOWER, opt-in:
// every workgroup re-reads each layer's weights and its K loop is a chain of dependent L2 round trips on four waves --
// 55 - 131 us per stage launch against 27 - 45 us for the launches it replaces; position chain alone 365 -> 652 us per step):
//   DENSE  y[16][n] = epilogue(W . x + bias): exact fp32 FMA chains on the vector ALUs (a thread owns one output channel and its 16
//          rows; x transposed in LDS, read as broadcasts; the weights K-major so that a wave's loads coalesce) -- 16-row layers of
//          this size do not fill a matrix tile, and fp32 FMAs need no operand split.  The epilogue is SlideEpi's (include/
//          slide_engine.h): bias, PRE_RELU, GroupNorm over (gs channels x 16 rows) or STATS sums, POST_RELU, + addvec, + residual.
//   PAIR   pair_norm2_body (pair_norm.h): tables ta / tb, key statistics, joint GroupNorm scale / shift.
// A step reads what the previous step of the same workgroup wrote through L2 (device-scope fence + barrier between steps).
struct PPDense {
  const float *X, *Wt;        // X [B*16][x_ld]; Wt K-major [k_pad][n_cob*32]
  const SlideEpi *epi;        // [n_cob]
  const float *in_scale, *in_shift;  // consumer-side affine x' = x scale[b*in_bs + k] + shift[...] or NULL
  int x_ld, k_pad, n_cob, in_bs;
};
struct PPPair {
  const float *y, *xyz, *wa, *wb;
  const SlideEpi *epi;
  float *ta, *tb;
  const int *nbr;
  const float *d2t, *wt, *vv_in;
  float *vv_out;
  const SlideGnFin *fin;
  int ld, K;
};
constexpr int PP_MAX = 8, PP_NT = 256, PP_KMAX = 192;
struct PPArgs {
  int n, B;
  int kind[PP_MAX];   // 0 = dense d[idx], 1 = pair p[idx]
  int idx[PP_MAX];
  PPDense d[PP_MAX];
  PPPair p[2];
};

__device__ __forceinline__ void pp_dense(const PPDense &d, const int b, float *xs) {
  const int tid = threadIdx.x;
  const int n = d.n_cob * 32;
  // x transposed into LDS: xs[k][16 rows] (+ the consumer-side affine)
  for (int i = tid; i < 16 * d.k_pad; i += PP_NT) {
    const int r = i / d.k_pad, k = i - r * d.k_pad;
    float v = d.X[((size_t)b * 16 + r) * d.x_ld + k];
    if (d.in_scale) v = fmaf(v, d.in_scale[(size_t)b * d.in_bs + k], d.in_shift[(size_t)b * d.in_bs + k]);
    xs[k * 16 + r] = v;
  }
  __syncthreads();
  for (int c0 = 0; c0 < n; c0 += PP_NT) {
    const int c = c0 + tid;
    if (c >= n) continue;  // (whole 32-lane halves: n is a multiple of 32)
    float acc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float *wp = d.Wt + c;
#pragma unroll 4
    for (int k = 0; k < d.k_pad; ++k) {
      const float w = wp[(size_t)k * n];
      const float4 x0 = *reinterpret_cast<const float4 *>(xs + k * 16), x1 = *reinterpret_cast<const float4 *>(xs + k * 16 + 4);
      const float4 x2 = *reinterpret_cast<const float4 *>(xs + k * 16 + 8), x3 = *reinterpret_cast<const float4 *>(xs + k * 16 + 12);
      acc[0] = fmaf(w, x0.x, acc[0]); acc[1] = fmaf(w, x0.y, acc[1]); acc[2] = fmaf(w, x0.z, acc[2]); acc[3] = fmaf(w, x0.w, acc[3]);
      acc[4] = fmaf(w, x1.x, acc[4]); acc[5] = fmaf(w, x1.y, acc[5]); acc[6] = fmaf(w, x1.z, acc[6]); acc[7] = fmaf(w, x1.w, acc[7]);
      acc[8] = fmaf(w, x2.x, acc[8]); acc[9] = fmaf(w, x2.y, acc[9]); acc[10] = fmaf(w, x2.z, acc[10]); acc[11] = fmaf(w, x2.w, acc[11]);
      acc[12] = fmaf(w, x3.x, acc[12]); acc[13] = fmaf(w, x3.y, acc[13]); acc[14] = fmaf(w, x3.z, acc[14]); acc[15] = fmaf(w, x3.w, acc[15]);
    }
    // ---- epilogue of channel c (SlideEpi of its 32-channel block; a group's channels are adjacent lanes)
    const SlideEpi e = d.epi[c >> 5];
    const int cl = c & 31;
    const float bias = e.bias ? e.bias[cl] : 0.f;
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = acc[r] + bias;
      if (e.flags & SLIDE_F_PRE_RELU) v = fmaxf(v, 0.f);
      acc[r] = v;
      s += v; ss = fmaf(v, v, ss);
    }
    if (e.mode == SLIDE_EPI_STATS) {
      e.stats_sum[(size_t)b * e.stats_bs + cl] = s * e.stats_scale;
      e.stats_sq[(size_t)b * e.stats_bs + cl] = ss * e.stats_scale;
    } else if (e.mode == SLIDE_EPI_NORM) {
      for (int m = 1; m < e.gs; m <<= 1) {
        s += __shfl_xor(s, m, 64);
        ss += __shfl_xor(ss, m, 64);
      }
      const float mean = s * e.inv_count;
      const float var = fmaxf(ss * e.inv_count - mean * mean, 0.f);
      float g = e.gamma[cl] * __builtin_amdgcn_rsqf(var + GN_EPS);
      float sh = e.beta[cl] - mean * g;
      if (cl >= e.n_norm) { g = 1.f; sh = 0.f; }  // MyGroupNorm leaves the last C % G channels as they are
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = fmaf(acc[r], g, sh);
    }
    float addv = 0.f;
    if (e.addvec) {
      const float *ap = e.addvec;
      if (e.addvec_idx) ap += (size_t)e.addvec_idx[0] * e.addvec_idx_stride;  // row t of a per-timestep table
      addv = ap[(size_t)b * e.addvec_bs + cl];
    }
    const float *res = reinterpret_cast<const float *>(e.residual);
    float *out = reinterpret_cast<float *>(e.out);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = acc[r];
      if (e.flags & SLIDE_F_POST_RELU) v = fmaxf(v, 0.f);
      v += addv;
      const size_t row = (size_t)b * 16 + r;
      if (res) v += res[row * e.res_ld + cl];
      out[row * e.out_ld + cl] = v;
    }
  }
}

__global__ __launch_bounds__(PP_NT) void pp_stage_kernel(PPArgs a) {
  extern __shared__ __attribute__((aligned(16))) float pp_dyn[];  // [16][PP_NT + 1] floats of the FP pair form
  __shared__ __attribute__((aligned(16))) float xs[PP_KMAX * 16];
  const int b = blockIdx.x;
  for (int st = 0; st < a.n; ++st) {
    if (st) {
      __threadfence();   // the previous step's stores are visible device-wide (and this CU's stale lines dropped)
      __syncthreads();
    }
    if (a.kind[st] == 0) {
      pp_dense(a.d[a.idx[st]], b, xs);
    } else {
      const PPPair &p = a.p[a.idx[st]];
      if (p.K == 8)
        pair_norm2_body<true, float, PP_NT>(p.ld, p.y, p.xyz, p.wa, p.wb, p.epi, p.ta, p.tb, p.nbr, p.d2t, p.wt, p.vv_in, p.vv_out, p.fin,
                                            b, PP_NT, pp_dyn);
      else
        pair_norm2_body<false, float, PP_NT>(p.ld, p.y, p.xyz, p.wa, p.wb, p.epi, p.ta, p.tb, nullptr, nullptr, nullptr, nullptr, nullptr,
                                             p.fin, b, PP_NT, pp_dyn);
    }
  }
}

#endif  // SLIDE_EXPERIMENTS

}  // namespace

static int gxs_args_from_op(const SlideOp &o, GemmArgs &a) {
  a = GemmArgs();
  a.gx_ta = o.p[0]; a.W = o.p[1]; a.epi = (const SlideEpi *)o.p[2];
  a.in_scale = (const float *)o.p[3]; a.in_shift = (const float *)o.p[4];
  a.gx_tb = o.p[5]; a.in_add = (const float *)o.p[6]; a.gx_add_idx = (const int *)o.p[7];
  a.gidx = (const int *)o.p[8]; a.gx_d2 = (const float *)o.p[9]; a.gx_w = (const float *)o.p[10];
  a.gx_vv = (const float *)o.p[11];
  a.rows = o.i[0]; a.gx_ld = o.i[1]; a.k_pad = o.i[2]; a.n_cob = o.i[3]; a.in_bs = o.i[5];
  a.gx_mode = o.i[6]; a.add_bs = o.i[7]; a.gx_add_idx_stride = o.i[8]; a.gx_vbs = o.i[9];
  a.aff_tps = 1;
  const int npxl = o.i[4];
  if (a.k_pad % 32 || a.k_pad <= 0 || a.gx_ld % 4 || a.rows <= 0 || a.n_cob <= 0 || !a.gx_ta || !a.gx_tb) return -3;
  if ((uintptr_t)a.gx_ta % 16 || (uintptr_t)a.gx_tb % 16 || (uintptr_t)a.W % 16) return -3;
  if (a.gx_mode != 0 && (!a.in_scale || !a.in_shift || (uintptr_t)a.in_scale % 16 || (uintptr_t)a.in_shift % 16 || a.in_bs % 4)) return -3;
  if (a.in_add && ((uintptr_t)a.in_add % 16 || a.add_bs % 4 || a.gx_add_idx_stride % 4)) return -3;
  if (a.gx_vv && ((uintptr_t)a.gx_vv % 16 || a.gx_vbs % 8)) return -3;
  if (npxl == 7 && (!a.gidx || !a.gx_d2 || !a.gx_w)) return -3;
  // chained second layer (p[12] = its float weights, p[13] = its epilogue descriptors, f[1] = its n_cob, f[2] = its k_pad): 16 x 16-row
  // samples, mode 0, every channel of this layer in one 64-channel tile, which is also the next layer's whole K
  a.ch_W = o.p[12]; a.ch_epi = (const SlideEpi *)o.p[13]; a.ch_n_cob = (int)o.f[1]; a.ch_k_pad = (int)o.f[2];
#ifdef SLIDE_TIMELINE
  if (!a.ch_W) { a.dbg = (unsigned long long *)o.p[13]; a.ch_epi = nullptr; }  // (tools/ab/op_timeline.py: stamps of an unchained launch)
#endif
  resolve_epi(a);
  if (a.ch_W && (npxl != 8 || a.gx_mode != 0 || a.n_cob > 2 || !a.ch_epi || a.ch_n_cob <= 0 || a.ch_k_pad != a.n_cob * 32 ||
                 (uintptr_t)a.ch_W % 16))
    return -3;
  return 0;
}

// SLIDE_OP_GEMM_GX with f[0] == 3: float tables, float row-major weights [n_cob*32][k_pad], float outputs (split arithmetic)
int slide_launch_gemm_gxs(const SlideOp &o, hipStream_t s) {
  GemmArgs a;
  const int st = gxs_args_from_op(o, a);
  if (st != 0) return st;
  const int npxl = o.i[4];
  const bool m1 = a.gx_mode != 0;
  if (a.ch_W) return launch_gxs<8, 0, true>(a, s);
  if (npxl == 8) return m1 ? launch_gxs<8, 1>(a, s) : launch_gxs<8, 0>(a, s);
  if (npxl == 7) return m1 ? launch_gxs<7, 1>(a, s) : launch_gxs<7, 0>(a, s);
  return -4;
}

namespace {
template <int NPXL, bool CHAIN = false>
int launch_gxs_dual(const GemmArgs &a1, const GemmArgs &a0, hipStream_t s) {
  constexpr int LDK = TileT<SLIDE_PREC_SPLIT>::LDK;
  constexpr int NSAMP = TM >> NPXL;
  auto lds = [&](const GemmArgs &a, int nvec) {
    return (size_t)(2 * TM + 3 * 64) * LDK * 2 + (2 * EPI_DW + (2 * EPI_DW) % 4 + 2 * 96) * 4 + (size_t)NSAMP * nvec * a.k_pad * 4 +
           (NPXL == 7 ? 3 * TM * 4 : 0) + 16;
  };
  const size_t s1 = lds(a1, 2 + (NPXL == 7 ? 2 : 0)), s0 = lds(a0, 1 + (NPXL == 7 ? 2 : 0));
  const size_t shm = s1 > s0 ? s1 : s0;
  if (shm > 80 * 1024) return -8;
  const int ntr = (a1.rows + TM - 1) / TM;
  const int g1 = ((ntr + 7) / 8) * 8 * ((a1.n_cob + 1) / 2), g0 = ((ntr + 7) / 8) * 8 * ((a0.n_cob + 1) / 2);
  static bool attr_done[GXS_MAX_DEVICES] = {};
  int d = 0;
  (void)hipGetDevice(&d);
  bool &attr_set = attr_done[d >= 0 && d < GXS_MAX_DEVICES ? d : 0];
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_gxs_dual_kernel<NPXL, CHAIN>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              80 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_gxs_dual_kernel<NPXL, CHAIN>), dim3(g1 + g0), dim3(256), shm, s, a1, a0, g1);
  return (int)hipGetLastError();
}
}  // namespace

// SLIDE_OP_GEMM_GX_DUAL whose two ops carry f[0] == 3 (mode 1 first, then mode 0; same rows and sample size)
int slide_launch_gemm_gxs_dual(const SlideOp *pr, hipStream_t s) {
  GemmArgs a1, a0;
  int st = gxs_args_from_op(pr[0], a1);
  if (st == 0) st = gxs_args_from_op(pr[1], a0);
  if (st != 0) return st;
  if (a1.gx_mode == 0 || a0.gx_mode != 0 || a1.rows != a0.rows || pr[0].i[4] != pr[1].i[4]) return -3;
  if (pr[0].i[4] == 8) st = a0.ch_W ? launch_gxs_dual<8, true>(a1, a0, s) : launch_gxs_dual<8>(a1, a0, s);
  else if (pr[0].i[4] == 7) st = launch_gxs_dual<7>(a1, a0, s);
  else return -4;
  if (st == -8) {  // (LDS of the dual form does not fit: two launches)
    st = slide_launch_gemm_gxs(pr[0], s);
    if (st == 0) st = slide_launch_gemm_gxs(pr[1], s);
  }
  return st;
}

// SLIDE_OP_ATTN_TAIL with f[1] bit 3: float rows, float row-major weights, float output (split arithmetic)
int slide_launch_attn_tail_split(const SlideOp &o, hipStream_t s) {
  TailSArgs a;
  a.X1 = (const float *)o.p[0]; a.W1 = (const float *)o.p[1]; a.X2 = (const float *)o.p[2]; a.W2 = (const float *)o.p[3];
  a.out = (float *)o.p[4]; a.vec = (const float *)o.p[5];
  a.out2 = (float *)o.p[7]; a.out2_ld = (int)o.f[2]; a.out2_n = (int)o.f[3];
  a.rows = o.i[0]; a.x1_ld = o.i[1]; a.k1 = o.i[2]; a.x2_ld = o.i[3]; a.k2 = o.i[4]; a.n_cob = o.i[5];
  a.gs = o.i[7]; a.n_norm = o.i[8]; a.out_ld = o.i[9];
  a.inv_count = o.f[0];
  const int npxl = o.i[6];
  if (a.k1 % 32 || a.k2 % 32 || a.k1 <= 0 || a.k2 <= 0 || a.rows <= 0 || a.n_cob <= 0 || a.x1_ld % 4 || a.x2_ld % 4) return -3;
  if ((uintptr_t)a.X1 % 16 || (uintptr_t)a.X2 % 16 || (uintptr_t)a.W1 % 16 || (uintptr_t)a.W2 % 16 || !a.out || !a.vec) return -3;
  constexpr int LDK = TileT<SLIDE_PREC_SPLIT>::LDK;
  const size_t shm = (size_t)(2 * TM + 3 * 64) * LDK * 2 + 4 * 2 * 32 * 4 + 64;
  const int ntc = (a.n_cob + 1) / 2, ntr = (a.rows + TM - 1) / TM;
  const int grid = ((ntr + 7) / 8) * 8 * ntc;
  if (npxl == 8) hipLaunchKernelGGL(attn_tail_split_kernel<8>, dim3(grid), dim3(256), shm, s, a);
  else if (npxl == 7) hipLaunchKernelGGL(attn_tail_split_kernel<7>, dim3(grid), dim3(256), shm, s, a);
  else return -4;
  return (int)hipGetLastError();
}

// SLIDE_OP_PP_STAGE: p[0] = HOST pointer to {int32 n, B; then n records of 16 x 8-byte slots}:
//   dense: slot 0 = 0, [1] X, [2] Wt (K-major float [k_pad][n_cob*32]), [3] epi, [4] in_scale, [5] in_shift, [6] x_ld, [7] k_pad, [8] n_cob, [9] in_bs
//   pair : slot 0 = 1, [1] y, [2] xyz, [3] wa, [4] wb, [5] epi, [6] ta, [7] tb, [8] nbr, [9] d2, [10] w, [11] vv_in, [12] vv_out, [13] SlideGnFin*,
//          [14] ld, [15] K
#ifndef SLIDE_EXPERIMENTS
int slide_launch_pp_stage(const SlideOp &, hipStream_t) { return -20; }  // experiments build only
#else
int slide_launch_pp_stage(const SlideOp &o, hipStream_t s) {
  const int64_t *h = (const int64_t *)o.p[0];
  if (!h) return -3;
  PPArgs a;
  a.n = (int)h[0]; a.B = (int)h[1];
  if (a.n <= 0 || a.n > PP_MAX || a.B <= 0) return -3;
  int nd = 0, np = 0;
  bool fp = false;
  for (int i = 0; i < a.n; ++i) {
    const int64_t *r = h + 2 + 16 * i;
    if (r[0] == 0) {
      if (nd >= PP_MAX) return -3;
      PPDense &d = a.d[nd];
      d.X = (const float *)r[1]; d.Wt = (const float *)r[2]; d.epi = (const SlideEpi *)r[3];
      d.in_scale = (const float *)r[4]; d.in_shift = (const float *)r[5];
      d.x_ld = (int)r[6]; d.k_pad = (int)r[7]; d.n_cob = (int)r[8]; d.in_bs = (int)r[9];
      if (!d.X || !d.Wt || !d.epi || d.k_pad <= 0 || d.k_pad > PP_KMAX || d.n_cob <= 0) return -3;
      a.kind[i] = 0; a.idx[i] = nd++;
    } else {
      if (np >= 2) return -3;
      PPPair &q = a.p[np];
      q.y = (const float *)r[1]; q.xyz = (const float *)r[2]; q.wa = (const float *)r[3]; q.wb = (const float *)r[4];
      q.epi = (const SlideEpi *)r[5]; q.ta = (float *)r[6]; q.tb = (float *)r[7]; q.nbr = (const int *)r[8];
      q.d2t = (const float *)r[9]; q.wt = (const float *)r[10]; q.vv_in = (const float *)r[11]; q.vv_out = (float *)r[12];
      q.fin = (const SlideGnFin *)r[13]; q.ld = (int)r[14]; q.K = (int)r[15];
      if (!q.y || !q.xyz || !q.ta || !q.tb || q.ld <= 0 || q.ld % 32 || q.ld > 2048 || (q.K != 8 && q.K != 16)) return -3;
      if (q.K == 8 && (!q.nbr || !q.d2t || !q.wt || !q.vv_in || !q.vv_out)) return -3;
      fp |= q.K == 8;
      a.kind[i] = 1; a.idx[i] = np++;
    }
  }
  const size_t shm = fp ? (size_t)16 * (PP_NT + 1) * 4 : 0;
  hipLaunchKernelGGL(pp_stage_kernel, dim3(a.B), dim3(PP_NT), shm, s, a);
  return (int)hipGetLastError();
}
#endif
