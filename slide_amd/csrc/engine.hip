// engine.hip -- fused latent-DDPM denoiser kernels for gfx950 (C-ABI: include/slide_engine.h).
//
// Data layout: every activation is a channel-minor fp32 matrix [B*npx][ld] in HBM (npx = 256 for the
// set-abstraction blocks: 16 points x 16 neighbours; 128 for the kNN-feature-propagation blocks; 16 for
// per-point tensors).  A 1x1 convolution is D[co][row] = sum_k W[co][k] X[row][k] on the matrix cores:
//   * fp32 mode: v_mfma_f32_32x32x2_f32  (bit-exact fp32 fma chain; parity mode)
//   * fp16 mode: v_mfma_f32_32x32x16_f16 (fp16 operands, fp32 accumulate; throughput mode)
// W is the A operand (rows = output channels), X the B operand (columns = points), so each lane ends up
// with 4 consecutive channels of one point -> 16-byte channel-minor stores, and the 16 neighbours of a point
// sit in 16 adjacent lanes.  One workgroup (4 waves, 64-wide) owns 256 rows = whole samples, so the
// GroupNorm statistics of a sample never leave the workgroup: bias, ReLU, GroupNorm, t-embedding /
// class-embedding add and the residual are all applied in the epilogue.
#include "gemm_common.h"
#include "gemm_small.h"
#include "ddpm_update.h"

#include <cstdlib>

// gemm_xs.hip: X-stationary kernel (-8: the X tile does not fit the LDS, -4: no such instantiation)
int slide_launch_rows_op(const SlideOp &o, hipStream_t s);  // rows_ops.hip
#ifdef SLIDE_EXPERIMENTS
int slide_launch_gemm_xs(const GemmArgs &a, int npxl, int cbw, bool aff, bool gat, int want_occ, hipStream_t s);  // gemm_xs.hip
#endif
// gemm_gx.hip: generated-X GEMM and the per-point table normalisation of the pair decomposition
int slide_launch_gemm_gx(const SlideOp &o, hipStream_t s);
int slide_launch_gemm_gxs(const SlideOp &o, hipStream_t s);  // gemm_gxs.hip (split arithmetic, float tables)
int slide_launch_attn_tail_split(const SlideOp &o, hipStream_t s);  // gemm_gxs.hip
int slide_launch_point_chain(const SlideOp &o, hipStream_t s);      // point_chain.hip
int slide_launch_pp_stage(const SlideOp &o, hipStream_t s);  // gemm_gxs.hip
int slide_launch_pair_norm(const SlideOp &o, hipStream_t s);
int slide_launch_sa_chain(const SlideOp &o, hipStream_t s);
int slide_launch_block_body(const SlideOp &o, hipStream_t s);  // experiments/block_body.hip
#ifdef SLIDE_EXPERIMENTS
int slide_launch_gemm_chain(const SlideOp &o, hipStream_t s);  // gemm_chain.hip
#endif
// Status of an op whose kernel only exists in the EXPERIMENTS build (slide_amd/build.py: libslide_hip_exp.so, -DSLIDE_EXPERIMENTS):
// the opt-in variants that lost their A/Bs (X-stationary tiles, per-point layer chains, head + update launch, wide / eight-wave
// attention tails, 128- / 32-channel and 64-deep ring tiles, the round-2 plan's gathered first layers, the register-staged fp16
// GEMM).  The product library carries only what a default plan dispatches.
#define SLIDE_ST_EXPERIMENT (-20)
int slide_launch_gemm_gx_dual(const SlideOp &o, hipStream_t s);  // gemm_gx.hip
int slide_launch_sa_chain_p(const SlideOp &o, hipStream_t s);    // gemm_gx.hip

namespace {

// (split mode: its two stages of four fp16 planes take 102 KB of LDS -- one workgroup per CU anyway, so it may use the whole
//  register file: the second accumulator set of the cross products does not fit 256 registers next to the 16-row epilogue)
template <int PREC, int NPXL, int CBW, bool PAIRRES = false>
__global__ __launch_bounds__(256, (PREC == SLIDE_PREC_SPLIT && NPXL == 4) ? 1 : 2) void gemm_kernel(GemmArgs a) {
  using T = typename TileT<PREC>::T;
  constexpr int LDK = TileT<PREC>::LDK;
  constexpr int EPL = TileT<PREC>::EPL;   // elements per 16-byte load
  constexpr int TPR = BK / EPL;           // threads per tile row
  constexpr int RPP = 256 / TPR;          // rows per pass
  constexpr int TN = 32 * CBW;
  constexpr int XP = TM / RPP, WP = TN / RPP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr bool SPLIT = PREC == SLIDE_PREC_SPLIT;
  // LDS element type of a stage: T, or (split mode) _Float16 with two planes per operand tile: [X hi | W hi | X lo | W lo]
  using TS = typename std::conditional<SPLIT, _Float16, T>::type;
  TS *const sbase = reinterpret_cast<TS *>(smem_raw);
  constexpr int STAGE = (SPLIT ? 2 : 1) * (TM + TN) * LDK;
  // split mode keeps ONE stage in LDS (51 KB: two workgroups per CU; the next chunk waits in registers, as in the other modes,
  // at the price of a second barrier per chunk)
  constexpr int NSTG = SPLIT ? 1 : 2;
  // consumer-side affine of the fp32 / split modes on 128- / 256-row samples: the tile's one or two samples' scale / shift vectors
  // are staged ONCE in LDS ([sample][scale | shift][k_pad] floats behind the epilogue tables) and applied when a chunk is written to
  // its stage -- loading them per X row (two more global loads per 16 bytes of X) made these launches 2x slower than their plain twins
  constexpr bool AFF_LDS = PREC != SLIDE_PREC_F16 && NPXL >= 7;
  constexpr int AFF_NS = TM >> (NPXL >= 7 ? NPXL : 7);

  const int ntc = (a.n_cob + CBW - 1) / CBW;
  const int ntr = (a.rows + TM - 1) / TM;
  // XCD-aware mapping: workgroup id % 8 picks the XCD (observed dispatch rule); all channel tiles of one
  // row tile share that XCD's L2, so the X panel is fetched from HBM once.
  const int xcd = blockIdx.x & 7, q0 = blockIdx.x >> 3;
  const int tc = q0 % ntc, tr = (q0 / ntc) * 8 + xcd;
  if (tr >= ntr) return;
  const int row0 = tr * TM, cob0 = tc * CBW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, col = lane & 31;
  const T *X = reinterpret_cast<const T *>(a.X);
  const T *W = reinterpret_cast<const T *>(a.W);

  f32x16 acc[CBW][2];
  f32x16 acc2[SPLIT ? CBW : 1][2];  // split mode: the two cross products hi*lo + lo*hi (scaled by 2^11), folded in at the end
#pragma unroll
  for (int i = 0; i < CBW; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[i][j][r] = 0.f;
        if (SPLIT) acc2[SPLIT ? i : 0][j][r] = 0.f;
      }

  float4 xr[XP], wr[WP];  // raw 16-byte pieces in flight
  const int l_row = tid / TPR, l_c = (tid % TPR) * EPL;

  auto load_chunk = [&](int kc) {
#pragma unroll
    for (int p = 0; p < XP; ++p) {
      const int grow = row0 + p * RPP + l_row;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (grow < a.rows) {
        v = *reinterpret_cast<const float4 *>(X + (size_t)grow * a.x_ld + kc * BK + l_c);
        if (a.in_scale && !AFF_LDS) {  // consumer-side GroupNorm affine (only the attention weight_conv.2 GEMMs)
          const size_t o = (size_t)(grow >> NPXL) * a.in_bs + kc * BK + l_c;
          if (PREC != SLIDE_PREC_F16) {
            const float4 sc = *reinterpret_cast<const float4 *>(a.in_scale + o);
            const float4 sh = *reinterpret_cast<const float4 *>(a.in_shift + o);
            v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
          } else {
            f16x8 h = *reinterpret_cast<f16x8 *>(&v);
#pragma unroll
            for (int j = 0; j < 8; ++j) h[j] = (_Float16)((float)h[j] * a.in_scale[o + j] + a.in_shift[o + j]);
            v = *reinterpret_cast<float4 *>(&h);
          }
        }
      }
      xr[p] = v;
    }
#pragma unroll
    for (int p = 0; p < WP; ++p) {
      const int gco = cob0 * 32 + p * RPP + l_row;
      wr[p] = gco < a.n_cob * 32 ? *reinterpret_cast<const float4 *>(W + (size_t)gco * a.k_pad + kc * BK + l_c)
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  // split mode: x = hi + 2^-11 lo, hi = fp16(x), lo = fp16(2^11 (x - hi)) -- the scaling keeps lo a NORMAL fp16 number
  // whatever the magnitude of x (unscaled, the low parts of values below ~0.1 would fall into fp16's denormal range)
  auto split4 = [](const float4 v, f16x4 &hi, f16x4 &lo) {
    hi = f16x4{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
    lo = f16x4{(_Float16)((v.x - (float)hi[0]) * 2048.f), (_Float16)((v.y - (float)hi[1]) * 2048.f),
               (_Float16)((v.z - (float)hi[2]) * 2048.f), (_Float16)((v.w - (float)hi[3]) * 2048.f)};
  };
  const float *aff_lds = nullptr;  // set below (behind the epilogue tables)
  auto store_chunk = [&](int s, int kc) {
    TS *Xs = sbase + s * STAGE;
    TS *Ws = Xs + TM * LDK;
    if constexpr (AFF_LDS) {
      if (a.in_scale) {
#pragma unroll
        for (int p = 0; p < XP; ++p) {
          const int trow = p * RPP + l_row;
          if (row0 + trow < a.rows) {
            const float *ap = aff_lds + (size_t)((trow >> NPXL) * 2) * a.k_pad + kc * BK + l_c;
            const float4 sc = *reinterpret_cast<const float4 *>(ap), sh = *reinterpret_cast<const float4 *>(ap + a.k_pad);
            float4 &v = xr[p];
            v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
          }
        }
      }
    }
    if constexpr (SPLIT) {
      _Float16 *Xl = Xs + (TM + TN) * LDK, *Wl = Xl + TM * LDK;
#pragma unroll
      for (int p = 0; p < XP; ++p) {
        f16x4 hi, lo;
        split4(xr[p], hi, lo);
        *reinterpret_cast<f16x4 *>(Xs + (p * RPP + l_row) * LDK + l_c) = hi;
        *reinterpret_cast<f16x4 *>(Xl + (p * RPP + l_row) * LDK + l_c) = lo;
      }
#pragma unroll
      for (int p = 0; p < WP; ++p) {
        f16x4 hi, lo;
        split4(wr[p], hi, lo);
        *reinterpret_cast<f16x4 *>(Ws + (p * RPP + l_row) * LDK + l_c) = hi;
        *reinterpret_cast<f16x4 *>(Wl + (p * RPP + l_row) * LDK + l_c) = lo;
      }
    } else {
#pragma unroll
      for (int p = 0; p < XP; ++p) *reinterpret_cast<float4 *>(Xs + (p * RPP + l_row) * LDK + l_c) = xr[p];
#pragma unroll
      for (int p = 0; p < WP; ++p) *reinterpret_cast<float4 *>(Ws + (p * RPP + l_row) * LDK + l_c) = wr[p];
    }
  };
  auto compute = [&](int s) {
    const TS *Xs = sbase + s * STAGE;
    const TS *Ws = Xs + TM * LDK;
    if constexpr (SPLIT) {
      const _Float16 *Xl = Xs + (TM + TN) * LDK, *Wl = Xl + TM * LDK;
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        f16x8 ah[CBW], al[CBW], bh[2], bl[2];
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb) {
          ah[cb] = *reinterpret_cast<const f16x8 *>(Ws + (cb * 32 + col) * LDK + st * 16 + half * 8);
          al[cb] = *reinterpret_cast<const f16x8 *>(Wl + (cb * 32 + col) * LDK + st * 16 + half * 8);
        }
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          bh[rb] = *reinterpret_cast<const f16x8 *>(Xs + (wave * 64 + rb * 32 + col) * LDK + st * 16 + half * 8);
          bl[rb] = *reinterpret_cast<const f16x8 *>(Xl + (wave * 64 + rb * 32 + col) * LDK + st * 16 + half * 8);
        }
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
          for (int rb = 0; rb < 2; ++rb) {
            acc[cb][rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cb], bh[rb], acc[cb][rb], 0, 0, 0);
            acc2[SPLIT ? cb : 0][rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cb], bl[rb], acc2[SPLIT ? cb : 0][rb], 0, 0, 0);
            acc2[SPLIT ? cb : 0][rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cb], bh[rb], acc2[SPLIT ? cb : 0][rb], 0, 0, 0);
          }
      }
    } else if (PREC == SLIDE_PREC_F32) {
      const float *Xf = reinterpret_cast<const float *>(Xs);
      const float *Wf = reinterpret_cast<const float *>(Ws);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 af[CBW], bf[2];
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb)
          af[cb] = *reinterpret_cast<const float4 *>(Wf + (cb * 32 + col) * LDK + q * 8 + half * 4);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
          bf[rb] = *reinterpret_cast<const float4 *>(Xf + (wave * 64 + rb * 32 + col) * LDK + q * 8 + half * 4);
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
          for (int rb = 0; rb < 2; ++rb) {
            acc[cb][rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cb].x, bf[rb].x, acc[cb][rb], 0, 0, 0);
            acc[cb][rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cb].y, bf[rb].y, acc[cb][rb], 0, 0, 0);
            acc[cb][rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cb].z, bf[rb].z, acc[cb][rb], 0, 0, 0);
            acc[cb][rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cb].w, bf[rb].w, acc[cb][rb], 0, 0, 0);
          }
      }
    } else {
      const _Float16 *Xh = reinterpret_cast<const _Float16 *>(Xs);
      const _Float16 *Wh = reinterpret_cast<const _Float16 *>(Ws);
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        f16x8 af[CBW], bf[2];
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb)
          af[cb] = *reinterpret_cast<const f16x8 *>(Wh + (cb * 32 + col) * LDK + st * 16 + half * 8);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
          bf[rb] = *reinterpret_cast<const f16x8 *>(Xh + (wave * 64 + rb * 32 + col) * LDK + st * 16 + half * 8);
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
          for (int rb = 0; rb < 2; ++rb)
            acc[cb][rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[cb], bf[rb], acc[cb][rb], 0, 0, 0);
      }
    }
  };

  uint32_t *const epi_lds = reinterpret_cast<uint32_t *>(smem_raw + NSTG * (size_t)STAGE * sizeof(TS));
  float *const vec_lds = reinterpret_cast<float *>(epi_lds + CBW * EPI_DW + (CBW * EPI_DW) % 4);
  stage_epilogue_tables<CBW>(a, cob0, tid, epi_lds, vec_lds);
  if constexpr (AFF_LDS) {
    if (a.in_scale) {
      float *al = vec_lds + CBW * 96;
      const int nb = a.rows >> NPXL;
      for (int i = tid * 4; i < AFF_NS * a.k_pad; i += 1024) {
        const int sm = i / a.k_pad, k = i - sm * a.k_pad;
        int b = (row0 >> NPXL) + sm;
        b = b < nb ? b : nb - 1;
        *reinterpret_cast<float4 *>(al + (size_t)(sm * 2 + 0) * a.k_pad + k) = *reinterpret_cast<const float4 *>(a.in_scale + (size_t)b * a.in_bs + k);
        *reinterpret_cast<float4 *>(al + (size_t)(sm * 2 + 1) * a.k_pad + k) = *reinterpret_cast<const float4 *>(a.in_shift + (size_t)b * a.in_bs + k);
      }
      aff_lds = al;
      __syncthreads();
    }
  }

  const int nk = a.k_pad / BK;
#ifdef SLIDE_STAGGER
  const int koff = (tc * 5 + tr * 3) % nk;
#define KIDX(k) (((k) + koff) % nk)
#else
#define KIDX(k) (k)
#endif
  load_chunk(KIDX(0));
  store_chunk(0, KIDX(0));
  __syncthreads();
  for (int kc = 0; kc < nk; ++kc) {
    if (kc + 1 < nk) load_chunk(KIDX(kc + 1));
    compute(kc & (NSTG - 1));
    if (NSTG == 1) __syncthreads();  // every wave is done reading the stage before it is overwritten
    if (kc + 1 < nk) store_chunk((kc + 1) & (NSTG - 1), KIDX(kc + 1));
    __syncthreads();
  }
#undef KIDX
  if constexpr (SPLIT) {
#pragma unroll
    for (int i = 0; i < CBW; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = fmaf(acc2[i][j][r], 1.f / 2048.f, acc[i][j][r]);
  }
  // (split mode stores float activations: the fp32 epilogue)
  gemm_epilogue<SPLIT ? SLIDE_PREC_F32 : PREC, NPXL, CBW, 2, PAIRRES>(a, acc, row0, cob0, wave, half, col, epi_lds, vec_lds,
                                                                      reinterpret_cast<float *>(smem_raw));
}

// ------------------------------------------------------------------------------------------------ small split GEMM
// Split-mode GEMM for SMALL launches (the 16-row per-point layers of the fp32-structured plan; the training step's layers at the
// reference's batch 32): tile 64 rows x 64 channels, wave (rb, cb) owns ONE 32x32 block over the whole K.  The 256-row tile of
// gemm_kernel leaves such a launch with a handful of workgroups (1408 rows x 512 channels: 48) whose cost is their own serial
// latency -- here the grid is 4x larger and a workgroup's K loop moves a quarter of the X rows per chunk.  Stage = four fp16
// planes [X hi | W hi | X lo | W lo] of 64 rows (20 KB), double-buffered; three workgroups per CU.  Same split arithmetic
// (hi*hi into one accumulator, hi*lo + lo*hi scaled by 2^11 into a second) and the common epilogue at one block per wave.
template <int NPXL>
__global__ __launch_bounds__(256, 3) void gemm_split_small_kernel(GemmArgs a) {
  constexpr int LDK = TileT<SLIDE_PREC_SPLIT>::LDK;
  constexpr int TR = 64;                       // tile rows = tile channels
  constexpr int PLANE = TR * LDK;              // halves per plane
  constexpr int STAGE = 4 * PLANE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  _Float16 *const sbase = reinterpret_cast<_Float16 *>(smem_raw);
  const int ntc = (a.n_cob + 1) / 2;
  const int tc = blockIdx.x % ntc, tr = blockIdx.x / ntc;
  const int row0 = tr * TR, cob0 = tc * 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, col = lane & 31;
  const int rb = wave >> 1, cb = wave & 1;
  const float *X = reinterpret_cast<const float *>(a.X);
  const float *W = reinterpret_cast<const float *>(a.W);
  f32x16 acc, acc2;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = acc2[r] = 0.f;
  // loads: 8 threads per 32-float row, 32 rows per pass, two passes per operand tile
  const int l_row = tid >> 3, l_c = (tid & 7) * 4;
  constexpr int PD = 3;  // chunks in flight in registers: a lone workgroup's global-load latency hides behind three K steps
  float4 xr[PD][2], wr[PD][2];
  auto load_chunk = [&](int kc, float4 (&xr)[2], float4 (&wr)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int grow = row0 + p * 32 + l_row;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (grow < a.rows) {
        v = *reinterpret_cast<const float4 *>(X + (size_t)grow * a.x_ld + kc * BK + l_c);
      }
      xr[p] = v;
      const int gco = cob0 * 32 + p * 32 + l_row;
      wr[p] = gco < a.n_cob * 32 ? *reinterpret_cast<const float4 *>(W + (size_t)gco * a.k_pad + kc * BK + l_c)
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto split4 = [](const float4 v, f16x4 &hi, f16x4 &lo) {
    hi = f16x4{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
    lo = f16x4{(_Float16)((v.x - (float)hi[0]) * 2048.f), (_Float16)((v.y - (float)hi[1]) * 2048.f),
               (_Float16)((v.z - (float)hi[2]) * 2048.f), (_Float16)((v.w - (float)hi[3]) * 2048.f)};
  };
  const float *aff_lds = nullptr;  // consumer-side affine: the tile's samples' [scale | shift][k_pad], staged once (set below)
  auto store_chunk = [&](int s, int kc, const float4 (&xr)[2], const float4 (&wr)[2]) __attribute__((always_inline)) {
    _Float16 *Xh = sbase + s * STAGE, *Wh = Xh + PLANE, *Xl = Wh + PLANE, *Wl = Xl + PLANE;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      f16x4 hi, lo;
      float4 v = xr[p];
      if (aff_lds && row0 + p * 32 + l_row < a.rows) {
        const float *ap = aff_lds + (size_t)(((p * 32 + l_row) >> NPXL) * 2) * a.k_pad + kc * BK + l_c;
        const float4 sc = *reinterpret_cast<const float4 *>(ap), sh = *reinterpret_cast<const float4 *>(ap + a.k_pad);
        v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
      }
      split4(v, hi, lo);
      *reinterpret_cast<f16x4 *>(Xh + (p * 32 + l_row) * LDK + l_c) = hi;
      *reinterpret_cast<f16x4 *>(Xl + (p * 32 + l_row) * LDK + l_c) = lo;
      split4(wr[p], hi, lo);
      *reinterpret_cast<f16x4 *>(Wh + (p * 32 + l_row) * LDK + l_c) = hi;
      *reinterpret_cast<f16x4 *>(Wl + (p * 32 + l_row) * LDK + l_c) = lo;
    }
  };
  uint32_t *const epi_lds = reinterpret_cast<uint32_t *>(smem_raw + 2 * (size_t)STAGE * sizeof(_Float16));
  float *const vec_lds = reinterpret_cast<float *>(epi_lds + 2 * EPI_DW + (2 * EPI_DW) % 4);
  SLIDE_STAMP(a, 0);
  stage_epilogue_tables<2>(a, cob0, tid, epi_lds, vec_lds);
  if (a.in_scale) {
    float *al = vec_lds + 2 * 96;
    const int nb = a.rows >> NPXL;
    for (int i = tid * 4; i < (TR >> NPXL) * a.k_pad; i += 1024) {
      const int sm = i / a.k_pad, k = i - sm * a.k_pad;
      int b = (row0 >> NPXL) + sm;
      b = b < nb ? b : nb - 1;
      *reinterpret_cast<float4 *>(al + (size_t)(sm * 2 + 0) * a.k_pad + k) = *reinterpret_cast<const float4 *>(a.in_scale + (size_t)b * a.in_bs + k);
      *reinterpret_cast<float4 *>(al + (size_t)(sm * 2 + 1) * a.k_pad + k) = *reinterpret_cast<const float4 *>(a.in_shift + (size_t)b * a.in_bs + k);
    }
    aff_lds = al;
    __syncthreads();
  }
  const int nk = a.k_pad / BK;
  load_chunk(0, xr[0], wr[0]);
  store_chunk(0, 0, xr[0], wr[0]);
#pragma unroll
  for (int j = 0; j < PD; ++j)
    if (j + 1 < nk) load_chunk(j + 1, xr[j], wr[j]);  // buffer j: chunks j + 1, j + 1 + PD, ...
  __syncthreads();
  SLIDE_STAMP(a, 1);
  for (int kc0 = 0; kc0 < nk; kc0 += PD) {
#pragma unroll
    for (int j = 0; j < PD; ++j) {
      const int kc = kc0 + j;
      if (kc >= nk) break;
      const _Float16 *Xh = sbase + (kc & 1) * STAGE, *Wh = Xh + PLANE, *Xl = Wh + PLANE, *Wl = Xl + PLANE;
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const f16x8 ah = *reinterpret_cast<const f16x8 *>(Wh + (cb * 32 + col) * LDK + st * 16 + half * 8);
        const f16x8 al = *reinterpret_cast<const f16x8 *>(Wl + (cb * 32 + col) * LDK + st * 16 + half * 8);
        const f16x8 bh = *reinterpret_cast<const f16x8 *>(Xh + (rb * 32 + col) * LDK + st * 16 + half * 8);
        const f16x8 bl = *reinterpret_cast<const f16x8 *>(Xl + (rb * 32 + col) * LDK + st * 16 + half * 8);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc2, 0, 0, 0);
      }
      // the other stage: its last readers passed the barrier of chunk kc - 1
      if (kc + 1 < nk) store_chunk((kc + 1) & 1, kc + 1, xr[j], wr[j]);
      __syncthreads();
      if (kc + 1 + PD < nk) load_chunk(kc + 1 + PD, xr[j], wr[j]);
    }
  }
  f32x16 one[1][1];
#pragma unroll
  for (int r = 0; r < 16; ++r) one[0][0][r] = fmaf(acc2[r], 1.f / 2048.f, acc[r]);
  SLIDE_STAMP(a, 2);
  gemm_epilogue<SLIDE_PREC_F32, NPXL, 1, 1>(a, one, row0 + rb * 32, cob0 + cb, 0, half, col, epi_lds + cb * EPI_DW, vec_lds + cb * 96,
                                            nullptr);
  SLIDE_STAMP(a, 5);
#ifdef SLIDE_TIMELINE
  if (a.dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); SLIDE_STAMP(a, 6); }
#endif
}

// ------------------------------------------------------------------------------------------------ LDS-DMA GEMM
// fp16 throughput variant of gemm_kernel (no consumer-side affine): the X / W chunks go HBM/L2 -> LDS directly with
// `global_load_lds_dwordx4` (no VGPR staging), NST chunks deep, so many more bytes are in flight per CU than a
// register-staged prefetch allows.  An LDS-DMA instruction writes lane-linearly (base + 16 B x lane), so a stage is an
// unpadded [rows][32] fp16 image (64 B rows) and bank conflicts are avoided by swizzling on the SOURCE side: slot
// (lane & 3) of row r receives the 16-byte piece p = slot ^ ((r >> 2) & 3); fragment reads apply the same XOR
// (conflict-free for ds_read_b128's 16-lane groups).  One raw s_barrier per chunk; counted vmcnt keeps NST-2 chunks in
// flight across it.
// AFF: consumer-side GroupNorm affine (attention weight_conv.2): the per-(sample, channel) scale / shift vectors of the
// workgroup's samples are staged once in LDS (fp16) and applied in fp32 to the X fragments between LDS and MFMA.
// WC = 1: four waves, tile 256 rows x 32*CBW channels.  WC = 2: eight waves, tile 256 rows x 64*CBW channels -- wave
// (wr, wc) owns rows 64*wr.. and channel half wc, so the X chunk is fetched once per 64*CBW channels (less L2 -> LDS
// traffic per MAC, half as many prologues); each channel half runs the 4-wave epilogue on its own LDS tables.
template <int NPXL, int CBW, int NST, int BKT, bool AFF, int WC = 1, bool GAT = false, bool PAIRRES = false, bool ATTN = false>
__device__ __forceinline__ void glds_tile(const GemmArgs &a, unsigned char *smem_raw, const int tr, const int tc) {
  using T = _Float16;
  constexpr int NW = 4 * WC, NT = 256 * WC;  // waves, threads
  constexpr int TN = 32 * CBW * WC;
  // W rows staged per chunk: TN rounded up until the stage splits into whole DMA instructions per wave
  constexpr int TNS = ((TM + TN + 16 * NW - 1) / (16 * NW)) * (16 * NW) - TM;
  constexpr int RT = TM + TNS;             // tile rows per stage (X rows then W rows)
  constexpr int ROWB = BKT * 2;            // bytes per tile row (64 or 128 = one full cache line)
  constexpr int PPR = ROWB / 16;           // 16-byte pieces per row (4 or 8)
  constexpr int RPI = 64 / PPR;            // rows per LDS-DMA instruction (16 or 8)
  constexpr int NI = RT / RPI;             // LDS-DMA instructions per stage
  constexpr int LPW = NI / NW;             // per wave
  constexpr int STAGE_B = RT * ROWB;       // bytes
  constexpr int SWS = BKT == 32 ? 2 : 1;   // swizzle: slot = piece ^ ((row >> SWS) & (PPR - 1))
  static_assert(NI % NW == 0, "tile rows must split evenly over the waves");
  static_assert(!AFF || WC == 1, "the affine variant is four-wave only");

  const int row0 = tr * TM, cob0 = tc * CBW * WC;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wave = wv & 3, wc = wv >> 2;
  const int half = lane >> 5, col = lane & 31;

  SLIDE_STAMP(a, 0);
  // per-lane source pointers of this wave's LPW instructions (chunk 0); out-of-range rows are clamped to a valid row:
  // they only feed accumulator rows / channel blocks that are never stored
  // GAT: the grouped input is never materialised -- the feature columns of row (sample, point, neighbour) are DMA-read
  // straight from the neighbour's row of the point-feature table (per-lane source addresses are free), only the last
  // chunks (coordinate channels) come from a small assembled buffer.  ga[j] serves chunks < g_nsplit, gp[j] the rest.
  // CHUNK-MAJOR operands (BKT == 32): X [k / 32][rows][32] when x_ld == 32, W [k / 32][n_cob * 32][32] when a.w_cm -- the
  // chunk kc of a row is then kc * (rows x 32) elements further instead of kc * 32, and the 16 rows of one DMA instruction
  // are consecutive memory.  x_cs / w_cs: elements between consecutive chunks of one row.
  constexpr int NXI = TM / (RPI * NW);  // instructions j < NXI carry X rows, the rest W rows
  static_assert(TM % (RPI * NW) == 0, "X rows must fill whole DMA instructions");
  const size_t x_cs = (BKT == 32 && a.x_ld == 32) ? (size_t)a.rows * 32 : BKT;
  const size_t w_cs = (BKT == 32 && a.w_cm) ? (size_t)a.n_cob * 32 * 32 : BKT;
  const int w_ld = (BKT == 32 && a.w_cm) ? 32 : a.k_pad;
  // gathered point-feature table: chunk-major [k / 32][samples * 16][32] when g_ldf == 32 -- the sixteen 64-byte pieces an
  // instruction gathers (the neighbours of one point) then lie inside ONE KB instead of sixteen rows
  const size_t g_cs = (GAT && BKT == 32 && a.g_ldf == 32) ? (size_t)(a.rows >> NPXL) * 16 * 32 : BKT;
  const T *gp[LPW];
  const T *ga[GAT ? LPW : 1];
#pragma unroll
  for (int j = 0; j < LPW; ++j) {
    const int trow = RPI * (j * NW + wv) + lane / PPR;
    const int piece = (lane % PPR) ^ ((trow >> SWS) & (PPR - 1));
    if (trow < TM) {
      int grow = row0 + trow;
      grow = grow < a.rows ? grow : a.rows - 1;
      gp[j] = reinterpret_cast<const T *>(a.X) + (size_t)grow * a.x_ld + piece * 8;
      if (GAT) {
        const int smp = grow >> NPXL, pxl = grow & ((1 << NPXL) - 1);
        const int nb = a.gidx[(smp * 16 + (pxl >> a.g_klog2)) * 16 + (pxl & ((1 << a.g_klog2) - 1))];
        ga[j] = reinterpret_cast<const T *>(a.gfeat) + (size_t)(smp * 16 + nb) * a.g_ldf + piece * 8;
        gp[j] -= (size_t)a.g_nsplit * x_cs;  // chunk index kc keeps counting over the whole K
      }
    } else {
      int gco = cob0 * 32 + (trow - TM);
      gco = gco < a.n_cob * 32 ? gco : a.n_cob * 32 - 1;
      gp[j] = reinterpret_cast<const T *>(a.W) + (size_t)gco * w_ld + piece * 8;
      if (GAT) ga[j] = gp[j];
    }
  }
  auto issue = [&](int kc, int st) {
#pragma unroll
    for (int j = 0; j < LPW; ++j) {
      const T *src = (j < NXI) ? ((GAT && kc < a.g_nsplit) ? ga[GAT ? j : 0] + (size_t)kc * g_cs : gp[j] + (size_t)kc * x_cs)
                               : gp[j] + (size_t)kc * w_cs;
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)src,
                                       (__attribute__((address_space(3))) void *)(smem_raw + (size_t)st * STAGE_B +
                                                                                  (j * NW + wv) * 1024),
                                       16, 0, 0);
    }
  };

  // the ring is primed BEFORE the epilogue tables are staged: the first chunks' L2 latency covers the table reads
  const int nk = a.k_pad / BKT;
#pragma unroll
  for (int s0 = 0; s0 < NST - 1; ++s0)
    if (s0 < nk) issue(s0, s0);
  uint32_t *const epi_lds = reinterpret_cast<uint32_t *>(smem_raw + (size_t)NST * STAGE_B);
  float *const vec_lds = reinterpret_cast<float *>(epi_lds + CBW * WC * EPI_DW + (CBW * WC * EPI_DW) % 4);
  stage_epilogue_tables<CBW * WC, NT>(a, cob0, tid, epi_lds, vec_lds);
  constexpr int NSAMP = (1 << NPXL) >= TM ? 1 : TM >> NPXL;  // samples per workgroup
  _Float16 *const aff_lds = reinterpret_cast<_Float16 *>(vec_lds + CBW * WC * 96);  // [sample][scale | shift | add][k_pad]
  if (AFF) {
    static_assert(!AFF || NPXL >= 6, "the affine variant assumes one sample per wave");
    const int tps = a.aff_tps > 1 ? a.aff_tps : 1;
    const int nb = (a.rows >> NPXL) / tps, n_aff = NSAMP * a.k_pad;
    for (int i0 = tid; i0 < n_aff; i0 += 1024) {  // four elements per trip, their loads issued together
      float sc4[4], sh4[4], ad4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + 256 * u < n_aff ? i0 + 256 * u : i0;
        const int sm = i / a.k_pad, k = i - sm * a.k_pad;
        int b = ((row0 >> NPXL) + sm) / tps;
        b = b < nb ? b : nb - 1;
        sc4[u] = a.in_scale[(size_t)b * a.in_bs + k];
        sh4[u] = a.in_shift[(size_t)b * a.in_bs + k];
        ad4[u] = (a.in_add && k < a.add_n) ? a.in_add[(size_t)b * a.add_bs + k] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + 256 * u;
        if (i >= n_aff) break;
        const int sm = i / a.k_pad, k = i - sm * a.k_pad;
        aff_lds[(sm * 3 + 0) * a.k_pad + k] = (_Float16)sc4[u];
        aff_lds[(sm * 3 + 1) * a.k_pad + k] = (_Float16)sh4[u];
        aff_lds[(sm * 3 + 2) * a.k_pad + k] = (_Float16)ad4[u];
      }
    }
  }
  const _Float16 *const aff_w = aff_lds + (size_t)((wave * 64) >> NPXL) * 3 * a.k_pad;  // this wave's sample
  const bool aff_relu = AFF && a.aff_relu;

  f32x16 acc[CBW][2];
#pragma unroll
  for (int i = 0; i < CBW; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // row base offsets and swizzle keys of this lane's fragment rows (bytes inside a stage)
  int wrow[CBW], wkey[CBW], xrow[2], xkey[2];
#pragma unroll
  for (int cb = 0; cb < CBW; ++cb) {
    const int trow = TM + (wc * CBW + cb) * 32 + col;
    wrow[cb] = trow * ROWB; wkey[cb] = (trow >> SWS) & (PPR - 1);
  }
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const int trow = wave * 64 + rb * 32 + col;
    xrow[rb] = trow * ROWB; xkey[rb] = (trow >> SWS) & (PPR - 1);
  }

  SLIDE_STAMP(a, 1);
  for (int kc = 0; kc < nk; ++kc) {
    // chunk kc must have landed; up to NST-2 younger chunks may stay in flight (fewer in the tail -> drain)
    if (kc + NST - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * LPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kc + NST - 1 < nk) issue(kc + NST - 1, (kc + NST - 1) % NST);  // overwrites the stage consumed at kc-1
    const unsigned char *sb = smem_raw + (size_t)(kc % NST) * STAGE_B;
#pragma unroll
    for (int st2 = 0; st2 < BKT / 16; ++st2) {
      f16x8 af[CBW], bf[2];
      const int piece = st2 * 2 + half;
#pragma unroll
      for (int cb = 0; cb < CBW; ++cb) af[cb] = *reinterpret_cast<const f16x8 *>(sb + wrow[cb] + ((piece ^ wkey[cb]) << 4));
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) bf[rb] = *reinterpret_cast<const f16x8 *>(sb + xrow[rb] + ((piece ^ xkey[rb]) << 4));
      if (AFF) {
        const f16x8 sc = *reinterpret_cast<const f16x8 *>(aff_w + kc * BKT + piece * 8);
        const f16x8 sh = *reinterpret_cast<const f16x8 *>(aff_w + a.k_pad + kc * BKT + piece * 8);
        // packed fp16 fma (v_pk_fma_f16, one rounding like the fp32-then-convert form it replaces, 1/6 of the VALU ops).
        // scale / shift / add are fp16 copies of the fp32 vectors (2^-11 relative each): against normalising in fp32 and
        // storing the fp16 result (SLIDE_MODULE_DEFER=0) the GEMM output moves by <= 3e-3 of its L2 norm at |shift| ~ 6 and
        // |add| ~ 50 (tests/test_hip_modules.py::test_deferred_normalisation_matches_the_materialised_path); values beyond
        // fp16's range (65504) do not occur: scale = gamma * rstd <= gamma / sqrt(eps), shift and add are O(activations)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) bf[rb] = __builtin_elementwise_fma(bf[rb], sc, sh);
        if (aff_relu) {  // deferred GroupNorm + ReLU + embedding add of the producing layer (module-level path)
          const f16x8 ad = *reinterpret_cast<const f16x8 *>(aff_w + 2 * a.k_pad + kc * BKT + piece * 8);
          const f16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
          for (int rb = 0; rb < 2; ++rb) bf[rb] = __builtin_elementwise_max(bf[rb], zero) + ad;
        }
      }
#pragma unroll
      for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
          acc[cb][rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[cb], bf[rb], acc[cb][rb], 0, 0, 0);
    }
  }
  __syncthreads();  // every wave is done with the tiles before `red` reuses them
  SLIDE_STAMP(a, 2);

  if constexpr (ATTN)
    attend_epilogue<CBW>(a, acc, row0, cob0, wave, half, col, vec_lds, reinterpret_cast<float *>(smem_raw));
  else
    gemm_epilogue<SLIDE_PREC_F16, NPXL, CBW, 2, PAIRRES>(a, acc, row0, cob0 + wc * CBW, wave, half, col, epi_lds + wc * CBW * EPI_DW,
                                             vec_lds + wc * CBW * 96,
                                             reinterpret_cast<float *>(smem_raw) + wc * (256 + 128) * CBW);
  SLIDE_STAMP(a, 5);
#ifdef SLIDE_TIMELINE
  if (a.dbg) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SLIDE_STAMP(a, 6);
  }
#endif
}

// Scheduler.  a.sched == nullptr: one tile per workgroup (grid = tiles).  Otherwise PERSISTENT: 2 workgroups per CU
// pull tiles from per-XCD counters (tile columns of one row tile stay on one XCD's L2), so workgroups drift out of
// phase instead of all bursting their loads, then all bursting their stores, and there is no last partial round.
// The workgroup in the odd wave slot of a CU starts `stagger` later so that the pair begins half a tile apart.
// sched[0..7] = next tile per XCD, sched[8] = finished workgroups; the last one to finish re-arms the counters.
template <int NPXL, int CBW, int NST, int BKT, bool AFF, bool GAT = false, bool PAIRRES = false>
__global__ __launch_bounds__(256, 2) void gemm_glds_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int ntc = (a.n_cob + CBW - 1) / CBW;
  const int ntr = (a.rows + TM - 1) / TM;
  const int xcd = blockIdx.x & 7;
  if (!a.sched) {
    const int q0 = blockIdx.x >> 3;
    const int tc = q0 % ntc, tr = (q0 / ntc) * 8 + xcd;
    if (tr >= ntr) return;
    glds_tile<NPXL, CBW, NST, BKT, AFF, 1, GAT, PAIRRES>(a, smem_raw, tr, tc);
    return;
  }
  const int my_tiles = ((ntr - xcd + 7) / 8) * ntc;  // row tiles tr = xcd, xcd + 8, ...
  volatile int *const s_tile = reinterpret_cast<volatile int *>(smem_raw + a.shm_bytes - 16);
  // a.stagger < 0: STATIC persistent schedule -- workgroup l of an XCD takes tiles l, l + n, l + 2n, ... (no counter):
  // the store drain and the relaunch of a workgroup are overlapped by its next tile's prologue
  const bool fixed = a.stagger < 0;
  int next = blockIdx.x >> 3;
  const int step = gridDim.x >> 3;
  if (a.stagger > 0 && threadIdx.x < 64) {
    const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);  // HW_ID.WAVE_ID
    if (slot & 1) {
      const unsigned long long t0 = wall_clock64();
      while (wall_clock64() - t0 < (unsigned long long)a.stagger) __builtin_amdgcn_s_sleep(32);
    }
  }
  for (;;) {
    int t = next;
    next += step;
    if (!fixed) {
      if (threadIdx.x == 0) *s_tile = atomicAdd(&a.sched[xcd], 1);
      __syncthreads();
      t = __builtin_amdgcn_readfirstlane(*s_tile);
    }
    if (t >= my_tiles) break;
#ifdef SLIDE_TIMELINE
    GemmArgs a2 = a;  // stamps indexed by tile instead of by workgroup
    if (a.dbg) a2.dbg = a.dbg + ((long long)(xcd + 8 * t) - (long long)blockIdx.x) * 16;
    glds_tile<NPXL, CBW, NST, BKT, AFF, 1, GAT, PAIRRES>(a2, smem_raw, (t / ntc) * 8 + xcd, t % ntc);
#else
    glds_tile<NPXL, CBW, NST, BKT, AFF, 1, GAT, PAIRRES>(a, smem_raw, (t / ntc) * 8 + xcd, t % ntc);
#endif
    __syncthreads();  // the epilogue's LDS reads are done before the next tile's tables / DMAs / s_tile land
  }
  if (!fixed && threadIdx.x == 0 && atomicAdd(&a.sched[8], 1) == (int)gridDim.x - 1) {
#pragma unroll
    for (int x = 0; x < 9; ++x) a.sched[x] = 0;
  }
}

// Three workgroups per CU: 64-channel tiles on a two-stage ring (41 KB of LDS) under a 168-VGPR budget -- one more
// resident workgroup to fill the epilogue / prologue bubbles of the other two (opt-in: GemmArgs.stagger == 3).
template <int NPXL, bool AFF, bool GAT, bool PAIRRES = false>
__global__ __launch_bounds__(256, 3) void gemm_glds_occ3_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int ntc = (a.n_cob + 1) / 2;
  const int ntr = (a.rows + TM - 1) / TM;
  const int xcd = blockIdx.x & 7, q0 = blockIdx.x >> 3;
  const int tc = q0 % ntc, tr = (q0 / ntc) * 8 + xcd;
  if (tr >= ntr) return;
  glds_tile<NPXL, 2, 2, 32, AFF, 1, GAT, PAIRRES>(a, smem_raw, tr, tc);
}

// SLIDE_OP_GEMM_ATTEND (round 6): the 256 x 64 ring tile with the ATTEND epilogue (gemm_common.h) -- the score GEMM of an
// AttentionModule of the module-level path, its soft-max over the neighbours and the weighted sum of the values in one launch
// (two-stage ring and 126 registers: three workgroups per CU, as gemm_glds_occ3_kernel -- the launch is HBM-bound)
template <bool AFF>
__global__ __launch_bounds__(256, 3) void gemm_attend_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int ntc = (a.n_cob + 1) / 2;
  const int ntr = (a.rows + TM - 1) / TM;
  const int xcd = blockIdx.x & 7, q0 = blockIdx.x >> 3;
  const int tc = q0 % ntc, tr = (q0 / ntc) * 8 + xcd;
  if (tr >= ntr) return;
  glds_tile<8, 2, 2, 32, AFF, 1, false, false, true>(a, smem_raw, tr, tc);
}

// eight-wave variant (one tile per workgroup, one workgroup per CU: its deeper ring needs the LDS of two)
template <int NPXL, int CBW, int NST>
__global__ __launch_bounds__(512, 2) void gemm_glds8_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int ntc = (a.n_cob + 2 * CBW - 1) / (2 * CBW);
  const int ntr = (a.rows + TM - 1) / TM;
  const int xcd = blockIdx.x & 7, q0 = blockIdx.x >> 3;
  const int tc = q0 % ntc, tr = (q0 / ntc) * 8 + xcd;
  if (tr >= ntr) return;
  glds_tile<NPXL, CBW, NST, 32, false, 2>(a, smem_raw, tr, tc);
}

template <int NST, bool AFF>
__global__ __launch_bounds__(256, 3) void gemm_small_kernel(GemmArgs a) {
  small_body<NST, AFF, 0>(a, PairArgs(), blockIdx.x);
}

// The per-point GEMM of a block's pair decomposition and the pair-table pass (SLIDE_OP_PAIR_NORM version 1) as ONE launch
// (SLIDE_OP_PAIR_FIRST): the products never go through memory.  FP: the 8-neighbour samples of the FP blocks.
template <bool FP>
__global__ __launch_bounds__(256, 3) void pair_first_kernel(GemmArgs a, PairArgs pa) {
  small_body<2, false, FP ? 2 : 1>(a, pa, blockIdx.x);
}

// ------------------------------------------------------------------------------------------------ attention tail
// The end of an AttentionModule (attention.py:90-95) as ONE launch: scores S = W5 . u + b5, values
// V = ReLU(GN(Wv . mo + bv)), out[point] = sum_k softmax_k(S) * V -- instead of two GEMMs that write S and V
// ([rows][C] each) and a third kernel that reads them back.  MFMA operands are SWAPPED with respect to gemm_glds_kernel
// (A = X rows, B = W rows): a lane then owns ONE channel and its registers run over rows, so GroupNorm statistics
// and the softmax over a point's K neighbours are register loops plus one exchange between the lane halves, and
// neither S nor V ever leaves the registers.  Tile: 256 rows x 64 channels, four waves x 64 rows; K = 2^(NPXL-4)
// neighbours per point (16 rows of a 256-row sample, 8 of a 128-row one).
struct AttnTailArgs {
  const void *X1, *W1, *X2, *W2;  // scores: u [rows][x1_ld] . W5 [C][k1];  values: mo [rows][x2_ld] . Wv [C][k2]
  const float *vec;               // [bias_s | bias_v | gamma | beta], n_cob * 32 floats each
  void *out;                      // [rows >> (NPXL - 4)][out_ld] fp16
  void *out_cm;                   // optional second copy, chunk-major [c / 32][rows >> (NPXL - 4)][32]
  void *out2;                     // optional copy of the first out2_n channels into another per-point buffer [..][out2_ld]
  int out2_ld, out2_n;
  int rows, x1_ld, k1, x2_ld, k2, n_cob, gs, n_norm, out_ld;
  unsigned long long *dbg;        // optional per-workgroup timeline (instrumented builds)
  int abl;                        // timing ablations of attn_tail8_kernel (tools only): 1 no DMA, 2 no fragment reads, 3 no MFMA
  int w_cm;                       // both weight matrices are chunk-major [k / 32][n_cob * 32][32] (u / mo are when their ld is 32)
  int x_fm;                       // u / mo are FRAGMENT-major (SLIDE_F_OUT_FM, include/slide_engine.h; register-X kernel only)
  float inv_count;
};

__device__ __forceinline__ float other_half(float x) {  // value of lane ^ 32
  uint32_t a = __float_as_uint(x), b = a;
  lane32_swap(a, b);  // a = [x.lo | x.lo in the upper lanes], b = [x.hi in the lower lanes | x.hi]
  return __uint_as_float((threadIdx.x & 32) ? a : b);
}

// The epilogue of the fused attention tail on the two accumulator sets (values: bias, GroupNorm over the sample, ReLU; scores: bias,
// soft-max over a point's K rows; weighted sum, one row out per point).  vec_lds: [bias_s | bias_v | gamma | beta] of the tile's 64
// channels of THIS wave (vstride floats apart), red: 2 KB of scratch shared by the waves of one channel group, wave: the row wave
// (rows 64 wave ..); vectors / scratch must be visible / free on entry (the callers end their K loops with a workgroup barrier).
template <int NPXL>
__device__ __forceinline__ void attn_tail_finish(const AttnTailArgs &a, f32x16 (&sacc)[2][2], f32x16 (&vacc)[2][2], const float *vec_lds,
                                                 int vstride, float *red, int row0, int cob0, int wave) {
  // Round 6: the per-workgroup timeline (tools/ab/op_timeline.py) put 4.4 - 6.1 us of a tail workgroup's 11 - 21 us into this epilogue,
  // VALU-issue-bound (~1340 issue slots per wave).  Rewritten on register PAIRS (accumulator registers 2 i, 2 i + 1 are rows of one
  // point: v_pk_add / v_pk_fma_f32), log2(e) folded into the score bias step (exp2 of a difference: no multiply per value), one
  // v_rcp per output instead of an IEEE division, ONE lane-half exchange for numerator and denominator together, store addresses as
  // scalar base + one per-lane offset.
  using T = _Float16;
  constexpr int CBW = 2;
  constexpr int KLOG = NPXL - 4, KN = 1 << KLOG, GPB = 32 / KN;
  constexpr int WPS = (1 << NPXL) / 64;
  constexpr float LOG2E = 1.44269504088896340736f;
  const int lane = threadIdx.x & 63, half = lane >> 5, col = lane & 31;
  auto pr = [](const f32x16 &v, int i) __attribute__((always_inline)) { return f32x2{v[2 * i], v[2 * i + 1]}; };

  // ---- values: bias, GroupNorm over the sample (rows of WPS waves x the gs adjacent channel lanes), ReLU
  // red: [wave][cb][32 channels][sum, sumsq]
  const float *b_s = vec_lds, *b_v = vec_lds + vstride, *gam = vec_lds + 2 * vstride, *bet = vec_lds + 3 * vstride;
#pragma unroll
  for (int cb = 0; cb < CBW; ++cb) {
    const float bv = b_v[cb * 32 + col];
    const f32x2 bv2 = {bv, bv};
    f32x2 s2 = {0.f, 0.f}, ss2 = {0.f, 0.f};
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const f32x2 x = pr(vacc[cb][rb], i) + bv2;
        vacc[cb][rb][2 * i] = x[0]; vacc[cb][rb][2 * i + 1] = x[1];
        s2 += x;
        ss2 = __builtin_elementwise_fma(x, x, ss2);
      }
    float s = s2[0] + s2[1], ss = ss2[0] + ss2[1];
    s += other_half(s);
    ss += other_half(ss);
    if (half == 0) *reinterpret_cast<f32x2 *>(red + ((wave * CBW + cb) * 32 + col) * 2) = f32x2{s, ss};
  }
  __syncthreads();
  const int w0 = (wave / WPS) * WPS;
  // output row of the wave's first point: scalar base + the lane's channel (bytes)
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const int pt0 = (row0 + wave_s * 64) >> KLOG, npts = a.rows >> KLOG;
  const uint32_t lofs = (uint32_t)col * 2;
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
    const float bsl = b_s[cb * 32 + col] * LOG2E;
    const f32x2 g2 = {g, g}, bt2 = {bt, bt}, bs2 = {bsl, bsl}, l2 = {LOG2E, LOG2E};
    const bool cb_ok = cob0 + cb < a.n_cob;  // (uniform)
    // ---- softmax over the K neighbour rows of every point (base 2: the scores carry log2 e), weighted sum of the values, one row out per point
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int pg = 0; pg < GPB; ++pg) {
        // rows of point pg inside the 32-row block: 16 -> regs 8pg .. 8pg+7 (both halves); 8 -> regs 4pg .. 4pg+3
        constexpr int PPG = 8 / GPB;  // register pairs per point
        f32x2 sc[PPG], vv[PPG];
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < PPG; ++j) {
          sc[j] = __builtin_elementwise_fma(pr(sacc[cb][rb], pg * PPG + j), l2, bs2);
          vv[j] = __builtin_elementwise_fma(pr(vacc[cb][rb], pg * PPG + j), g2, bt2);
          vv[j][0] = fmaxf(vv[j][0], 0.f); vv[j][1] = fmaxf(vv[j][1], 0.f);
          m = fmaxf(m, fmaxf(sc[j][0], sc[j][1]));
        }
        m = fmaxf(m, other_half(m));
        const f32x2 m2 = {m, m};
        f32x2 den2 = {0.f, 0.f}, num2 = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < PPG; ++j) {
          const f32x2 d = sc[j] - m2;
          const f32x2 e = {__builtin_amdgcn_exp2f(d[0]), __builtin_amdgcn_exp2f(d[1])};
          den2 += e;
          num2 = __builtin_elementwise_fma(e, vv[j], num2);
        }
        // one exchange for both sums: afterwards the LOWER lanes hold (num.lo, num.hi), the upper lanes (den.lo, den.hi); the upper
        // lanes' total (den) then comes down with a second exchange
        uint32_t un = __float_as_uint(num2[0] + num2[1]), ud = __float_as_uint(den2[0] + den2[1]);
        lane32_swap(un, ud);
        const float tot = __uint_as_float(un) + __uint_as_float(ud);  // lower lanes: numerator, upper lanes: denominator
        uint32_t ua = __float_as_uint(tot), ub = ua;
        lane32_swap(ua, ub);  // ub (lower lanes) = the upper lanes' tot
        const int pidx = rb * GPB + pg;  // point of the wave
        if (half == 0 && pt0 + pidx < npts && cb_ok) {
          const T v = (T)(tot * __builtin_amdgcn_rcpf(__uint_as_float(ub)));
          const int ch = (cob0 + cb) * 32;
          *reinterpret_cast<T *>(reinterpret_cast<char *>(reinterpret_cast<T *>(a.out) + (size_t)(pt0 + pidx) * a.out_ld + ch) + lofs) = v;
          // chunk-major copy of the per-point table for the next block's gather-on-load GEMM
          if (a.out_cm)
            *reinterpret_cast<T *>(reinterpret_cast<char *>(reinterpret_cast<T *>(a.out_cm) + ((size_t)(cob0 + cb) * npts + pt0 + pidx) * 32) + lofs) = v;
          // second copy into the columns of a later concatenation buffer (the skip input of an FP block's second Mlp)
          if (a.out2 && ch + col < a.out2_n)
            *reinterpret_cast<T *>(reinterpret_cast<char *>(reinterpret_cast<T *>(a.out2) + (size_t)(pt0 + pidx) * a.out2_ld + ch) + lofs) = v;
        }
      }
  }
}

#ifndef SLIDE_ATTN_NST
#define SLIDE_ATTN_NST 3  // ring stages of the fused attention tail
#endif
template <int NPXL, int NST>
__device__ __forceinline__ void attn_tail_body(const AttnTailArgs &a) {
  using T = _Float16;
  constexpr int CBW = 2, RT = TM + 64, STAGE_B = RT * 64, LPW = RT / 16 / 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int ntc = (a.n_cob + CBW - 1) / CBW;
  const int xcd = blockIdx.x & 7, q0 = blockIdx.x >> 3;
  const int tc = q0 % ntc, tr = (q0 / ntc) * 8 + xcd;
  if (tr * TM >= a.rows) return;
  const int row0 = tr * TM, cob0 = tc * CBW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
  float *const vec_lds = reinterpret_cast<float *>(smem_raw + (size_t)NST * STAGE_B);  // [4 vectors][CBW*32]
  for (int i = tid; i < 4 * CBW * 32; i += 256) {
    const int which = i / (CBW * 32), c = i - which * (CBW * 32), gc = cob0 * 32 + c;
    vec_lds[i] = gc < a.n_cob * 32 ? a.vec[(size_t)which * a.n_cob * 32 + gc] : 0.f;
  }
  int wrow[CBW], wkey[CBW], xrow[2], xkey[2];
#pragma unroll
  for (int cb = 0; cb < CBW; ++cb) {
    const int trow = TM + cb * 32 + col;
    wrow[cb] = trow * 64; wkey[cb] = (trow >> 2) & 3;
  }
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const int trow = wave * 64 + rb * 32 + col;
    xrow[rb] = trow * 64; xkey[rb] = (trow >> 2) & 3;
  }
  // one LDS-DMA ring GEMM: acc[cb][rb] = D[row][channel] (lane: channel col of block cb; reg r: row (r&3)+8(r>>2)+4 half)
  auto run = [&](const void *Xp, const void *Wp, int x_ld, int k_pad, f32x16 (&acc)[CBW][2]) __attribute__((always_inline)) {
    // chunk-major operands as in glds_tile: X when x_ld == 32, the weights when a.w_cm
    const size_t x_cs = x_ld == 32 ? (size_t)a.rows * 32 : 32;
    const size_t w_cs = a.w_cm ? (size_t)a.n_cob * 32 * 32 : 32;
    const int w_ld = a.w_cm ? 32 : k_pad;
    const T *gp[LPW];
#pragma unroll
    for (int j = 0; j < LPW; ++j) {
      const int trow = 16 * (j * 4 + wave) + (lane >> 2);
      const int piece = (lane & 3) ^ ((trow >> 2) & 3);
      if (trow < TM) {
        int grow = row0 + trow;
        grow = grow < a.rows ? grow : a.rows - 1;
        gp[j] = reinterpret_cast<const T *>(Xp) + (size_t)grow * x_ld + piece * 8;
      } else {
        int gco = cob0 * 32 + (trow - TM);
        gco = gco < a.n_cob * 32 ? gco : a.n_cob * 32 - 1;
        gp[j] = reinterpret_cast<const T *>(Wp) + (size_t)gco * w_ld + piece * 8;
      }
    }
    auto issue = [&](int kc, int st) {
#pragma unroll
      for (int j = 0; j < LPW; ++j)
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)(gp[j] + (size_t)kc * (j < TM / 64 ? x_cs : w_cs)),
                                         (__attribute__((address_space(3))) void *)(smem_raw + (size_t)st * STAGE_B +
                                                                                    (j * 4 + wave) * 1024),
                                         16, 0, 0);
    };
#pragma unroll
    for (int i = 0; i < CBW; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nk = k_pad / 32;
#pragma unroll
    for (int s0 = 0; s0 < NST - 1; ++s0)
      if (s0 < nk) issue(s0, s0);
    for (int kc = 0; kc < nk; ++kc) {
      if (kc + NST - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * LPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kc + NST - 1 < nk) issue(kc + NST - 1, (kc + NST - 1) % NST);
      const unsigned char *sb = smem_raw + (size_t)(kc % NST) * STAGE_B;
#pragma unroll
      for (int st2 = 0; st2 < 2; ++st2) {
        f16x8 wf[CBW], xf[2];
        const int piece = st2 * 2 + half;
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb) wf[cb] = *reinterpret_cast<const f16x8 *>(sb + wrow[cb] + ((piece ^ wkey[cb]) << 4));
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) xf[rb] = *reinterpret_cast<const f16x8 *>(sb + xrow[rb] + ((piece ^ xkey[rb]) << 4));
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
          for (int rb = 0; rb < 2; ++rb)
            acc[cb][rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xf[rb], wf[cb], acc[cb][rb], 0, 0, 0);  // rows x channels
      }
    }
    __syncthreads();  // ring drained and free (also orders the staged vectors before their first use)
  };
  f32x16 sacc[CBW][2], vacc[CBW][2];
  run(a.X1, a.W1, a.x1_ld, a.k1, sacc);
  run(a.X2, a.W2, a.x2_ld, a.k2, vacc);
  attn_tail_finish<NPXL>(a, sacc, vacc, vec_lds, CBW * 32, reinterpret_cast<float *>(smem_raw), row0, cob0, wave);
}

// REGISTER-X form (round 5; the default, SLIDE_TAIL_RX=0 restores the ring form above).  16 KB of a 20 KB ring stage is the X tile,
// which no two waves share -- a wave's MFMAs read only its own 64 rows.  Here a wave loads ITS X fragments straight into registers
// (chunk-major u / mo: a 32-row block of one chunk is 2 KB contiguous, a lane's 16 bytes are its MFMA A fragment as stored) RXD chunks
// ahead, and only the weights (64 channels x 32 k = 4 KB per chunk, read by all four waves) go through an LDS-DMA ring of RXD + 1
// stages: 23 KB of LDS per workgroup instead of 61, a quarter of the ds_reads, RXD - 1 ... RXD chunks in flight per workgroup instead
// of two.  ONE pipeline over both contractions, values first (their chunk count must be a multiple of RXD -- the launcher checks --
// so that the register slot of a chunk is a compile-time index), then scores.  Measured (tools/ab/r05_tailrx.sh): the feature step's two
// SA tails 67.0 -> 64.4 us stand-alone, 378.9 -> 382.6 shapes/s in the arrangement (three alternating pairs) -- the deeper prefetch
// buys little: the tile's fill rate (~58 GB/s per CU, round 3's ablations) is a throughput cap, not a latency one.
constexpr int RXD = 4;
// WC = 2 (eight waves, tile 256 rows x 128 channels: wave (wr, wc) owns rows 64 wr .. and the channel half wc, a row block's fragments
// are requested by two waves) was measured and is not instantiated: 76.5 us per feature step's two SA tails against 64.4 (WC = 1) and
// 67.0 (ring form) -- the second request is not free, and one eight-wave workgroup per CU overlaps less than two of four.
// FM (round 6): u / mo FRAGMENT-major -- inside a 32-row group the chunk's 2 KB are [k16 step][k half][row][8 halves], i.e. the two
// A fragments of the group as the wave's lanes hold them: each global_load_dwordx4 below then reads 1 KB of consecutive memory instead
// of 32 B from each of 32 rows 64 B apart (the request-bound pattern; tools/lds_fill.hip XP vs XF: 28 -> 44-49 B/clk/CU into VGPRs).
template <int NPXL, int WC, bool FM>
__device__ __forceinline__ void attn_tail_rx_body(const AttnTailArgs &a) {
  using T = _Float16;
  constexpr int CBW = 2, CBWT = CBW * WC, NSTW = RXD + 1, WSTAGE = 64 * WC * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int ntc = (a.n_cob + CBWT - 1) / CBWT;
  const int xcd = blockIdx.x & 7, q0 = blockIdx.x >> 3;
  const int tc = q0 % ntc, tr = (q0 / ntc) * 8 + xcd;
  if (tr * TM >= a.rows) return;
  const int row0 = tr * TM, cob0 = tc * CBWT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
  const int wr = wave & 3, wc = wave >> 2;
  float *const vec_lds = reinterpret_cast<float *>(smem_raw + (size_t)NSTW * WSTAGE);  // [4 vectors][CBWT*32]
  float *const red = vec_lds + 4 * CBWT * 32;                                          // [wc][4 row waves][CBW][32][2]
  // the tile's four vectors go to LDS by LDS-DMA (wave w: vector w, 64 floats) -- through registers the ds_write's wait was a full
  // memory round trip BEFORE the first chunk load could issue (~1 us of every workgroup: "primed" in tools/ab/op_timeline.py); as the
  // oldest loads of the pipeline they have landed with chunk 0 and the first step's barrier publishes them.  (Channels past n_cob * 32
  // read a valid address: their values are never stored and never mix with valid channels -- GroupNorm groups lie inside a block.)
  static_assert(WC == 1, "vector staging: one LDS-DMA instruction per wave covers the tile's 64 channels");
  {
    int gc = cob0 * 32 + lane;
    gc = gc < a.n_cob * 32 ? gc : a.n_cob * 32 - 1;
    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)(a.vec + (size_t)wave * a.n_cob * 32 + gc),
                                     (__attribute__((address_space(3))) void *)(vec_lds + wave * 64), 4, 0, 0);
  }
  // weights: wave w stages channels 16 w .. 16 w + 15 of the tile (one 1 KB piece per chunk); fragments: lane = channel, swizzled pieces
  const int wch = 16 * wave + (lane >> 2);
  int gco = cob0 * 32 + wch;
  gco = gco < a.n_cob * 32 ? gco : a.n_cob * 32 - 1;
  const int wpiece = (lane & 3) ^ ((wch >> 2) & 3);
  const size_t w_cs = a.w_cm ? (size_t)a.n_cob * 32 * 32 : 32;
  const T *const w2p = reinterpret_cast<const T *>(a.W2) + (size_t)gco * (a.w_cm ? 32 : a.k2) + wpiece * 8;
  const T *const w1p = reinterpret_cast<const T *>(a.W1) + (size_t)gco * (a.w_cm ? 32 : a.k1) + wpiece * 8;
  int wrow[CBW], wkey[CBW];
#pragma unroll
  for (int cb = 0; cb < CBW; ++cb) {
    const int trow = wc * 64 + cb * 32 + col;
    wrow[cb] = trow * 64; wkey[cb] = (trow >> 2) & 3;
  }
  // X: lane (col, half) of row block rb reads row  row0 + 64 wave + 32 rb + col,  k pieces  2 st2 + half  of the chunk
  const size_t x2_cs = a.x2_ld == 32 ? (size_t)a.rows * 32 : 32, x1_cs = a.x1_ld == 32 ? (size_t)a.rows * 32 : 32;
  const T *x2p[2], *x1p[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    int grow = row0 + wr * 64 + rb * 32 + col;
    grow = grow < a.rows ? grow : a.rows - 1;
    const size_t fmo = (size_t)(grow & ~31) * 32 + half * 256 + (grow & 31) * 8;
    x2p[rb] = reinterpret_cast<const T *>(a.X2) + (FM ? fmo : (size_t)grow * a.x2_ld + half * 8);
    x1p[rb] = reinterpret_cast<const T *>(a.X1) + (FM ? fmo : (size_t)grow * a.x1_ld + half * 8);
  }
  const int nk2 = a.k2 / 32, total = nk2 + a.k1 / 32;
  f16x8 xq[RXD][2][2];
  // EVERY chunk slot issues its five loads, also past the last chunk (there: all lanes read one valid address -- a broadcast, next to no
  // traffic -- and nobody consumes the result): with unconditional issues the number of loads behind a chunk's is the constant
  // 5 (RXD - 1), for the manual wait below and for the compiler's own wait-count insertion alike (a conditional issue made it fall
  // back to vmcnt(0) at every use, which serialises the pipeline).
  // FM: every address is SCALAR base + one per-lane 32-bit offset (lane * 16 bytes: a fragment is 1 KB in lane order; the second k16
  // step, the second row group are immediate offsets when the group exists) -- the chunk-major form below spends ~10 VALU
  // instructions per load on 64-bit pointer arithmetic, ~50 per chunk against the chunk's 8 MFMAs (32 clocks each): as much issue
  // time as the matrix work itself.  Idle slots (past the last chunk) read one address in all lanes (offset 0 of chunk 0).
  const int wr_s = __builtin_amdgcn_readfirstlane(wr), ngrp = a.rows >> 5;
  const int g0u = (row0 >> 5) + wr_s * 2, g0 = g0u < ngrp ? g0u : ngrp - 1, g1 = g0u + 1 < ngrp ? g0u + 1 : ngrp - 1;
  const uint64_t xg0 = (uint64_t)g0 * 2048, xg1 = (uint64_t)g1 * 2048, xcsb = (uint64_t)a.rows * 64;  // bytes
  const uint32_t lane16 = lane * 16, woff = (uint32_t)(gco * 32 + wpiece * 8) * 2;
  auto issue_fm = [&](int c, f16x8 (&x)[2][2]) __attribute__((always_inline)) {
    const bool live = c < total;
    const int cl = live ? c : 0;
    const bool second = cl >= nk2;
    const int kc = second ? cl - nk2 : cl;
    const uint32_t vo = live ? lane16 : 0u, vw = live ? woff : 0u;
    const uint64_t xb = reinterpret_cast<uint64_t>(second ? a.X1 : a.X2) + (uint64_t)kc * xcsb, b0 = xb + xg0, b1 = xb + xg1;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(x[0][0]) : "v"(vo), "s"(b0) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(x[0][1]) : "v"(vo), "s"(b0) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(x[1][0]) : "v"(vo), "s"(b1) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(x[1][1]) : "v"(vo), "s"(b1) : "memory");
    const uint64_t wb = reinterpret_cast<uint64_t>(second ? a.W1 : a.W2) + (uint64_t)kc * (w_cs * 2);
    __builtin_amdgcn_global_load_lds(reinterpret_cast<const GLOBAL_AS void *>(wb + vw),
                                     (__attribute__((address_space(3))) void *)(smem_raw + (size_t)(c % NSTW) * WSTAGE + wave * 1024), 16, 0, 0);
  };
  auto issue = [&](int c, f16x8 (&x)[2][2]) __attribute__((always_inline)) {
    if constexpr (FM) { issue_fm(c, x); return; }
    // (branch-free: an idle slot's addresses collapse onto `dummy` through a mask, not through a select the compiler could turn into
    //  control flow -- every path through the pipeline must carry the same loads)
    const bool second = c >= nk2;
    const int kc = second ? c - nk2 : c;
    const uint64_t mask = c < total ? ~0ull : 0ull;
    const uint64_t dummy = reinterpret_cast<uint64_t>(a.vec);
    const uint64_t wp = dummy + ((reinterpret_cast<uint64_t>((second ? w1p : w2p) + (size_t)kc * w_cs) - dummy) & mask);
    const size_t xo = (size_t)kc * (second ? x1_cs : x2_cs);
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const uint64_t xp = dummy + ((reinterpret_cast<uint64_t>((second ? x1p[rb] : x2p[rb]) + xo) - dummy) & mask);
      // (asm: hipcc's wait-count insertion answers ANY register load pending beside an LDS-DMA load with vmcnt(0) -- the two may
      //  return out of order for all it knows -- which drains the pipeline once per round; these loads are waited for by hand)
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(x[rb][0]) : "v"(xp) : "memory");
      if constexpr (FM) asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(x[rb][1]) : "v"(xp) : "memory");
      else asm volatile("global_load_dwordx4 %0, %1, off offset:32" : "=v"(x[rb][1]) : "v"(xp) : "memory");
    }
    __builtin_amdgcn_global_load_lds(reinterpret_cast<const GLOBAL_AS void *>(wp),
                                     (__attribute__((address_space(3))) void *)(smem_raw + (size_t)(c % NSTW) * WSTAGE + wave * 1024), 16, 0, 0);
  };
  f32x16 sacc[CBW][2], vacc[CBW][2];
#pragma unroll
  for (int i = 0; i < CBW; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { sacc[i][j][r] = 0.f; vacc[i][j][r] = 0.f; }
  SLIDE_STAMP(a, 0);
#pragma unroll
  for (int j = 0; j < RXD; ++j) issue(j, xq[j]);
  SLIDE_STAMP(a, 1);
  auto step = [&](int c, f16x8 (&x)[2][2], f32x16 (&acc)[CBW][2]) __attribute__((always_inline)) {
    // chunk c's loads have landed when only those of chunks c + 1 .. c + RXD - 1 are outstanding (the operands tie the fragments'
    // uses to this wait)
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(x[0][0]), "+v"(x[0][1]), "+v"(x[1][0]), "+v"(x[1][1]) : "n"((RXD - 1) * 5) : "memory");
    __builtin_amdgcn_s_barrier();  // every wave's piece of W chunk c has landed; W stage (c - 1) % NSTW is free
    const unsigned char *sb = smem_raw + (size_t)(c % NSTW) * WSTAGE;
#pragma unroll
    for (int st2 = 0; st2 < 2; ++st2) {
      f16x8 wf[CBW];
      const int piece = st2 * 2 + half;
#pragma unroll
      for (int cb = 0; cb < CBW; ++cb) wf[cb] = *reinterpret_cast<const f16x8 *>(sb + wrow[cb] + ((piece ^ wkey[cb]) << 4));
#pragma unroll
      for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
          acc[cb][rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[rb][st2], wf[cb], acc[cb][rb], 0, 0, 0);  // rows x channels
    }
    issue(c + RXD, x);
  };
  for (int c0 = 0; c0 < nk2; c0 += RXD) {
#pragma unroll
    for (int j = 0; j < RXD; ++j) step(c0 + j, xq[j], vacc);
    if (c0 == 0) SLIDE_STAMP(a, 2);
  }
  SLIDE_STAMP(a, 3);
  for (int c0 = nk2; c0 < total; c0 += RXD) {  // (leaves from the middle of a round after the last chunk: no path re-joins the pipeline)
#pragma unroll
    for (int j = 0; j < RXD; ++j) {
      step(c0 + j, xq[j], sacc);
      if (c0 + j + 1 >= total) break;
    }
  }
  // the idle slots' loads: their registers stay reserved until they have landed
#pragma unroll
  for (int j = 0; j < RXD; ++j)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(xq[j][0][0]), "+v"(xq[j][0][1]), "+v"(xq[j][1][0]), "+v"(xq[j][1][1]) :: "memory");
  SLIDE_STAMP(a, 4);
  __syncthreads();  // (orders the staged vectors before their first use)
  SLIDE_STAMP(a, 5);
  attn_tail_finish<NPXL>(a, sacc, vacc, vec_lds + wc * 64, CBWT * 32, red + wc * (4 * CBW * 32 * 2), row0, cob0 + wc * CBW, wr);
  SLIDE_STAMP(a, 6);
#ifdef SLIDE_TIMELINE
  if (a.dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); SLIDE_STAMP(a, 7); }
#endif
}

// WIDE form (round 3): 256 rows x 128 channels on the same four waves.  Per MFMA the tile moves 1.1 KB through the LDS instead
// of 1.6 KB (X fragments feed four channel blocks, an X chunk is written once per 128 channels) -- the 64-channel tile's K
// loop is LDS-bandwidth-bound at ~60 % of the matrix pipe -- and a sample's u / mo tiles are read from L2
// by half as many workgroups.  128 accumulator registers hold ONE contraction at a time: VALUES first (GroupNorm statistics,
// normalise, ReLU, packed to fp16: 64 registers), then the SCORES into the same accumulators, then the soft-max weighted sum.
template <int NPXL, int NST>
__device__ __forceinline__ void attn_tail_wide_body(const AttnTailArgs &a) {
  using T = _Float16;
  constexpr int CBW = 4, RT = TM + 32 * CBW, STAGE_B = RT * 64, LPW = RT / 16 / 4;
  constexpr int KLOG = NPXL - 4, KN = 1 << KLOG, GPB = 32 / KN;  // neighbours per point, points per 32-row block
  constexpr int WPS = (1 << NPXL) / 64;                            // waves per sample
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int ntc = (a.n_cob + CBW - 1) / CBW;
  const int xcd = blockIdx.x & 7, q0 = blockIdx.x >> 3;
  const int tc = q0 % ntc, tr = (q0 / ntc) * 8 + xcd;
  if (tr * TM >= a.rows) return;
  const int row0 = tr * TM, cob0 = tc * CBW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
  float *const vec_lds = reinterpret_cast<float *>(smem_raw + (size_t)NST * STAGE_B);  // [4 vectors][CBW*32]
  for (int i = tid; i < 4 * CBW * 32; i += 256) {
    const int which = i / (CBW * 32), c = i - which * (CBW * 32), gc = cob0 * 32 + c;
    vec_lds[i] = gc < a.n_cob * 32 ? a.vec[(size_t)which * a.n_cob * 32 + gc] : 0.f;
  }
  int wrow[CBW], wkey[CBW], xrow[2], xkey[2];
#pragma unroll
  for (int cb = 0; cb < CBW; ++cb) {
    const int trow = TM + cb * 32 + col;
    wrow[cb] = trow * 64; wkey[cb] = (trow >> 2) & 3;
  }
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const int trow = wave * 64 + rb * 32 + col;
    xrow[rb] = trow * 64; xkey[rb] = (trow >> 2) & 3;
  }
  // one LDS-DMA ring GEMM: acc[cb][rb] = D[row][channel] (lane: channel col of block cb; reg r: row (r&3)+8(r>>2)+4 half)
  auto run = [&](const void *Xp, const void *Wp, int x_ld, int k_pad, f32x16 (&acc)[CBW][2]) __attribute__((always_inline)) {
    // chunk-major operands as in glds_tile: X when x_ld == 32, the weights when a.w_cm
    const size_t x_cs = x_ld == 32 ? (size_t)a.rows * 32 : 32;
    const size_t w_cs = a.w_cm ? (size_t)a.n_cob * 32 * 32 : 32;
    const int w_ld = a.w_cm ? 32 : k_pad;
    const T *gp[LPW];
#pragma unroll
    for (int j = 0; j < LPW; ++j) {
      const int trow = 16 * (j * 4 + wave) + (lane >> 2);
      const int piece = (lane & 3) ^ ((trow >> 2) & 3);
      if (trow < TM) {
        int grow = row0 + trow;
        grow = grow < a.rows ? grow : a.rows - 1;
        gp[j] = reinterpret_cast<const T *>(Xp) + (size_t)grow * x_ld + piece * 8;
      } else {
        int gco = cob0 * 32 + (trow - TM);
        gco = gco < a.n_cob * 32 ? gco : a.n_cob * 32 - 1;
        gp[j] = reinterpret_cast<const T *>(Wp) + (size_t)gco * w_ld + piece * 8;
      }
    }
    auto issue = [&](int kc, int st) {
#pragma unroll
      for (int j = 0; j < LPW; ++j)
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)(gp[j] + (size_t)kc * (j < TM / 64 ? x_cs : w_cs)),
                                         (__attribute__((address_space(3))) void *)(smem_raw + (size_t)st * STAGE_B +
                                                                                    (j * 4 + wave) * 1024),
                                         16, 0, 0);
    };
#pragma unroll
    for (int i = 0; i < CBW; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nk = k_pad / 32;
#pragma unroll
    for (int s0 = 0; s0 < NST - 1; ++s0)
      if (s0 < nk) issue(s0, s0);
    for (int kc = 0; kc < nk; ++kc) {
      if (kc + NST - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * LPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kc + NST - 1 < nk) issue(kc + NST - 1, (kc + NST - 1) % NST);
      const unsigned char *sb = smem_raw + (size_t)(kc % NST) * STAGE_B;
#pragma unroll
      for (int st2 = 0; st2 < 2; ++st2) {
        f16x8 wf[CBW], xf[2];
        const int piece = st2 * 2 + half;
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb) wf[cb] = *reinterpret_cast<const f16x8 *>(sb + wrow[cb] + ((piece ^ wkey[cb]) << 4));
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) xf[rb] = *reinterpret_cast<const f16x8 *>(sb + xrow[rb] + ((piece ^ xkey[rb]) << 4));
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
          for (int rb = 0; rb < 2; ++rb)
            acc[cb][rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xf[rb], wf[cb], acc[cb][rb], 0, 0, 0);  // rows x channels
      }
    }
    __syncthreads();  // ring drained and free (also orders the staged vectors before their first use)
  };
  f32x16 acc[CBW][2];
  run(a.X2, a.W2, a.x2_ld, a.k2, acc);  // values first

  // ---- values: bias, GroupNorm over the sample (rows of WPS waves x the gs adjacent channel lanes), ReLU, packed to fp16
  float *const red = reinterpret_cast<float *>(smem_raw);  // [wave][cb][32 channels][sum, sumsq]
  const float *b_s = vec_lds, *b_v = vec_lds + CBW * 32, *gam = vec_lds + 2 * CBW * 32, *bet = vec_lds + 3 * CBW * 32;
#pragma unroll
  for (int cb = 0; cb < CBW; ++cb) {
    const float bv = b_v[cb * 32 + col];
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float x = acc[cb][rb][r] + bv;
        acc[cb][rb][r] = x;
        s += x;
        ss = fmaf(x, x, ss);
      }
    s += other_half(s);
    ss += other_half(ss);
    if (half == 0) *reinterpret_cast<f32x2 *>(red + ((wave * CBW + cb) * 32 + col) * 2) = f32x2{s, ss};
  }
  __syncthreads();
  const int w0 = (wave / WPS) * WPS;
  f16x2 vp[CBW][2][8];  // relu(GN(values)) of this lane's channel, rows (2 j, 2 j + 1) of the block
#pragma unroll
  for (int cb = 0; cb < CBW; ++cb) {
    f32x2 t = {0.f, 0.f};
#pragma unroll
    for (int w = 0; w < WPS; ++w) t += *reinterpret_cast<const f32x2 *>(red + (((w0 + w) * CBW + cb) * 32 + col) * 2);
    float s = t[0], ss = t[1];
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
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        vp[cb][rb][j] = f16x2{(T)fmaxf(fmaf(acc[cb][rb][2 * j], g, bt), 0.f), (T)fmaxf(fmaf(acc[cb][rb][2 * j + 1], g, bt), 0.f)};
  }
  __syncthreads();  // every wave has read the statistics: the ring area is free for the score contraction
  run(a.X1, a.W1, a.x1_ld, a.k1, acc);  // scores into the same accumulators
#pragma unroll
  for (int cb = 0; cb < CBW; ++cb) {
    const float bs = b_s[cb * 32 + col];
    // ---- softmax over the K neighbour rows of every point, weighted sum of the values, one row out per point
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int pg = 0; pg < GPB; ++pg) {
        constexpr int RPG = 16 / GPB;
        float sc[RPG], vv[RPG];
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < RPG; ++j) {
          sc[j] = acc[cb][rb][pg * RPG + j] + bs;
          vv[j] = (float)vp[cb][rb][(pg * RPG + j) >> 1][(pg * RPG + j) & 1];
          m = fmaxf(m, sc[j]);
        }
        m = fmaxf(m, other_half(m));
        float den = 0.f, num = 0.f;
#pragma unroll
        for (int j = 0; j < RPG; ++j) {
          const float e = __expf(sc[j] - m);
          den += e;
          num = fmaf(e, vv[j], num);
        }
        den += other_half(den);
        num += other_half(num);
        const int rbase = row0 + wave * 64 + rb * 32 + pg * KN;
        if (half == 0 && rbase < a.rows && cob0 + cb < a.n_cob) {
          const T v = (T)(num / den);
          reinterpret_cast<T *>(a.out)[(size_t)(rbase >> KLOG) * a.out_ld + (cob0 + cb) * 32 + col] = v;
          if (a.out_cm)
            reinterpret_cast<T *>(a.out_cm)[((size_t)(cob0 + cb) * (a.rows >> KLOG) + (rbase >> KLOG)) * 32 + col] = v;
          if (a.out2 && (cob0 + cb) * 32 + col < a.out2_n)
            reinterpret_cast<T *>(a.out2)[(size_t)(rbase >> KLOG) * a.out2_ld + (cob0 + cb) * 32 + col] = v;
        }
      }
  }
}

template <int NPXL>
__global__ __launch_bounds__(256, 1) void attn_tail_wide_kernel(AttnTailArgs a) {
  attn_tail_wide_body<NPXL, 3>(a);
}

template <int NPXL>
__global__ __launch_bounds__(256, 2) void attn_tail_kernel(AttnTailArgs a) {
  attn_tail_body<NPXL, SLIDE_ATTN_NST>(a);
}

template <int NPXL, bool FM>
__global__ __launch_bounds__(256, 2) void attn_tail_rx_kernel(AttnTailArgs a) {
  attn_tail_rx_body<NPXL, 1, FM>(a);
}

// the same tile on a TWO-stage ring (41 KB) inside the 168-register budget: three workgroups per CU instead of two
template <int NPXL>
__global__ __launch_bounds__(256, 3) void attn_tail_occ3_kernel(AttnTailArgs a) {
  attn_tail_body<NPXL, 2>(a);
}

// Eight-wave form of the fused attention tail (round 3): tile 256 rows x 128 channels -- wave (wr, wc) owns rows 64 wr .. and the
// channel half wc, so an X chunk is fetched once per 128 channels (24 KB of L2 -> LDS per 2 MFLOP instead of 20 KB per 1) --,
// ring stages 64 deep (two chunk images: one barrier per 16 MFMAs of a wave), ONE continuous ring over the chunks of both
// GEMMs (no drain between the score and the value contraction), fragment reads of the next 16-deep step issued before the
// current step's MFMAs.  Same arithmetic, same epilogue as attn_tail_kernel; used when the layer has at least eight blocks.
template <int NPXL>
__global__ __launch_bounds__(512, 2) void attn_tail8_kernel(AttnTailArgs a) {
  using T = _Float16;
  constexpr int CBW = 2, NST = 3, RT = TM + 128, CH_B = RT * 64, STAGE_B = 2 * CH_B, LPW = RT / 16 / 8;  // 3 DMA / wave / chunk
  constexpr int KLOG = NPXL - 4, KN = 1 << KLOG, GPB = 32 / KN;
  constexpr int WPS = (1 << NPXL) / 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int ntc = (a.n_cob + 3) / 4;
  const int xcd = blockIdx.x & 7, q0 = blockIdx.x >> 3;
  const int tc = q0 % ntc, tr = (q0 / ntc) * 8 + xcd;
  if (tr * TM >= a.rows) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), wave = wv & 3, wc = wv >> 2;
  const int half = lane >> 5, col = lane & 31;
  const int row0 = tr * TM, cobt = tc * 4, cob0 = cobt + wc * CBW;
  SLIDE_STAMP(a, 0);
  float *const vec_lds = reinterpret_cast<float *>(smem_raw + (size_t)NST * STAGE_B);  // [4 vectors][4 * 32]
  for (int i = tid; i < 4 * 128; i += 512) {
    const int which = i >> 7, c = i & 127, gc = cobt * 32 + c;
    vec_lds[i] = gc < a.n_cob * 32 ? a.vec[(size_t)which * a.n_cob * 32 + gc] : 0.f;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int wrow[CBW], wkey[CBW], xrow[2], xkey[2];
#pragma unroll
  for (int cb = 0; cb < CBW; ++cb) {
    const int trow = TM + (wc * CBW + cb) * 32 + col;
    wrow[cb] = trow * 64; wkey[cb] = (trow >> 2) & 3;
  }
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const int trow = wave * 64 + rb * 32 + col;
    xrow[rb] = trow * 64; xkey[rb] = (trow >> 2) & 3;
  }
  const int nk1 = a.k1 >> 5, nk2 = a.k2 >> 5, nkt = nk1 + nk2, nks = (nkt + 1) >> 1;
  // this lane's source piece of the wave's three DMA instructions per chunk, for both GEMMs
  const T *gp[2][LPW];
  size_t cs[2][LPW];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const void *Xp = g ? a.X2 : a.X1, *Wp = g ? a.W2 : a.W1;
    const int x_ld = g ? a.x2_ld : a.x1_ld, k_pad = g ? a.k2 : a.k1;
    const size_t x_cs = x_ld == 32 ? (size_t)a.rows * 32 : 32;
    const size_t w_cs = a.w_cm ? (size_t)a.n_cob * 32 * 32 : 32;
    const int w_ld = a.w_cm ? 32 : k_pad;
#pragma unroll
    for (int j = 0; j < LPW; ++j) {
      const int trow = 16 * (j * 8 + wv) + (lane >> 2);
      const int piece = (lane & 3) ^ ((trow >> 2) & 3);
      if (trow < TM) {
        int grow = row0 + trow;
        grow = grow < a.rows ? grow : a.rows - 1;
        gp[g][j] = reinterpret_cast<const T *>(Xp) + (size_t)grow * x_ld + piece * 8;
        cs[g][j] = x_cs;
      } else {
        int gco = cobt * 32 + (trow - TM);
        gco = gco < a.n_cob * 32 ? gco : a.n_cob * 32 - 1;
        gp[g][j] = reinterpret_cast<const T *>(Wp) + (size_t)gco * w_ld + piece * 8;
        cs[g][j] = w_cs;
      }
    }
  }
  auto issue = [&](int st) __attribute__((always_inline)) {
    unsigned char *dst = smem_raw + (size_t)(st % NST) * STAGE_B;
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
      int h = st * 2 + c2;
      h = h < nkt ? h : nkt - 1;  // odd total: the last image is a dummy (never read)
      const int g = h >= nk1, kc = g ? h - nk1 : h;
#pragma unroll
      for (int j = 0; j < LPW; ++j)
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)((g ? gp[1][j] : gp[0][j]) + (size_t)kc * (g ? cs[1][j] : cs[0][j])),
                                         (__attribute__((address_space(3))) void *)(dst + c2 * CH_B + (j * 8 + wv) * 1024), 16, 0, 0);
    }
  };
  // stage st must have landed; the next one (2 * LPW instructions per wave) stays in flight: with ONE workgroup per CU the
  // bytes in flight are what the L2 -> LDS rate hangs on (two-stage ring: 1.26 us per 48 KB stage, DMA-latency bound)
  auto stage_ready = [&](int st) __attribute__((always_inline)) {
    if (st + 1 < nks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (st + 2 < nks && a.abl != 1) issue(st + 2);  // overwrites the stage consumed at st - 1
  };
  f32x16 sacc[CBW][2], vacc[CBW][2];
#pragma unroll
  for (int i = 0; i < CBW; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { sacc[i][j][r] = 0.f; vacc[i][j][r] = 0.f; }
  struct Frag { f16x8 wf[CBW], xf[2]; };
  auto loadf = [&](Frag &o, const unsigned char *sb, int st2) __attribute__((always_inline)) {
    if (a.abl == 2 && sb != smem_raw) return;
    const int piece = st2 * 2 + half;
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb) o.wf[cb] = *reinterpret_cast<const f16x8 *>(sb + wrow[cb] + ((piece ^ wkey[cb]) << 4));
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) o.xf[rb] = *reinterpret_cast<const f16x8 *>(sb + xrow[rb] + ((piece ^ xkey[rb]) << 4));
  };
  auto mma = [&](const Frag &o, f32x16 (&acc)[CBW][2]) __attribute__((always_inline)) {
    if (a.abl == 3) { asm volatile("" :: "v"(o.xf[0]), "v"(o.xf[1]), "v"(o.wf[0]), "v"(o.wf[1])); return; }
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
        acc[cb][rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.xf[rb], o.wf[cb], acc[cb][rb], 0, 0, 0);  // rows x channels
  };
  issue(0);
  if (nks > 1) issue(1);
  SLIDE_STAMP(a, 1);
  stage_ready(0);
  SLIDE_STAMP(a, 2);
  Frag cur, nxt;
  loadf(cur, smem_raw, 0);
  // (nk1, nk2 even: a ring stage never straddles the two GEMMs, each accumulator set has its own loop)
  auto run = [&](int st_lo, int st_hi, f32x16 (&acc)[CBW][2]) __attribute__((always_inline)) {
    for (int st = st_lo; st < st_hi; ++st) {
      const unsigned char *sb = smem_raw + (size_t)(st % NST) * STAGE_B;
      loadf(nxt, sb, 1);
      mma(cur, acc);
      loadf(cur, sb + CH_B, 0);
      mma(nxt, acc);
      loadf(nxt, sb + CH_B, 1);
      mma(cur, acc);
      if (st + 1 < nks) {
        stage_ready(st + 1);
        loadf(cur, smem_raw + (size_t)((st + 1) % NST) * STAGE_B, 0);
      }
      mma(nxt, acc);
    }
  };
  run(0, nk1 >> 1, sacc);
  SLIDE_STAMP(a, 3);
  run(nk1 >> 1, nks, vacc);
  SLIDE_STAMP(a, 4);
  __syncthreads();  // ring drained and free

  // ---- values: bias, GroupNorm over the sample (rows of WPS waves x the gs adjacent channel lanes), ReLU
  float *const red = reinterpret_cast<float *>(smem_raw);  // [wave 0..7][cb][32 channels][sum, sumsq]
  const float *b_s = vec_lds + wc * 64, *b_v = vec_lds + 128 + wc * 64, *gam = vec_lds + 256 + wc * 64, *bet = vec_lds + 384 + wc * 64;
#pragma unroll
  for (int cb = 0; cb < CBW; ++cb) {
    const float bv = b_v[cb * 32 + col];
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float x = vacc[cb][rb][r] + bv;
        vacc[cb][rb][r] = x;
        s += x;
        ss = fmaf(x, x, ss);
      }
    s += other_half(s);
    ss += other_half(ss);
    if (half == 0) *reinterpret_cast<f32x2 *>(red + ((wv * CBW + cb) * 32 + col) * 2) = f32x2{s, ss};
  }
  __syncthreads();
  SLIDE_STAMP(a, 5);
  const int w0 = wc * 4 + (wave / WPS) * WPS;
#pragma unroll
  for (int cb = 0; cb < CBW; ++cb) {
    f32x2 t = {0.f, 0.f};
#pragma unroll
    for (int w = 0; w < WPS; ++w) t += *reinterpret_cast<const f32x2 *>(red + (((w0 + w) * CBW + cb) * 32 + col) * 2);
    float s = t[0], ss = t[1];
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
    const float bs = b_s[cb * 32 + col];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int pg = 0; pg < GPB; ++pg) {
        constexpr int RPG = 16 / GPB;
        float sc[RPG], vv[RPG];
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < RPG; ++j) {
          sc[j] = sacc[cb][rb][pg * RPG + j] + bs;
          vv[j] = fmaxf(fmaf(vacc[cb][rb][pg * RPG + j], g, bt), 0.f);
          m = fmaxf(m, sc[j]);
        }
        m = fmaxf(m, other_half(m));
        float den = 0.f, num = 0.f;
#pragma unroll
        for (int j = 0; j < RPG; ++j) {
          const float e = __expf(sc[j] - m);
          den += e;
          num = fmaf(e, vv[j], num);
        }
        den += other_half(den);
        num += other_half(num);
        const int rbase = row0 + wave * 64 + rb * 32 + pg * KN;
        if (half == 0 && rbase < a.rows && cob0 + cb < a.n_cob) {
          const T v = (T)(num / den);
          reinterpret_cast<T *>(a.out)[(size_t)(rbase >> KLOG) * a.out_ld + (cob0 + cb) * 32 + col] = v;
          if (a.out_cm)
            reinterpret_cast<T *>(a.out_cm)[((size_t)(cob0 + cb) * (a.rows >> KLOG) + (rbase >> KLOG)) * 32 + col] = v;
          if (a.out2 && (cob0 + cb) * 32 + col < a.out2_n)
            reinterpret_cast<T *>(a.out2)[(size_t)(rbase >> KLOG) * a.out2_ld + (cob0 + cb) * 32 + col] = v;
        }
      }
  }
  SLIDE_STAMP(a, 6);
#ifdef SLIDE_TIMELINE
  if (a.dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); SLIDE_STAMP(a, 7); }
#endif
}

// ------------------------------------------------------------------------------------------------ points
__device__ __forceinline__ float sqdist3(float ax, float ay, float az, float bx, float by, float bz) {
#pragma clang fp contract(off)
  const float dx = ax - bx, dy = ay - by, dz = az - bz;
  return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

// Per sample (16 latent points): split x into xyz + features [x[3:], xyz] (attach_position_to_input_feature,
// pointnet2_with_pcld_condition.py:332-346) and build the full 16x16 neighbour table sorted by
// (squared distance, index) -- the K=16 and K=8 queries of every SA / FP level are prefixes of it
// (knn_points semantics, oracle/ops_cpu.c ora_knn_points).
template <typename T>
__global__ __launch_bounds__(256) void prep_points_kernel(int cx, int ldf, const float *__restrict__ x,
                                                          float *__restrict__ xyz, T *__restrict__ feat0,
                                                          int *__restrict__ kidx, float *__restrict__ kd2,
                                                          T *__restrict__ feat0_cm, const SlidePrepCopy *__restrict__ copies,
                                                          int n_copies, float *__restrict__ kw) {
  __shared__ float sp[48];
  __shared__ float sd[16][17];
  __shared__ float ssort[16][17];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float *xb = x + (size_t)b * 16 * cx;
  if (tid < 48) {
    const float v = xb[(tid / 3) * cx + tid % 3];
    sp[tid] = v;
    xyz[(size_t)b * 48 + tid] = v;
  }
  const int nf = cx - 3;
  for (int e = tid; e < 16 * cx; e += 256) {
    const int p = e / cx, c = e % cx;
    const T v = (T)(c < nf ? xb[p * cx + 3 + c] : xb[p * cx + (c - nf)]);
    feat0[((size_t)b * 16 + p) * ldf + c] = v;
    // second, chunk-major copy [c / 32][samples * 16][32] for the gather-on-load GEMM of the first SA block
    if (feat0_cm) feat0_cm[((size_t)(c >> 5) * gridDim.x * 16 + (size_t)b * 16 + p) * 32 + (c & 31)] = v;
  }
  // columns of later concatenation buffers that hold nothing but this sample's features / coordinates (the skip input and
  // the xyz columns of the FP blocks' second Mlp, the xyz columns of the head): written here instead of by COPY launches
  for (int q = 0; q < n_copies; ++q) {
    const SlidePrepCopy cp = copies[q];
    T *dst = reinterpret_cast<T *>(cp.dst);
    for (int e = tid; e < 16 * cp.n; e += 256) {
      const int p = e / cp.n, c = e - p * cp.n;
      const float v = cp.kind ? xb[p * cx + c] : (c < nf ? xb[p * cx + 3 + c] : xb[p * cx + (c - nf)]);
      dst[((size_t)b * 16 + p) * cp.ld + c] = (T)v;
    }
  }
  __syncthreads();
  const int i = tid >> 4, j = tid & 15;
  const float d = sqdist3(sp[i * 3], sp[i * 3 + 1], sp[i * 3 + 2], sp[j * 3], sp[j * 3 + 1], sp[j * 3 + 2]);
  sd[i][j] = d;
  __syncthreads();
  int rank = 0;
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) {
    const float o = sd[i][jj];
    rank += (o < d || (o == d && jj < j)) ? 1 : 0;
  }
  kidx[((size_t)b * 16 + i) * 16 + rank] = j;
  kd2[((size_t)b * 16 + i) * 16 + rank] = d;
  if (kw) {  // group_knn's interpolation weights of the 8 nearest, from SQUARED distances (pointnet2_utils.py:510-513)
#pragma clang fp contract(off)
    ssort[i][rank] = d;
    __syncthreads();
    float norm = 0.f;
    for (int kk = 0; kk < 8; ++kk) norm += 1.0f / (ssort[i][kk] + 1e-8f);
    kw[((size_t)b * 16 + i) * 16 + j] = j < 8 ? (1.0f / (ssort[i][j] + 1e-8f)) / norm : 0.f;
  }
}

// QueryAndGroup feature assembly ('nn', use_xyz, abs + center coordinates; pointnet2_utils.py:383-430):
// g[b][p*K+k][:] = [feat[nbr][0:C], xyz[nbr]-xyz[p], xyz[nbr], xyz[p], 0-pad]
// group_knn feature assembly (pointnet2_utils.py:497-524), FP = true:
// g[b][p*K+k][:] = [feat[nbr][0:C], d2, w, xyz[nbr], xyz[nbr]-xyz[p], xyz[p], 0-pad], w from squared distances
// One thread moves 8 channels (16 bytes in fp16): the gathered feature rows are copied as whole vectors, only the
// chunk that holds the coordinate channels is assembled element-wise.
template <typename T, bool FP>
__global__ __launch_bounds__(256) void assemble_kernel(int C, int ldf, int ldg, int K, int nch_log2, int bulk_blocks,
                                                       int c_begin, int ld_out,
                                                       const float *__restrict__ xyz,
                                                       const T *__restrict__ feat, const int *__restrict__ kidx,
                                                       const float *__restrict__ kd2, T *__restrict__ g) {
#pragma clang fp contract(off)
  // one thread per (row, 16-byte piece), no integer division.  Blocks y < bulk_blocks copy the whole-vector pieces of
  // the gathered feature rows (piece = low `nch_log2` bits of the thread id); the remaining blocks assemble the pieces
  // that hold coordinate channels element-wise -- kept in separate waves so the copy waves never diverge into that path.
  const int b = blockIdx.x;
  const int npx = 16 * K;
  const int nch = ldg / 8;
  const int nbulk = (sizeof(T) * ldf % 16 == 0) ? C / 8 : 0, ntail = nch - nbulk;
  const float *px = xyz + (size_t)b * 48;
  {
    int pxl, c0;
    if ((int)blockIdx.y < bulk_blocks) {
      const int e = blockIdx.y * 256 + threadIdx.x;
      pxl = e >> nch_log2;
      c0 = c_begin + (e & ((1 << nch_log2) - 1)) * 8;  // c_begin > 0: the leading columns are gathered by the GEMM itself
      if (pxl >= npx || c0 >= nbulk * 8) return;
    } else {
      const int e = (blockIdx.y - bulk_blocks) * 256 + threadIdx.x;
      pxl = e / ntail;
      c0 = (nbulk + (e - pxl * ntail)) * 8;
      if (pxl >= npx) return;
    }
    const int p = pxl / K, k = pxl - p * K;
    const size_t o = ((size_t)b * 16 + p) * 16;
    const int nb = kidx[o + k];
    const T *frow = feat + ((size_t)b * 16 + nb) * ldf;
    T *dst = g + ((size_t)b * npx + pxl) * ld_out + (c0 - c_begin);
    if (c0 + 8 <= C && sizeof(T) * ldf % 16 == 0) {
      if (sizeof(T) == 2) {
        *reinterpret_cast<float4 *>(dst) = *reinterpret_cast<const float4 *>(frow + c0);
      } else {
        *reinterpret_cast<float4 *>(dst) = *reinterpret_cast<const float4 *>(frow + c0);
        *reinterpret_cast<float4 *>(dst + 4) = *reinterpret_cast<const float4 *>(frow + c0 + 4);
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = c0 + j;
      float v = 0.f;
      if (c < C) v = (float)frow[c];
      else if (!FP) {
        if (c < C + 3) v = px[nb * 3 + (c - C)] - px[p * 3 + (c - C)];
        else if (c < C + 6) v = px[nb * 3 + (c - C - 3)];
        else if (c < C + 9) v = px[p * 3 + (c - C - 6)];
      } else {
        if (c == C) v = kd2[o + k];
        else if (c == C + 1) {
          float norm = 0.f;
          for (int kk = 0; kk < K; ++kk) norm += 1.0f / (kd2[o + kk] + 1e-8f);
          v = (1.0f / (kd2[o + k] + 1e-8f)) / norm;
        } else if (c < C + 5) v = px[nb * 3 + (c - C - 2)];
        else if (c < C + 8) v = px[nb * 3 + (c - C - 5)] - px[p * 3 + (c - C - 5)];
        else if (c < C + 11) v = px[p * 3 + (c - C - 8)];
      }
      dst[j] = (T)v;
    }
  }
}

// GroupNorm over a channel-concatenated tensor whose groups straddle producers (attention weight_conv.1,
// attention.py:45-47): per-sample channel sums -> per-channel scale / shift applied by the consumer GEMM.
__global__ __launch_bounds__(256) void finalize_gn_kernel(int B, int C, int bs, float inv_count,
                                                          const float *__restrict__ sum, const float *__restrict__ sq,
                                                          const int *__restrict__ gid, const int *__restrict__ gstart,
                                                          const int *__restrict__ gend, const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, float *__restrict__ scale,
                                                          float *__restrict__ shift) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= B * C) return;
  const int b = e / C, c = e - b * C;
  const int g = gid[c];
  float sc = 1.f, sh = 0.f;
  if (g >= 0) {
    float S = 0.f, SS = 0.f;
    for (int cc = gstart[g]; cc < gend[g]; ++cc) {
      S += sum[(size_t)b * bs + cc];
      SS += sq[(size_t)b * bs + cc];
    }
    const float mean = S * inv_count;
    const float var = fmaxf(SS * inv_count - mean * mean, 0.f);
    const float rstd = 1.0f / sqrtf(var + GN_EPS);
    sc = gamma[c] * rstd;
    sh = beta[c] - mean * sc;
  }
  scale[(size_t)b * bs + c] = sc;
  shift[(size_t)b * bs + c] = sh;
}

// out[bp][c] = sum_k softmax_k(S[bp*K+k][c]) * V[bp*K+k][c]     (attention.py:90-95, mask == all ones)
template <int K, typename T>
__global__ __launch_bounds__(256) void attn_combine_kernel(int nbp, int C, int ldS, int ldV, int ldo,
                                                           const T *__restrict__ S, const T *__restrict__ V,
                                                           T *__restrict__ out) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= nbp * C) return;
  const int bp = e / C, c = e - bp * C;
  float s[K];
  float m = -INFINITY;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    s[k] = (float)S[((size_t)bp * K + k) * ldS + c];
    m = fmaxf(m, s[k]);
  }
  float den = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    s[k] = expf(s[k] - m);
    den += s[k];
  }
  float o = 0.f;
  const float inv = 1.0f / den;
#pragma unroll
  for (int k = 0; k < K; ++k) o += (float)V[((size_t)bp * K + k) * ldV + c] * (s[k] * inv);
  out[(size_t)bp * ldo + c] = (T)o;
}

template <typename TS, typename TD>
__global__ __launch_bounds__(256) void copy_cols_kernel(int rows, int n, int src_ld, int dst_ld,
                                                        const TS *__restrict__ src, TD *__restrict__ dst) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= rows * n) return;
  const int r = e / n, c = e - r * n;
  dst[(size_t)r * dst_ld + c] = (TD)(float)src[(size_t)r * src_ld + c];
}

// Generic GroupNorm for the module-level compatibility path: x is NCHW (B, C, HW) fp32 like the reference's tensors;
// the first n_norm channels are normalised in G groups (MyGroupNorm, pointnet2_modules.py:24-42), the rest copied.
// One workgroup per (sample, group): a group's channels are one contiguous run of gs*HW floats.
__global__ __launch_bounds__(256) void group_norm_nchw_kernel(int C, int HW, int G, int n_norm, int relu,
                                                              const float *__restrict__ x, const float *__restrict__ gamma,
                                                              const float *__restrict__ beta, float *__restrict__ y) {
  __shared__ float red[2][4];
  const int b = blockIdx.y, g = blockIdx.x, tid = threadIdx.x;
  const int gs = n_norm / G;
  const size_t base = ((size_t)b * C + (size_t)g * gs) * HW;
  const int n = gs * HW;
  if (g == G) {  // pass-through tail channels
    const size_t tb = ((size_t)b * C + n_norm) * HW;
    for (int i = tid; i < (C - n_norm) * HW; i += 256) y[tb + i] = relu ? fmaxf(x[tb + i], 0.f) : x[tb + i];
    return;
  }
  float s = 0.f, ss = 0.f;
  for (int i = tid; i < n; i += 256) {
    const float v = x[base + i];
    s += v; ss += v * v;
  }
  for (int off = 32; off >= 1; off >>= 1) { s += __shfl_xor(s, off); ss += __shfl_xor(ss, off); }
  if ((tid & 63) == 0) { red[0][tid >> 6] = s; red[1][tid >> 6] = ss; }
  __syncthreads();
  s = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
  ss = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  const float mean = s / n;
  const float var = fmaxf(ss / n - mean * mean, 0.f);
  const float rstd = 1.0f / sqrtf(var + GN_EPS);
  for (int i = tid; i < n; i += 256) {
    const int c = g * gs + i / HW;
    float v = (x[base + i] - mean) * rstd * gamma[c] + beta[c];
    y[base + i] = relu ? fmaxf(v, 0.f) : v;
  }
}

__device__ __forceinline__ float swishf(float x) { return x * (1.0f / (1.0f + expf(-x))); }

// t-embedding path (pointnet2_ssg_sem.py:14-31 + pointnet2_with_pcld_condition.py:354-359) followed by every
// Mlp_plus_t_emb.fc (pointnet2_modules.py:139-143), one workgroup per distinct timestep.
// Weights are stored input-major ([in][out]) so lanes read consecutive outputs.
__global__ __launch_bounds__(256) void temb_kernel(int t_dim, int n_out, const float *__restrict__ ts,
                                                   const int *__restrict__ t_dev, const float *__restrict__ freq,
                                                   const float *__restrict__ w1, const float *__restrict__ b1,
                                                   const float *__restrict__ w2, const float *__restrict__ b2,
                                                   const float *__restrict__ wfc, const float *__restrict__ bfc,
                                                   float *__restrict__ out) {
  extern __shared__ float sm[];
  float *emb = sm;               // t_dim
  float *h1 = sm + t_dim;        // 4*t_dim
  float *h2 = h1 + 4 * t_dim;    // 4*t_dim
  const int b = blockIdx.x, tid = threadIdx.x;
  const float t = ts ? ts[b] : (float)t_dev[0];
  const int hd = t_dim / 2, H = 4 * t_dim;
  for (int k = tid; k < hd; k += 256) {
    const float arg = t * freq[k];
    emb[k] = sinf(arg);
    emb[hd + k] = cosf(arg);
  }
  __syncthreads();
  // input-major weights: lanes read consecutive outputs; 8 loads in flight per thread (the dims are multiples of 8)
  auto gemv = [&](const float *__restrict__ in, int n_in, const float *__restrict__ w, int n_o, int j) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int k = 0; k < n_in; k += 8) {
      const float w0 = w[(size_t)(k + 0) * n_o + j], w1_ = w[(size_t)(k + 1) * n_o + j], w2_ = w[(size_t)(k + 2) * n_o + j],
                  w3 = w[(size_t)(k + 3) * n_o + j], w4 = w[(size_t)(k + 4) * n_o + j], w5 = w[(size_t)(k + 5) * n_o + j],
                  w6 = w[(size_t)(k + 6) * n_o + j], w7 = w[(size_t)(k + 7) * n_o + j];
      a0 += in[k] * w0; a1 += in[k + 1] * w1_; a2 += in[k + 2] * w2_; a3 += in[k + 3] * w3;
      a0 += in[k + 4] * w4; a1 += in[k + 5] * w5; a2 += in[k + 6] * w6; a3 += in[k + 7] * w7;
    }
    return (a0 + a1) + (a2 + a3);
  };
  for (int j = tid; j < H; j += 256) h1[j] = swishf(b1[j] + gemv(emb, t_dim, w1, H, j));
  __syncthreads();
  for (int j = tid; j < H; j += 256) h2[j] = swishf(b2[j] + gemv(h1, H, w2, H, j));
  __syncthreads();
  for (int j = tid; j < n_out; j += 256) out[(size_t)b * n_out + j] = bfc[j] + gemv(h2, H, wfc, n_out, j);
}

// class embedding lookup (pointnet2_with_pcld_condition.py:363-365) + every Mlp_plus_t_emb.fc_condition
__global__ __launch_bounds__(256) void cond_kernel(int dim, int n_out, const int64_t *__restrict__ label,
                                                   const float *__restrict__ class_emb, const float *__restrict__ wfc,
                                                   const float *__restrict__ bfc, float *__restrict__ out) {
  extern __shared__ float sm[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float *ce = class_emb + (size_t)label[b] * dim;
  for (int k = tid; k < dim; k += 256) sm[k] = ce[k];
  __syncthreads();
  for (int j = tid; j < n_out; j += 256) {
    float acc = bfc[j];
    for (int k = 0; k < dim; ++k) acc += sm[k] * wfc[(size_t)k * n_out + j];
    out[(size_t)b * n_out + j] = acc;
  }
}

// ------------------------------------------------------------------------------------------------ DDPM updates
// (Philox noise, timestep advance, update_feat_element: ddpm_update.h)
// sampling() update (pointnet2/util.py:247-253): x = (x - c_eps[t]*eps)/sqrt_alpha[t]; t>0: x += sigma[t]*z
__global__ __launch_bounds__(256) void update_pos_kernel(int n, int eps_ld, uint32_t seed_lo, uint32_t seed_hi, float *__restrict__ x,
                                                         const float *__restrict__ eps, const float *__restrict__ noise,
                                                         int *__restrict__ t_dev, const float *__restrict__ c_eps,
                                                         const float *__restrict__ sqrt_alpha,
                                                         const float *__restrict__ sigma) {
#pragma clang fp contract(off)
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int t = t_dev[0], step = t_dev[1];
  if (e < n) {
    const int ep = eps_ld ? (e / 3) * eps_ld + e % 3 : e;  // eps rows may be padded (the plan's last GEMM output)
    float v = (x[e] - c_eps[t] * eps[ep]) / sqrt_alpha[t];
    if (t > 0) {
      const float z = noise ? noise[(size_t)step * n + e]
                            : philox_normal(seed_lo, seed_hi, (uint32_t)step, (uint32_t)e + (uint32_t)t_dev[4] * 48u, (uint32_t)t_dev[3]);
      v = v + sigma[t] * z;
    }
    x[e] = v;
  }
  advance_t_last_block(t_dev, t, step);
}

// denoising_step (pointnet2/diffusion_utils/diffusion.py:58-95) with the key-point channels re-clamped to the
// condition (:383-385): x0 = rc*x - rm1*eps [clamp]; mean = c1*x0 + c2*x; x = mean + [t>0] std*z
__global__ __launch_bounds__(256) void update_feat_kernel(int npts, int C, int kdim, int eps_ld, float clamp, uint32_t seed_lo,
                                                          uint32_t seed_hi, float *__restrict__ x,
                                                          const float *__restrict__ eps, const float *__restrict__ noise,
                                                          int *__restrict__ t_dev, const float *__restrict__ keypoint,
                                                          const float *__restrict__ rc, const float *__restrict__ rm1,
                                                          const float *__restrict__ c1, const float *__restrict__ c2,
                                                          const float *__restrict__ stdv,
                                                          const float *__restrict__ complete_x0,
                                                          const float *__restrict__ kmask, void *__restrict__ feat0, int ldf,
                                                          int half_out, const SlidePrepCopy *__restrict__ copies, int n_copies) {
#pragma clang fp contract(off)
  const int t = t_dev[0], step = t_dev[1];
  const uint32_t nonce = (uint32_t)t_dev[3];
  // the noise element index is GLOBAL: t_dev[4] = global index of the chain's first sample, so that a shape's noise does not
  // depend on how the run was split into ranks, batches and sub-batch chains
  const uint32_t eoff = (uint32_t)t_dev[4] * (uint32_t)(16 * C);
  // four elements per thread: a quarter of the blocks queue on the completion counter
#pragma unroll
  for (int j = 0; j < 4; ++j)
    update_feat_element(blockIdx.x * 1024 + j * 256 + threadIdx.x, npts, C, kdim, eps_ld, clamp, seed_lo, seed_hi, x, eps, noise,
                        t, step, nonce, eoff, complete_x0, kmask, keypoint, rc, rm1, c1, c2, stdv, feat0, ldf, half_out, copies,
                        n_copies);
  advance_t_last_block(t_dev, t, step);
}

// ------------------------------------------------------------------------------------------------ output head + DDPM update
// fc_lyaer (conv -> GroupNorm(32, 128) -> ReLU -> conv, pointnet2_with_pcld_condition.py:480-483) and the DDPM update of the
// sampler as ONE launch (SLIDE_OP_HEAD_UPDATE) instead of two small GEMM launches + the update kernel.  A workgroup owns 64
// rows (four samples); both layers' weights are tiny (<= 40 KB + 16 KB), so each wave loads the rows of ITS channel block
// straight into A-fragment registers at kernel start together with everything else the kernel reads (one L2 round trip), the
// hidden activation crosses the waves through LDS, and the prediction eps never leaves the registers: the lane that holds
// eps[row][channel] applies the update to x[row][channel] (and, for the feature DDPM with fixed key points, writes the per-point
// table / concatenation columns of the next step, as update_feat_kernel does).
typedef SlideHeadArgs HeadArgs;  // include/slide_engine.h

template <int K0MAX>  // k0 <= K0MAX (multiple of 32)
__global__ __launch_bounds__(256, 2) void head_update_kernel(HeadArgs a) {
#pragma clang fp contract(off)
  using T = _Float16;
  constexpr int LDX = K0MAX + 8, LDH = 128 + 8;
  __shared__ __attribute__((aligned(16))) T xs[64 * LDX];
  __shared__ __attribute__((aligned(16))) T hs[64 * LDH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
  const int row0 = blockIdx.x * 64;
  const int t = a.t_dev[0], step = a.t_dev[1];
  const uint32_t nonce = (uint32_t)a.t_dev[3];
  const uint32_t eoff = (uint32_t)a.t_dev[4] * (uint32_t)(16 * a.C);
  // ---- every global read of the two layers, issued together
  const int nk0 = a.k0 >> 4;  // 16-deep steps of layer 1
  f16x8 w0[K0MAX / 16];
  {
    const GLOBAL_AS T *wp = gptr<const T>((uint64_t)a.W0) + (size_t)(wave * 32 + col) * a.k0 + half * 8;
#pragma unroll
    for (int s2 = 0; s2 < K0MAX / 16; ++s2)
      if (s2 < nk0) w0[s2] = *(const GLOBAL_AS f16x8 *)(wp + s2 * 16);
  }
  const int cb2 = a.n1c == 2 ? (wave & 1) : 0, rb2 = a.n1c == 2 ? (wave >> 1) : wave;  // layer-2 block of this wave
  const bool l2 = rb2 < 2;
  f16x8 w1[8];
  if (l2) {
    const GLOBAL_AS T *wp = gptr<const T>((uint64_t)a.W1) + (size_t)(cb2 * 32 + col) * 128 + half * 8;
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) w1[s2] = *(const GLOBAL_AS f16x8 *)(wp + s2 * 16);
  }
  {
    const int ppr = a.k0 >> 3;
    for (int i = tid; i < 64 * ppr; i += 256) {
      const int r = i / ppr, pc = i - r * ppr;
      int grow = row0 + r;
      grow = grow < a.rows ? grow : a.rows - 1;
      *reinterpret_cast<u32x4 *>(xs + r * LDX + pc * 8) = *(const GLOBAL_AS u32x4 *)(gptr<const T>((uint64_t)a.X) + (size_t)grow * a.x_ld + pc * 8);
    }
  }
  float bia[16], gam[16], bet[16];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 b4 = *reinterpret_cast<const float4 *>(a.v0 + wave * 32 + 8 * q + 4 * half);
    const float4 g4 = *reinterpret_cast<const float4 *>(a.v0 + 128 + wave * 32 + 8 * q + 4 * half);
    const float4 t4 = *reinterpret_cast<const float4 *>(a.v0 + 256 + wave * 32 + 8 * q + 4 * half);
    bia[4 * q] = b4.x; bia[4 * q + 1] = b4.y; bia[4 * q + 2] = b4.z; bia[4 * q + 3] = b4.w;
    gam[4 * q] = g4.x; gam[4 * q + 1] = g4.y; gam[4 * q + 2] = g4.z; gam[4 * q + 3] = g4.w;
    bet[4 * q] = t4.x; bet[4 * q + 1] = t4.y; bet[4 * q + 2] = t4.z; bet[4 * q + 3] = t4.w;
  }
  float b1v[16];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 b4 = l2 ? *reinterpret_cast<const float4 *>(a.b1 + cb2 * 32 + 8 * q + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
    b1v[4 * q] = b4.x; b1v[4 * q + 1] = b4.y; b1v[4 * q + 2] = b4.z; b1v[4 * q + 3] = b4.w;
  }
  __syncthreads();
  // ---- layer 1: channel block `wave`, both 32-row blocks; D[channel][row]: lane = row, reg r = channel (r&3)+8(r>>2)+4 half
  f32x16 acc[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
#pragma unroll
  for (int s2 = 0; s2 < K0MAX / 16; ++s2)
    if (s2 < nk0) {
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const f16x8 xb = *reinterpret_cast<const f16x8 *>(xs + (rb * 32 + col) * LDX + s2 * 16 + half * 8);
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0[s2], xb, acc[rb], 0, 0, 0);
      }
    }
  // bias, GroupNorm(32, 128) = groups of four consecutive channels = the four registers 4q .. 4q+3 of a lane, statistics over
  // the sample's 16 rows = 16 lanes; ReLU; fp16 into LDS [row][channel]
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v[4], s = 0.f, ss = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = acc[rb][4 * q + j] + bia[4 * q + j];
        s += v[j];
        ss = fmaf(v[j], v[j], ss);
      }
      s = lane_group_sum<16>(s);
      ss = lane_group_sum<16>(ss);
      const float mean = s * (1.0f / 64.0f);
      const float var = fmaxf(ss * (1.0f / 64.0f) - mean * mean, 0.f);
      const float rstd = __builtin_amdgcn_rsqf(var + GN_EPS);
      f16x4 h;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float g = gam[4 * q + j] * rstd;
        h[j] = (T)fmaxf(fmaf(v[j], g, bet[4 * q + j] - mean * g), 0.f);
      }
      *reinterpret_cast<f16x4 *>(hs + (rb * 32 + col) * LDH + wave * 32 + 8 * q + 4 * half) = h;
    }
  }
  __syncthreads();
  // ---- layer 2 + update
  if (l2) {
    f32x16 e2;
#pragma unroll
    for (int r = 0; r < 16; ++r) e2[r] = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) {
      const f16x8 hb = *reinterpret_cast<const f16x8 *>(hs + (rb2 * 32 + col) * LDH + s2 * 16 + half * 8);
      e2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[s2], hb, e2, 0, 0, 0);
    }
    const int p = row0 + rb2 * 32 + col;
    if (p < a.rows) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = cb2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const float ev = e2[r] + b1v[r];
        if (a.eps_out && c < a.eps_ld) a.eps_out[(size_t)p * a.eps_ld + c] = ev;
        if (c >= a.C) continue;
        const int e = p * a.C + c;
        if (a.kind == 1) {
          update_feat_element(e, a.rows, a.C, a.kdim, 0, a.clamp, a.seed_lo, a.seed_hi, a.x, nullptr, a.noise, t, step, nonce, eoff,
                              a.complete_x0, a.kmask, a.keypoint, a.t0, a.t1, a.t2, a.t3, a.t4, a.feat0, a.ldf, a.half_out,
                              a.copies, a.n_copies, ev);
        } else {  // position DDPM: x = (x - c_eps[t] eps) / sqrt_alpha[t] (+ sigma[t] z)   (tables t0, t1, t2)
          float v = (a.x[e] - a.t0[t] * ev) / a.t1[t];
          if (t > 0) {
            const float z = a.noise ? a.noise[(size_t)step * a.rows * 3 + e]
                                    : philox_normal(a.seed_lo, a.seed_hi, (uint32_t)step, (uint32_t)e + (uint32_t)a.t_dev[4] * 48u, nonce);
            v = v + a.t2[t] * z;
          }
          a.x[e] = v;
        }
      }
    }
  }
  advance_t_last_block(a.t_dev, t, step);
}

__global__ void advance_t_kernel(int *t_dev) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    t_dev[0] -= 1;
    t_dev[1] += 1;
  }
}

// hipFuncSetAttribute is per device: the "already raised the dynamic-LDS limit" flags are kept per device so that one
// process may drive plans on several GPUs (first use of a kernel on each device must still happen outside stream capture
// and from one thread, as for any lazily initialised runtime state)
constexpr int SLIDE_MAX_DEVICES = 64;
inline int current_device_slot() {
  int d = 0;
  (void)hipGetDevice(&d);
  return d >= 0 && d < SLIDE_MAX_DEVICES ? d : 0;
}

template <int PREC, int NPXL, int CBW, bool PAIRRES = false>
int launch_gemm(const GemmArgs &a, hipStream_t s) {
  constexpr int LDK = TileT<PREC>::LDK;
  // (split mode: two fp16 planes per operand tile)
  const size_t aff = (PREC != SLIDE_PREC_F16 && NPXL >= 7 && a.in_scale) ? (size_t)(TM >> NPXL) * 2 * a.k_pad * 4 : 0;
  const size_t shm = (PREC == SLIDE_PREC_SPLIT ? (size_t)(TM + 32 * CBW) * LDK * 4 : 2 * (size_t)(TM + 32 * CBW) * LDK * sizeof(typename TileT<PREC>::T)) +
                     CBW * (sizeof(SlideEpi) + 96 * 4) + 16 + aff;
  if (shm > 160 * 1024) return -8;
  const int ntc = (a.n_cob + CBW - 1) / CBW, ntr = (a.rows + TM - 1) / TM;
  const int grid = ((ntr + 7) / 8) * 8 * ntc;
  static bool attr_done[SLIDE_MAX_DEVICES] = {};
  bool &attr_set = attr_done[current_device_slot()];
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_kernel<PREC, NPXL, CBW, PAIRRES>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_kernel<PREC, NPXL, CBW, PAIRRES>), dim3(grid), dim3(256), shm, s, a);
  return (int)hipGetLastError();
}

int launch_gemm_split_small(const GemmArgs &a, hipStream_t s) {
  constexpr int LDK = TileT<SLIDE_PREC_SPLIT>::LDK;
  const size_t shm = (size_t)2 * 4 * 64 * LDK * 2 + 2 * (sizeof(SlideEpi) + 96 * 4) + 32 + (a.in_scale ? (size_t)4 * 2 * a.k_pad * 4 : 0);
  if (shm > 64 * 1024) return -8;
  const int grid = ((a.rows + 63) / 64) * ((a.n_cob + 1) / 2);
  hipLaunchKernelGGL((gemm_split_small_kernel<4>), dim3(grid), dim3(256), shm, s, a);
  return (int)hipGetLastError();
}

template <int NPXL, int CBW, int NST, int BKT, bool AFF, bool GAT = false, bool PAIRRES = false>
int launch_gemm_glds(const GemmArgs &a, hipStream_t s) {
  constexpr int NSAMP = (1 << NPXL) >= TM ? 1 : TM >> NPXL;
  const size_t shm = (size_t)NST * (TM + (CBW < 2 ? 64 : 32 * CBW)) * BKT * 2 + CBW * (sizeof(SlideEpi) + 96 * 4) + 32 +
                     (AFF ? (size_t)NSAMP * 3 * a.k_pad * 2 : 0);
  if (shm > 80 * 1024 && BKT == 32 && NST <= 3) return -8;  // two workgroups per CU must fit
  if (shm > 160 * 1024) return -8;
  const int ntc = (a.n_cob + CBW - 1) / CBW, ntr = (a.rows + TM - 1) / TM;
  int grid = ((ntr + 7) / 8) * 8 * ntc;
  GemmArgs b = a;
  b.shm_bytes = (int)((shm + 15) & ~(size_t)15);
  if (b.sched && grid > 512 && NST <= 3) grid = 512;  // persistent: two resident workgroups per CU pull the tiles
  else b.sched = nullptr;
  static bool attr_done[SLIDE_MAX_DEVICES] = {};
  bool &attr_set = attr_done[current_device_slot()];
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_glds_kernel<NPXL, CBW, NST, BKT, AFF, GAT, PAIRRES>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (NST > 3 || BKT > 32) ? 160 * 1024 : 84 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_glds_kernel<NPXL, CBW, NST, BKT, AFF, GAT, PAIRRES>), dim3(grid), dim3(256), (size_t)b.shm_bytes, s, b);
  return (int)hipGetLastError();
}

template <int NPXL, bool AFF, bool GAT, bool PAIRRES = false>
int launch_gemm_occ3(const GemmArgs &a, hipStream_t s) {
  constexpr int NSAMP = (1 << NPXL) >= TM ? 1 : TM >> NPXL;
  const size_t shm = (size_t)2 * (TM + 64) * 32 * 2 + 2 * (sizeof(SlideEpi) + 96 * 4) + 32 + (AFF ? (size_t)NSAMP * 3 * a.k_pad * 2 : 0);
  if (shm > 53 * 1024) return -8;  // three workgroups per CU must fit
  const int ntc = (a.n_cob + 1) / 2, ntr = (a.rows + TM - 1) / TM;
  const int grid = ((ntr + 7) / 8) * 8 * ntc;
  GemmArgs b = a;
  b.shm_bytes = (int)((shm + 15) & ~(size_t)15);
  b.sched = nullptr;
  static bool attr_done[SLIDE_MAX_DEVICES] = {};
  bool &attr_set = attr_done[current_device_slot()];
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_glds_occ3_kernel<NPXL, AFF, GAT, PAIRRES>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 53 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_glds_occ3_kernel<NPXL, AFF, GAT, PAIRRES>), dim3(grid), dim3(256), (size_t)b.shm_bytes, s, b);
  return (int)hipGetLastError();
}

template <int NPXL, int CBW, int NST>
int launch_gemm_glds8(const GemmArgs &a, hipStream_t s) {
  constexpr int TNS8 = ((TM + 64 * CBW + 127) / 128) * 128 - TM;
  const size_t shm = (size_t)NST * (TM + TNS8) * 32 * 2 + 2 * CBW * (sizeof(SlideEpi) + 96 * 4) + 32;
  if (shm > 160 * 1024) return -8;
  const int ntc = (a.n_cob + 2 * CBW - 1) / (2 * CBW), ntr = (a.rows + TM - 1) / TM;
  const int grid = ((ntr + 7) / 8) * 8 * ntc;
  static bool attr_done[SLIDE_MAX_DEVICES] = {};
  bool &attr_set = attr_done[current_device_slot()];
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_glds8_kernel<NPXL, CBW, NST>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  GemmArgs b = a;
  b.sched = nullptr;
  b.shm_bytes = (int)shm;
  hipLaunchKernelGGL((gemm_glds8_kernel<NPXL, CBW, NST>), dim3(grid), dim3(512), shm, s, b);
  return (int)hipGetLastError();
}


template <int NST, bool AFF>
int launch_gemm_small_t(const GemmArgs &a, hipStream_t s) {
  const size_t shm = (size_t)4 * NST * 6144 + 2 * (sizeof(SlideEpi) + 96 * 4) + 32 + (AFF ? (size_t)4 * 2 * a.k_pad * 2 + 1024 : 0);
  const int grid = ((a.rows + 63) / 64) * ((a.n_cob + 1) / 2);
  static bool attr_done[SLIDE_MAX_DEVICES] = {};
  bool &attr_set = attr_done[current_device_slot()];
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_small_kernel<NST, AFF>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_small_kernel<NST, AFF>), dim3(grid), dim3(256), shm, s, a);
  return (int)hipGetLastError();
}

template <bool FP>
int launch_pair_first_t(const GemmArgs &a, const PairArgs &pa, hipStream_t s) {
  const int grid = ((a.rows + 63) / 64) * ((a.n_cob + 1) / 2);
  // (51 KB -> THREE workgroups per CU: the four samples' coordinates behind the rings; the FP blocks' neighbour / distance / weight
  //  slots wait in registers and land in the dead ring area after the K loop, gemm_small.h)
  const size_t shm = (size_t)4 * 2 * 6144 + 2 * (sizeof(SlideEpi) + 96 * 4) + 32 + 192 * 4;
  static bool attr_done[SLIDE_MAX_DEVICES] = {};
  bool &attr_set = attr_done[current_device_slot()];
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&pair_first_kernel<FP>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((pair_first_kernel<FP>), dim3(grid), dim3(256), shm, s, a, pa);
  return (int)hipGetLastError();
}

// SLIDE_OP_PAIR_FIRST (include/slide_engine.h)
int run_pair_first(const SlideOp &o, hipStream_t s) {
  GemmArgs a = GemmArgs();
  a.X = o.p[0]; a.W = o.p[1]; a.epi = (const SlideEpi *)o.p[2];
  a.aff_tps = 1;
  a.rows = o.i[0]; a.x_ld = o.i[1]; a.k_pad = o.i[2]; a.n_cob = o.i[3];
  resolve_epi(a);
  a.dbg = (unsigned long long *)o.p[13];
  PairArgs pa;
  pa.pair_cob0 = o.i[4]; pa.ld = o.i[5];
  pa.xyz = (const float *)o.p[3]; pa.wa = (const float *)o.p[4]; pa.wb = (const float *)o.p[5];
  pa.ta = (_Float16 *)o.p[6]; pa.tb = (_Float16 *)o.p[7];
  pa.nbr = (const int *)o.p[8]; pa.d2t = (const float *)o.p[9]; pa.wt = (const float *)o.p[10];
  pa.vv_in = (const float *)o.p[11]; pa.vv_out = (float *)o.p[12];
  if (a.k_pad % BK || a.x_ld % 8 || a.rows <= 0 || a.n_cob <= 0 || a.rows % 16 || pa.pair_cob0 < 0 || pa.pair_cob0 > a.n_cob ||
      pa.ld != (a.n_cob - pa.pair_cob0) * 32)
    return -3;
  if (o.i[6] == 8) {
    if (!pa.nbr || !pa.d2t || !pa.wt || !pa.vv_in || !pa.vv_out) return -3;
    return launch_pair_first_t<true>(a, pa, s);
  }
  return launch_pair_first_t<false>(a, pa, s);
}

int launch_gemm_small(const GemmArgs &a, hipStream_t s) {
  const int grid = ((a.rows + 63) / 64) * ((a.n_cob + 1) / 2);
  // two stages per wave (64 KB of LDS: the size of the partial-sum exchange) rather than three (96 KB): the workgroup
  // then fits a CU beside two 41 KB GEMM workgroups of the other chains (0.913 vs 0.927 ms/step); a.stagger == 5 keeps
  // three stages on single-round grids, for A/B timing
#ifdef SLIDE_EXPERIMENTS
  const bool three = grid <= 256 && a.stagger == 5;
  if (three) return a.in_scale ? launch_gemm_small_t<3, true>(a, s) : launch_gemm_small_t<3, false>(a, s);
#endif
  (void)grid;
  return a.in_scale ? launch_gemm_small_t<2, true>(a, s) : launch_gemm_small_t<2, false>(a, s);
}

template <bool AFF>
int launch_gemm_attend(const GemmArgs &a, hipStream_t s) {
  const size_t shm = (size_t)2 * (TM + 64) * 32 * 2 + 2 * (sizeof(SlideEpi) + 96 * 4) + 32 + (AFF ? (size_t)3 * a.k_pad * 2 : 0);
  if (shm > 53 * 1024) return -8;  // three workgroups per CU must fit
  const int ntc = (a.n_cob + 1) / 2, ntr = (a.rows + TM - 1) / TM;
  const int grid = ((ntr + 7) / 8) * 8 * ntc;
  GemmArgs b = a;
  b.shm_bytes = (int)((shm + 15) & ~(size_t)15);
  b.sched = nullptr;
  static bool attr_done[SLIDE_MAX_DEVICES] = {};
  bool &attr_set = attr_done[current_device_slot()];
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_attend_kernel<AFF>), hipFuncAttributeMaxDynamicSharedMemorySize, 53 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_attend_kernel<AFF>), dim3(grid), dim3(256), (size_t)b.shm_bytes, s, b);
  return (int)hipGetLastError();
}

// SLIDE_OP_GEMM_ATTEND (include/slide_engine.h)
int run_gemm_attend(const SlideOp &o, hipStream_t s) {
  GemmArgs a = GemmArgs();
  a.X = o.p[0]; a.W = o.p[1]; a.epi = (const SlideEpi *)o.p[2];
  a.in_scale = (const float *)o.p[3]; a.in_shift = (const float *)o.p[4];
  a.aff_tps = 1;
  if (a.in_scale) {
    a.in_add = (const float *)o.p[11];
    a.aff_tps = (int)o.f[1] > 1 ? (int)o.f[1] : 1;
    a.add_bs = (int)o.f[2];
    a.add_n = (int)o.f[3] >> 1;
    a.aff_relu = (int)o.f[3] & 1;
  }
  a.at_V = o.p[5]; a.at_out = o.p[6]; a.at_counts = (const int *)o.p[7]; a.at_vss = (const float *)o.p[8];
  a.rows = o.i[0]; a.x_ld = o.i[1]; a.k_pad = o.i[2]; a.n_cob = o.i[3]; a.in_bs = o.i[5];
  resolve_epi(a);
  const int K = o.i[4];
  a.at_ldv = o.i[6]; a.at_ldo = o.i[7]; a.at_pps = o.i[8] > 0 ? o.i[8] : 1; a.at_vrelu = o.i[9]; a.at_C = o.i[10];
  a.at_klog2 = K == 4 ? 2 : K == 8 ? 3 : K == 16 ? 4 : K == 32 ? 5 : -1;
  if (a.at_klog2 < 0 || a.rows <= 0 || a.rows % K || a.k_pad % BK || a.x_ld % 8 || a.n_cob <= 0 || !a.X || !a.W || !a.epi || !a.at_V ||
      !a.at_out || a.at_ldv % 4 || a.at_ldo % 4 || a.at_ldv < a.n_cob * 32 || a.at_ldo < a.n_cob * 32 || (a.in_scale && !a.in_shift))
    return -3;
  return a.in_scale ? launch_gemm_attend<true>(a, s) : launch_gemm_attend<false>(a, s);
}

int run_gemm(const SlideOp &o, hipStream_t s) {
  GemmArgs a;
  a.X = o.p[0]; a.W = o.p[1]; a.epi = (const SlideEpi *)o.p[2];
  a.in_scale = (const float *)o.p[3]; a.in_shift = (const float *)o.p[4];
  // deferred normalisation (no gather): p[11] = add vectors, f[1] = tiles per sample, f[2] = add_bs, f[3] = 2 * add_n + relu
  a.in_add = nullptr; a.aff_relu = 0; a.add_bs = 0; a.add_n = 0; a.aff_tps = 1;
  if (a.in_scale && !o.p[8]) {
    a.in_add = (const float *)o.p[11];
    a.aff_tps = (int)o.f[1] > 1 ? (int)o.f[1] : 1;
    a.add_bs = (int)o.f[2];
    a.add_n = (int)o.f[3] >> 1;
    a.aff_relu = (int)o.f[3] & 1;
  }
  a.dbg = (unsigned long long *)o.p[5];
  a.stagger = (int)(o.f[0] * 100.f);
  a.sched = (int *)o.p[7];
  a.gfeat = o.p[8]; a.gidx = (const int *)o.p[9];
  a.gn_fin = (const SlideGnFin *)o.p[6];
  a.gx_d2 = (const float *)o.p[12]; a.gx_w = (const float *)o.p[13];  // PAIR_NBR residual (with p[9] the neighbour table)
  a.gx_ta = a.gx_tb = nullptr; a.gx_vv = nullptr; a.gx_add_idx = nullptr;
  a.g_nsplit = (int)o.f[1]; a.g_ldf = (int)o.f[2]; a.g_klog2 = (int)o.f[3];
  a.rows = o.i[0]; a.x_ld = o.i[1]; a.k_pad = o.i[2]; a.n_cob = o.i[3]; a.in_bs = o.i[5];
  a.ch_epi = nullptr; a.ch_n_cob = 0;
  resolve_epi(a);
  const int npxl = o.i[4], prec = o.i[6], cbw = o.i[7], glds = o.i[8] & 1;
  a.w_cm = (o.i[8] >> 1) & 1;  // chunk-major weights (ring kernels of the 128 / 256-row samples only)
  if ((o.i[8] >> 2) & 1) {     // a block of this GEMM carries a PAIR residual: the instantiations compiled for it
    if (prec == SLIDE_PREC_SPLIT && !glds && !a.in_scale && !o.p[8] && !o.p[10] && cbw == 2) {  // float tables (round 5)
      if (a.k_pad % BK || a.x_ld % 4 || a.rows <= 0 || a.n_cob <= 0) return -3;
      if (npxl == 8) return launch_gemm<SLIDE_PREC_SPLIT, 8, 2, true>(a, s);
      if (npxl == 7) return launch_gemm<SLIDE_PREC_SPLIT, 7, 2, true>(a, s);
      return -12;
    }
    if (!glds || prec != SLIDE_PREC_F16 || a.in_scale || o.p[8] || o.p[10]) return -12;
    // (two workgroups per CU, 256 registers: the three-workgroup form spills with the pair address arithmetic)
    if (npxl == 8) return launch_gemm_glds<8, 2, 3, 32, false, false, true>(a, s);
    if (npxl == 7) return launch_gemm_glds<7, 2, 3, 32, false, false, true>(a, s);
    return -12;
  }
  if (a.w_cm && (!glds || (npxl != 7 && npxl != 8) || o.p[10] || o.i[9] == 1)) return -11;
  if (a.k_pad % BK || a.x_ld % 8 || a.rows <= 0 || a.n_cob <= 0) return -3;
  // fp16 16-row launches: split-K small-launch kernel, with or without the input affine (i[9] == 3 keeps the 256-row
  // kernels, for A/B timing)
  // (up to 1024 tiles with the statistics finalisation, 8192 without: the wide per-point GEMMs of the pair decomposition --
  //  N = 1056 .. 1568 -- stay on this spill-free kernel instead of the 256-row ring tiles, which spill at 16 rows per sample)
  // (round 4: no tile limit without the finalisation -- a chain of 2048 samples used to fall back to the 256-row ring tiles,
  //  which spill at 16 rows per sample; those are experiments-build kernels now)
  if (prec == SLIDE_PREC_F16 && npxl == 4 && o.i[9] != 3 &&
      (!a.gn_fin || ((a.rows + 63) / 64) * ((a.n_cob + 1) / 2) <= 1024))
    return launch_gemm_small(a, s);
  if (a.gn_fin) return -10;  // only the small-launch kernel finalises statistics
  // X-stationary kernel (SlideOp.p[10] = the weights as MFMA A fragments): one workgroup per row tile computes every
  // column tile from an LDS-resident X.  i[9] == 5 keeps the ring kernels, for A/B timing.
#ifndef SLIDE_EXPERIMENTS
  if (o.p[10]) return SLIDE_ST_EXPERIMENT;
#else
  if (o.p[10] && glds && prec == SLIDE_PREC_F16 && o.i[9] != 5 && (npxl == 8 || npxl == 7) && !(a.gfeat && a.in_scale)) {
    int st = -8;
    const bool aff = a.in_scale != nullptr, gat = a.gfeat != nullptr;
    st = slide_launch_gemm_xs(a, npxl, cbw, aff, gat, o.i[9] >= 11 && o.i[9] <= 13 ? o.i[9] - 10 : 0, s);
    if (st != -8) return st;  // -8: the X tile does not fit the LDS -> ring kernels
  }
#endif
  if (glds) {
    if (prec != SLIDE_PREC_F16) return -7;
#ifndef SLIDE_EXPERIMENTS
    // PRODUCT build: 256 x 64 tiles at three workgroups per CU (plain or with the input affine), the two-workgroup form of the
    // affine tile where its vectors do not fit beside three -- what the default DDPM plans and the module path (decode, encode)
    // dispatch.  Every other ring variant is an experiments-build kernel.
    if (o.i[9] != 0 || cbw != 2 || a.gfeat || a.stagger == 7 || (npxl != 7 && npxl != 8)) return SLIDE_ST_EXPERIMENT;
    {
      int st3 = -8;
      if (npxl == 8) st3 = a.in_scale ? launch_gemm_occ3<8, true, false>(a, s) : launch_gemm_occ3<8, false, false>(a, s);
      else if (!a.in_scale) return SLIDE_ST_EXPERIMENT;  // (128-row samples on stored inputs: the round-2 plan's FP blocks)
      else return launch_gemm_glds<7, 2, 3, 32, true>(a, s);
      if (st3 != -8) return st3;
      if (a.in_scale) return launch_gemm_glds<8, 2, 3, 32, true>(a, s);
      return -4;
    }
#else
    // i[9]: 0 = BK 32, three stages (two workgroups / CU); 1 = BK 64 (full 128-B lines), three stages (one / CU)
    const int wide = o.i[9] == 1 && (a.k_pad % 64 == 0) && !a.in_scale;
#define GCASE(L, C)                                                                                        \
  if (npxl == L && cbw == C)                                                                               \
    return wide ? launch_gemm_glds<L, C, 3, 64, false>(a, s) : launch_gemm_glds<L, C, 3, 32, false>(a, s)
#define ACASE(L, C) if (npxl == L && cbw == C) return launch_gemm_glds<L, C, 3, 32, true>(a, s)
    // launches of at most one workgroup per CU (the 16-row per-point GEMMs) are bound by the latency of their K loop:
    // a 7-stage ring keeps five chunks in flight instead of one
    if (npxl == 4 && cbw == 2 && !a.in_scale && !wide &&
        ((a.rows + TM - 1) / TM) * ((a.n_cob + 1) / 2) <= 256 && a.k_pad >= 128)
      return launch_gemm_glds<4, 2, 7, 32, false>(a, s);
    // wide outputs: eight-wave 256 x 256 tiles when the channel blocks fill them and enough tiles remain for the chip
    if (o.i[9] == 2 && cbw == 4 && !a.in_scale && a.n_cob % 8 == 0 &&
        ((a.rows + TM - 1) / TM) * (a.n_cob / 8) >= 256) {
      if (npxl == 8) return launch_gemm_glds8<8, 4, 4>(a, s);
      if (npxl == 7) return launch_gemm_glds8<7, 4, 4>(a, s);
    }
    // narrow outputs on a grid that does not fill the chip: the same 256 x 64 tile on eight waves (one channel block
    // per wave) halves each wave's epilogue
    if (o.i[9] == 4 && cbw == 2 && !a.in_scale && ((a.rows + TM - 1) / TM) * ((a.n_cob + 1) / 2) <= 512) {
      if (npxl == 8) return launch_gemm_glds8<8, 1, 3>(a, s);
      if (npxl == 7) return launch_gemm_glds8<7, 1, 3>(a, s);
    }
    // 64-channel tiles: three workgroups per CU (two-stage ring of 41 KB, 168-VGPR budget) instead of two on a three-stage
    // ring -- 8-13 % faster per launch at N >= 512 and, with four chains in flight, 2.5 % per step (0.921 vs 0.944 ms)
    // (a.stagger == 7: the two-workgroup form, for A/B timing)
    if (cbw == 2 && !wide && !(a.gfeat && a.in_scale) && a.stagger != 7 && (npxl == 7 || npxl == 8)) {
      int st3 = -8;
#define OCASE(L, A, G) if (npxl == L && (a.in_scale != nullptr) == A && (a.gfeat != nullptr) == G) st3 = launch_gemm_occ3<L, A, G>(a, s)
      OCASE(7, false, false); OCASE(8, false, false); OCASE(7, true, false); OCASE(8, true, false);
      OCASE(7, false, true); OCASE(8, false, true);
#undef OCASE
      if (st3 != -8) return st3;
    }
    if (a.gfeat) {  // gathered grouped input (first GEMM of an SA / FP block)
      if (a.in_scale || wide) return -4;
      if (npxl == 7 && cbw == 2) return launch_gemm_glds<7, 2, 3, 32, false, true>(a, s);
      if (npxl == 8 && cbw == 2) return launch_gemm_glds<8, 2, 3, 32, false, true>(a, s);
      if (npxl == 7 && cbw == 4) return launch_gemm_glds<7, 4, 3, 32, false, true>(a, s);
      if (npxl == 8 && cbw == 4) return launch_gemm_glds<8, 4, 3, 32, false, true>(a, s);
      return -4;
    }
    if (a.in_scale) { ACASE(7, 2); ACASE(8, 2); ACASE(7, 4); ACASE(8, 4); return -4; }
    if (cbw == 1 && !wide) {
      if (npxl == 7) return launch_gemm_glds<7, 1, 3, 32, false>(a, s);
      if (npxl == 8) return launch_gemm_glds<8, 1, 3, 32, false>(a, s);
      return -4;
    }
    GCASE(4, 2); GCASE(7, 2); GCASE(8, 2); GCASE(4, 4); GCASE(7, 4); GCASE(8, 4);
#undef ACASE
#undef GCASE
    return -4;
#endif
  }
  // split mode, 16-row samples (and RAW-epilogue launches that ask for them): 64-row tiles
  if (prec == SLIDE_PREC_SPLIT && npxl == 4 && cbw == 2 && o.i[9] != 3) return launch_gemm_split_small(a, s);
#define CASE(P, L, C) if (prec == P && npxl == L && cbw == C) return launch_gemm<P, L, C>(a, s)
  CASE(SLIDE_PREC_F32, 4, 2); CASE(SLIDE_PREC_F32, 7, 2); CASE(SLIDE_PREC_F32, 8, 2);
  CASE(SLIDE_PREC_SPLIT, 4, 2); CASE(SLIDE_PREC_SPLIT, 7, 2); CASE(SLIDE_PREC_SPLIT, 8, 2);
#ifdef SLIDE_EXPERIMENTS
  CASE(SLIDE_PREC_F16, 4, 2); CASE(SLIDE_PREC_F16, 7, 2); CASE(SLIDE_PREC_F16, 8, 2);
  CASE(SLIDE_PREC_F16, 4, 4); CASE(SLIDE_PREC_F16, 7, 4); CASE(SLIDE_PREC_F16, 8, 4);
#else
  if (prec == SLIDE_PREC_F16) return SLIDE_ST_EXPERIMENT;  // register-staged fp16 GEMM [SLIDE_GLDS=0]
#endif
#undef CASE
  return -4;
}

// batched fp32 transpose through a 32x33 LDS tile (module-level path: NCHW activations <-> the GEMM's row-major
// [pixel][channel] matrices); coalesced on both sides
template <typename TO>
__global__ __launch_bounds__(256) void transpose_kernel(int R, int C, int in_ld, int out_ld, long long in_bs,
                                                        long long out_bs, const float *__restrict__ in,
                                                        TO *__restrict__ out) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  in += (size_t)b * in_bs;
  out += (size_t)b * out_bs;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    if (r < R && c < C) tile[ty + 8 * i][tx] = in[(size_t)r * in_ld + c];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (r < R && c < C) out[(size_t)c * out_ld + r] = (TO)tile[tx][ty + 8 * i];
  }
}

int run_attn_tail(const SlideOp &o, hipStream_t s) {
  AttnTailArgs a;
  a.X1 = o.p[0]; a.W1 = o.p[1]; a.X2 = o.p[2]; a.W2 = o.p[3]; a.out = o.p[4]; a.vec = (const float *)o.p[5];
  a.out_cm = o.p[6];
  a.dbg = (unsigned long long *)o.p[8];
  a.out2 = o.p[7]; a.out2_ld = (int)o.f[2]; a.out2_n = (int)o.f[3];
  a.rows = o.i[0]; a.x1_ld = o.i[1]; a.k1 = o.i[2]; a.x2_ld = o.i[3]; a.k2 = o.i[4]; a.n_cob = o.i[5];
  a.gs = o.i[7]; a.n_norm = o.i[8]; a.out_ld = o.i[9];
  a.inv_count = o.f[0];
  a.w_cm = ((int)o.f[1] & 1) != 0;
  a.x_fm = ((int)o.f[1] & 16) != 0;
  static const int tail_abl = [] { const char *e = getenv("SLIDE_TAIL_ABL"); return e ? atoi(e) : 0; }();
  a.abl = tail_abl;
  const int npxl = o.i[6];
  if (a.k1 % 32 || a.k2 % 32 || a.rows <= 0 || a.n_cob <= 0) return -3;
  const int ntr8 = (a.rows + TM - 1) / TM;
  // (opt-in, SLIDE_TAIL8=1: measured equal to the four-wave form for one chain and 2 % slower with four chains in flight --
  //  both forms are bound by the per-CU L2 -> LDS fill rate of the non-resident u / mo tiles)
  static const bool tail8_on = [] { const char *e = getenv("SLIDE_TAIL8"); return e && e[0] == '1'; }();
  const int ntc = (a.n_cob + 1) / 2, ntr = (a.rows + TM - 1) / TM;
  const int grid = ((ntr + 7) / 8) * 8 * ntc;
#ifndef SLIDE_EXPERIMENTS
  (void)ntr8;
  if (tail8_on || ((int)o.f[1] & 6)) return SLIDE_ST_EXPERIMENT;  // eight-wave / wide / three-workgroup tails
#else
  if (tail8_on && a.n_cob >= 8 && (npxl == 7 || npxl == 8) && a.k1 % 64 == 0 && a.k2 % 64 == 0) {  // eight-wave 256 x 128 tiles (see attn_tail8_kernel)
    const size_t shm8 = (size_t)3 * 2 * (TM + 128) * 64 + 4 * 128 * 4 + 64;
    const int grid8 = ((ntr8 + 7) / 8) * 8 * ((a.n_cob + 3) / 4);
    static bool attr8_done[SLIDE_MAX_DEVICES] = {};
    bool &attr8 = attr8_done[current_device_slot()];
    if (!attr8) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&attn_tail8_kernel<7>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&attn_tail8_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr8 = true;
    }
    if (npxl == 8) hipLaunchKernelGGL(attn_tail8_kernel<8>, dim3(grid8), dim3(512), shm8, s, a);
    else hipLaunchKernelGGL(attn_tail8_kernel<7>, dim3(grid8), dim3(512), shm8, s, a);
    return (int)hipGetLastError();
  }
  if (((int)o.f[1] & 4) && npxl == 8 && a.n_cob % 4 == 0) {  // plan knob SLIDE_TAIL_WIDE: 256 x 128 tiles (attn_tail_wide_kernel)
    const int ntc4 = a.n_cob / 4;
    const int grid4 = ((ntr8 + 7) / 8) * 8 * ntc4;
    const size_t shm4 = (size_t)3 * (TM + 128) * 64 + 4 * 4 * 32 * 4 + 64;
    static bool attrw_done[SLIDE_MAX_DEVICES] = {};
    bool &attrw = attrw_done[current_device_slot()];
    if (!attrw) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&attn_tail_wide_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      attrw = true;
    }
    hipLaunchKernelGGL(attn_tail_wide_kernel<8>, dim3(grid4), dim3(256), shm4, s, a);
    return (int)hipGetLastError();
  }
  const bool tail_occ3 = ((int)o.f[1] & 2) != 0;  // (plan knob SLIDE_TAIL_OCC3: two-stage ring, three workgroups per CU)
  if (tail_occ3 && (npxl == 7 || npxl == 8)) {
    const size_t shm3 = (size_t)2 * (TM + 64) * 64 + 4 * 2 * 32 * 4 + 64;
    if (npxl == 8) hipLaunchKernelGGL(attn_tail_occ3_kernel<8>, dim3(grid), dim3(256), shm3, s, a);
    else hipLaunchKernelGGL(attn_tail_occ3_kernel<7>, dim3(grid), dim3(256), shm3, s, a);
    return (int)hipGetLastError();
  }
#endif
  static const int tail_rx = [] { const char *e = getenv("SLIDE_TAIL_RX"); return e ? atoi(e) : 1; }();
  if (tail_rx && (npxl == 7 || npxl == 8) && (a.k2 / 32) % RXD == 0) {  // X fragments through registers (attn_tail_rx_kernel)
    const size_t shmr = (size_t)(RXD + 1) * 64 * 64 + 4 * 2 * 32 * 4 + 4 * 2 * 32 * 2 * 4;
    if (a.x_fm) {
      if (a.x1_ld != 32 || a.x2_ld != 32 || a.rows % 32 || !a.w_cm) return -3;
      if (npxl == 8) hipLaunchKernelGGL((attn_tail_rx_kernel<8, true>), dim3(grid), dim3(256), shmr, s, a);
      else hipLaunchKernelGGL((attn_tail_rx_kernel<7, true>), dim3(grid), dim3(256), shmr, s, a);
      return (int)hipGetLastError();
    }
    if (npxl == 8) hipLaunchKernelGGL((attn_tail_rx_kernel<8, false>), dim3(grid), dim3(256), shmr, s, a);
    else hipLaunchKernelGGL((attn_tail_rx_kernel<7, false>), dim3(grid), dim3(256), shmr, s, a);
    return (int)hipGetLastError();
  }
  if (a.x_fm) return -3;  // fragment-major u / mo: only the register-X kernel reads that layout
  const size_t shm = (size_t)SLIDE_ATTN_NST * (TM + 64) * 64 + 4 * 2 * 32 * 4 + 64;
  static bool attr_done[SLIDE_MAX_DEVICES] = {};
  bool &attr_set = attr_done[current_device_slot()];
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&attn_tail_kernel<7>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&attn_tail_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    attr_set = true;
  }
  if (npxl == 8) hipLaunchKernelGGL(attn_tail_kernel<8>, dim3(grid), dim3(256), shm, s, a);
  else if (npxl == 7) hipLaunchKernelGGL(attn_tail_kernel<7>, dim3(grid), dim3(256), shm, s, a);
  else return -4;
  return (int)hipGetLastError();
}

int run_op(const SlideOp &o, hipStream_t s) {
  switch (o.kind) {
    case SLIDE_OP_GEMM:
      return run_gemm(o, s);
    case SLIDE_OP_PREP_POINTS:
      if (o.i[3] == SLIDE_PREC_F16)
        hipLaunchKernelGGL(prep_points_kernel<_Float16>, dim3(o.i[0]), dim3(256), 0, s, o.i[1], o.i[2],
                           (const float *)o.p[0], (float *)o.p[1], (_Float16 *)o.p[2], (int *)o.p[3], (float *)o.p[4],
                           (_Float16 *)o.p[5], (const SlidePrepCopy *)o.p[6], o.i[4], (float *)o.p[7]);
      else
        hipLaunchKernelGGL(prep_points_kernel<float>, dim3(o.i[0]), dim3(256), 0, s, o.i[1], o.i[2],
                           (const float *)o.p[0], (float *)o.p[1], (float *)o.p[2], (int *)o.p[3], (float *)o.p[4],
                           (float *)o.p[5], (const SlidePrepCopy *)o.p[6], o.i[4], (float *)o.p[7]);
      break;
    case SLIDE_OP_ASSEMBLE_SA:
    case SLIDE_OP_ASSEMBLE_FP: {
      const bool fp = o.kind == SLIDE_OP_ASSEMBLE_FP;
      // bulk (whole 16-byte pieces of the gathered rows) and tail (coordinate pieces) blocks, see the kernel.
      // i[6] = c_begin (multiple of 32): only columns >= c_begin are produced, into rows of i[7] elements
      const int esz = o.i[5] == SLIDE_PREC_F16 ? 2 : 4;
      const int c_begin = o.i[6], ld_out = o.i[6] ? o.i[7] : o.i[3];
      const int nch = o.i[3] / 8, nbulk = (esz * o.i[2] % 16 == 0) ? o.i[1] / 8 : 0, ntail = nch - nbulk;
      const int nb_eff = nbulk - c_begin / 8 > 0 ? nbulk - c_begin / 8 : 0;
      int nch_log2 = 0;
      while ((1 << nch_log2) < nb_eff) ++nch_log2;
      const int bulk_blocks = nb_eff ? (((16 * o.i[4]) << nch_log2) + 255) / 256 : 0;
      const int tail_blocks = (16 * o.i[4] * ntail + 255) / 256;
      const dim3 g(o.i[0], bulk_blocks + tail_blocks), blk(256);
      const float *kd2 = fp ? (const float *)o.p[3] : nullptr;
      void *dst = fp ? o.p[4] : o.p[3];
#define ASM(TT, FPB)                                                                                                  \
  hipLaunchKernelGGL((assemble_kernel<TT, FPB>), g, blk, 0, s, o.i[1], o.i[2], o.i[3], o.i[4], nch_log2, bulk_blocks, c_begin, ld_out, (const float *)o.p[0], \
                     (const TT *)o.p[1], (const int *)o.p[2], kd2, (TT *)dst)
      if (o.i[5] == SLIDE_PREC_F16) { if (fp) ASM(_Float16, true); else ASM(_Float16, false); }
      else { if (fp) ASM(float, true); else ASM(float, false); }
#undef ASM
      break;
    }
    case SLIDE_OP_FINALIZE_GN:
      hipLaunchKernelGGL(finalize_gn_kernel, dim3((o.i[0] * o.i[1] + 255) / 256), dim3(256), 0, s, o.i[0], o.i[1],
                         o.i[2], o.f[0], (const float *)o.p[0], (const float *)o.p[1], (const int *)o.p[2],
                         (const int *)o.p[3], (const int *)o.p[4], (const float *)o.p[5], (const float *)o.p[6],
                         (float *)o.p[7], (float *)o.p[8]);
      break;
    case SLIDE_OP_ATTN_COMBINE: {
      const int n = o.i[0] * o.i[1];
      const dim3 g((n + 255) / 256), blk(256);
#define ATTN(KK, TT)                                                                                              \
  hipLaunchKernelGGL((attn_combine_kernel<KK, TT>), g, blk, 0, s, o.i[0], o.i[1], o.i[2], o.i[3], o.i[4],         \
                     (const TT *)o.p[0], (const TT *)o.p[1], (TT *)o.p[2])
      if (o.i[5] == 16 && o.i[6] == SLIDE_PREC_F16) ATTN(16, _Float16);
      else if (o.i[5] == 16) ATTN(16, float);
      else if (o.i[5] == 8 && o.i[6] == SLIDE_PREC_F16) ATTN(8, _Float16);
      else if (o.i[5] == 8) ATTN(8, float);
      else return -5;
#undef ATTN
      break;
    }
    case SLIDE_OP_COPY_COLS: {
      if (o.i[0] <= 0 || o.i[1] <= 0) break;  // nothing to copy (a zero-sized grid is a launch error)
      const dim3 g((o.i[0] * o.i[1] + 255) / 256), blk(256);
#define CPY(TS, TD)                                                                                               \
  hipLaunchKernelGGL((copy_cols_kernel<TS, TD>), g, blk, 0, s, o.i[0], o.i[1], o.i[2], o.i[3], (const TS *)o.p[0], \
                     (TD *)o.p[1])
      if (o.i[4] && o.i[5]) CPY(_Float16, _Float16);
      else if (o.i[4]) CPY(_Float16, float);
      else if (o.i[5]) CPY(float, _Float16);
      else CPY(float, float);
#undef CPY
      break;
    }
    case SLIDE_OP_TEMB:
      hipLaunchKernelGGL(temb_kernel, dim3(o.i[0]), dim3(256), (size_t)o.i[1] * 9 * sizeof(float), s, o.i[1], o.i[2],
                         (const float *)o.p[0], (const int *)o.p[1], (const float *)o.p[9], (const float *)o.p[2],
                         (const float *)o.p[3], (const float *)o.p[4], (const float *)o.p[5], (const float *)o.p[6],
                         (const float *)o.p[7], (float *)o.p[8]);
      break;
    case SLIDE_OP_COND:
      hipLaunchKernelGGL(cond_kernel, dim3(o.i[0]), dim3(256), (size_t)o.i[1] * sizeof(float), s, o.i[1], o.i[2],
                         (const int64_t *)o.p[0], (const float *)o.p[1], (const float *)o.p[2], (const float *)o.p[3],
                         (float *)o.p[4]);
      break;
    case SLIDE_OP_UPDATE_POS:
      hipLaunchKernelGGL(update_pos_kernel, dim3((o.i[0] + 255) / 256), dim3(256), 0, s, o.i[0], o.i[1], (uint32_t)o.i[2],
                         (uint32_t)o.i[3], (float *)o.p[0], (const float *)o.p[1], (const float *)o.p[2],
                         (int *)o.p[3], (const float *)o.p[4], (const float *)o.p[5], (const float *)o.p[6]);
      break;
    case SLIDE_OP_UPDATE_FEAT:
      hipLaunchKernelGGL(update_feat_kernel, dim3((o.i[0] * o.i[1] + 1023) / 1024), dim3(256), 0, s, o.i[0], o.i[1],
                         o.i[2], o.i[5], o.f[0], (uint32_t)o.i[3], (uint32_t)o.i[4], (float *)o.p[0], (const float *)o.p[1],
                         (const float *)o.p[2], (int *)o.p[3], (const float *)o.p[4], (const float *)o.p[5],
                         (const float *)o.p[6], (const float *)o.p[7], (const float *)o.p[8], (const float *)o.p[9],
                         (const float *)o.p[10], (const float *)o.p[11], o.p[12], o.i[6], o.i[7],
                         (const SlidePrepCopy *)o.p[13], o.i[8]);
      break;
#ifndef SLIDE_EXPERIMENTS
    case SLIDE_OP_HEAD_UPDATE:
    case SLIDE_OP_GEMM_CHAIN:
      return SLIDE_ST_EXPERIMENT;
#else
    case SLIDE_OP_GEMM_CHAIN:
      return slide_launch_gemm_chain(o, s);
    case SLIDE_OP_HEAD_UPDATE: {
      const SlideHeadArgs *h = (const SlideHeadArgs *)o.p[0];  // HOST pointer, kept alive by the plan
      if (!h || h->rows <= 0 || h->rows % 16 || h->k0 % 32 || h->k0 <= 0 || h->k0 > 160 || h->x_ld < h->k0 || h->x_ld % 8 ||
          (h->n1c != 1 && h->n1c != 2) || !h->X || !h->W0 || !h->W1 || !h->v0 || !h->b1 || !h->x || !h->t_dev ||
          (h->kind != 0 && h->kind != 1) || h->C > 32 * h->n1c)
        return -3;
      const int grid = (h->rows + 63) / 64;
      if (h->k0 <= 96) hipLaunchKernelGGL(head_update_kernel<96>, dim3(grid), dim3(256), 0, s, *h);
      else hipLaunchKernelGGL(head_update_kernel<160>, dim3(grid), dim3(256), 0, s, *h);
      break;
    }
#endif
    case SLIDE_OP_POINT_CHAIN:
      return slide_launch_point_chain(o, s);
    case SLIDE_OP_GEMM_ATTEND:
      return run_gemm_attend(o, s);
    case SLIDE_OP_ATTN_TAIL:
      return ((int)o.f[1] & 8) ? slide_launch_attn_tail_split(o, s) : run_attn_tail(o, s);
    case SLIDE_OP_GEMM_GX:
      return (int)o.f[0] == 3 ? slide_launch_gemm_gxs(o, s) : slide_launch_gemm_gx(o, s);
    case SLIDE_OP_GEMM_GX_DUAL:
      return slide_launch_gemm_gx_dual(o, s);
    case SLIDE_OP_PAIR_NORM:
      return slide_launch_pair_norm(o, s);
    case SLIDE_OP_PP_STAGE:
      return slide_launch_pp_stage(o, s);
    case SLIDE_OP_PAIR_FIRST:
      return run_pair_first(o, s);
    case SLIDE_OP_SA_CHAIN:
      return slide_launch_sa_chain(o, s);
    case SLIDE_OP_SA_CHAIN_P:
      return slide_launch_sa_chain_p(o, s);
    case SLIDE_OP_BLOCK_BODY:
#ifdef SLIDE_EXPERIMENTS
      return slide_launch_block_body(o, s);
#else
      return SLIDE_ST_EXPERIMENT;  // (opt-in since round 5: block_body.hip is part of the experiments build)
#endif
    case SLIDE_OP_TRANSPOSE:
      if (o.i[7])  // fp16 destination (module-level throughput mode)
        hipLaunchKernelGGL(transpose_kernel<_Float16>, dim3((o.i[2] + 31) / 32, (o.i[1] + 31) / 32, o.i[0]), dim3(256), 0, s,
                           o.i[1], o.i[2], o.i[3], o.i[4], (long long)o.i[5], (long long)o.i[6], (const float *)o.p[0],
                           (_Float16 *)o.p[1]);
      else
        hipLaunchKernelGGL(transpose_kernel<float>, dim3((o.i[2] + 31) / 32, (o.i[1] + 31) / 32, o.i[0]), dim3(256), 0, s,
                           o.i[1], o.i[2], o.i[3], o.i[4], (long long)o.i[5], (long long)o.i[6], (const float *)o.p[0],
                           (float *)o.p[1]);
      break;
    case SLIDE_OP_GROUPNORM_NCHW:  // i: B, C, HW, G, n_norm, relu   p: x, gamma, beta, y
      hipLaunchKernelGGL(group_norm_nchw_kernel, dim3(o.i[3] + (o.i[4] < o.i[1] ? 1 : 0), o.i[0]), dim3(256), 0, s, o.i[1],
                         o.i[2], o.i[3], o.i[4], o.i[5], (const float *)o.p[0], (const float *)o.p[1],
                         (const float *)o.p[2], (float *)o.p[3]);
      break;
    case SLIDE_OP_ADVANCE_T:
      hipLaunchKernelGGL(advance_t_kernel, dim3(1), dim3(64), 0, s, (int *)o.p[0]);
      break;
    default:
      if ((o.kind >= SLIDE_OP_ROWS_FROM_NCX && o.kind <= SLIDE_OP_ROWS_GN_JOINT) || o.kind == SLIDE_OP_ROWS_PAIR_EXPAND)
        return slide_launch_rows_op(o, s);
      return -1;
  }
  return (int)hipGetLastError();
}

}  // namespace

extern "C" {

int slide_run_ops(const SlideOp *ops, int n, slide_stream_t stream) { return slide_run_ops2(ops, n, stream, stream); }

// Two-lane replay: op.i[10] selects the lane (stream); SLIDE_OP_SYNC(from, to) makes lane `to` wait for everything
// issued so far on lane `from` (event record + stream wait -> a plain edge when captured into a hipGraph).
int slide_run_ops2(const SlideOp *ops, int n, slide_stream_t stream0, slide_stream_t stream1) {
  hipStream_t ss[2] = {(hipStream_t)stream0, (hipStream_t)stream1};
  // fork / join events of the two-lane plans: one small pool per host thread (events belong to the device that was
  // current when they were created; a thread drives one device)
  static thread_local hipEvent_t pool[256];
  static thread_local int pool_n = 0, pool_next = 0;
  for (int i = 0; i < n; ++i) {
    const SlideOp &o = ops[i];
    if (o.kind == SLIDE_OP_SYNC) {
      const int from = o.i[0] & 1, to = o.i[1] & 1;
      if (ss[from] == ss[to]) continue;
      if (pool_n < 256) {
        if (hipEventCreateWithFlags(&pool[pool_n], hipEventDisableTiming) != hipSuccess) return -9;
        ++pool_n;
      }
      hipEvent_t ev = pool[pool_next % pool_n];
      pool_next = (pool_next + 1) % 256;
      hipError_t e = hipEventRecord(ev, ss[from]);
      if (e == hipSuccess) e = hipStreamWaitEvent(ss[to], ev, 0);
      if (e != hipSuccess) return (int)e;
      continue;
    }
    const int st = run_op(o, ss[o.i[10] & 1]);
    if (st != 0) return st > 0 ? st : st * 1000 - i;
  }
  return 0;
}

// `reps` eager replays of one plan in a single call: a host thread per chain can keep its stream fed without returning
// to the interpreter between steps (plans advance their own device-side timestep).  Thread-safe for plans without
// SLIDE_OP_SYNC on distinct streams once every kernel has been launched at least once (first-use attribute calls).
int slide_run_ops_repeat(const SlideOp *ops, int n, slide_stream_t stream0, slide_stream_t stream1, int reps) {
  for (int r = 0; r < reps; ++r) {
    const int st = slide_run_ops2(ops, n, stream0, stream1);
    if (st != 0) return st;
  }
  return 0;
}

// `reps` steps of several independent chains from ONE host thread, round-robin: step r of every chain is issued before
// step r + 1 of any (chain c replays ops[c][0..n[c]) on streams[c]; single-lane plans).
int slide_run_chains(const SlideOp *const *ops, const int *n, const slide_stream_t *streams, int n_chains, int reps) {
  for (int r = 0; r < reps; ++r)
    for (int c = 0; c < n_chains; ++c) {
      const int st = slide_run_ops2(ops[c], n[c], streams[c], streams[c]);
      if (st != 0) return st;
    }
  return 0;
}

// The same with chain c stepping only on every every[c]-th round (a chain over a multiple of the common batch advances once per
// `every` rounds of the others: same shapes per unit time, fewer dependent launches on the critical path).
int slide_run_chains_every(const SlideOp *const *ops, const int *n, const slide_stream_t *streams, const int *every, int n_chains,
                           int reps) {
  for (int r = 0; r < reps; ++r)
    for (int c = 0; c < n_chains; ++c) {
      if (every[c] > 1 && r % every[c] != 0) continue;
      const int st = slide_run_ops2(ops[c], n[c], streams[c], streams[c]);
      if (st != 0) return st;
    }
  return 0;
}

// Eager replay with a HIP event between consecutive launches (recorded on the launch stream): ms_out[i] = device
// time of ops[i].  Used by bench.py for the per-kernel roofline figure; not used on the timed path.
int slide_run_ops_timed(const SlideOp *ops, int n, slide_stream_t stream, float *ms_out) {
  hipStream_t s = (hipStream_t)stream;
  if (n <= 0 || n > 4096) return -6;
  hipEvent_t *ev = new hipEvent_t[n + 1];
  int st = 0;
  for (int i = 0; i <= n; ++i) st |= (int)hipEventCreate(&ev[i]);
  if (st == 0) st = (int)hipEventRecord(ev[0], s);
  for (int i = 0; i < n && st == 0; ++i) {
    if (ops[i].kind != SLIDE_OP_SYNC) st = run_op(ops[i], s);  // single-stream replay: syncs are no-ops
    if (st == 0) st = (int)hipEventRecord(ev[i + 1], s);
  }
  if (st == 0) st = (int)hipEventSynchronize(ev[n]);
  for (int i = 0; i < n && st == 0; ++i) st = (int)hipEventElapsedTime(&ms_out[i], ev[i], ev[i + 1]);
  for (int i = 0; i <= n; ++i) (void)hipEventDestroy(ev[i]);
  delete[] ev;
  return st;
}

int slide_graph_begin(slide_stream_t stream) {
  return (int)hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal);
}
int slide_graph_end(slide_stream_t stream, void **graph_exec_out) {
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture((hipStream_t)stream, &g);
  if (e != hipSuccess) return (int)e;
  hipGraphExec_t ex = nullptr;
  e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) return (int)e;
  *graph_exec_out = (void *)ex;
  return 0;
}
int slide_graph_launch(void *graph_exec, slide_stream_t stream) {
  return (int)hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream);
}
int slide_graph_destroy(void *graph_exec) { return (int)hipGraphExecDestroy((hipGraphExec_t)graph_exec); }

int slide_stream_create_cu_mask(const uint32_t *mask, int n_words, slide_stream_t *stream_out) {
  if (!mask || n_words <= 0 || !stream_out) return -3;
  hipStream_t st = nullptr;
  const hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)n_words, mask);
  *stream_out = (slide_stream_t)st;
  return (int)e;
}
int slide_stream_destroy(slide_stream_t stream) { return (int)hipStreamDestroy((hipStream_t)stream); }

int slide_event_create(void **ev) {
  hipEvent_t e;
  const hipError_t st = hipEventCreate(&e);
  *ev = (void *)e;
  return (int)st;
}
int slide_event_record(void *ev, slide_stream_t stream) { return (int)hipEventRecord((hipEvent_t)ev, (hipStream_t)stream); }
int slide_event_elapsed_ms(void *start, void *stop, float *ms) {
  hipError_t st = hipEventSynchronize((hipEvent_t)stop);
  if (st != hipSuccess) return (int)st;
  return (int)hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
}
int slide_event_destroy(void *ev) { return (int)hipEventDestroy((hipEvent_t)ev); }

int slide_sizeof_epi(void) { return (int)sizeof(SlideEpi); }
int slide_sizeof_op(void) { return (int)sizeof(SlideOp); }

}  // extern "C"
