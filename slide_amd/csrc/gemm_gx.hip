// gemm_gx.hip -- the PAIR DECOMPOSITION of the SA / FP blocks' first layers (round 3).
//
// Every SA / FP block of PointNet2CloudCondition starts with 1x1 convolutions over the grouped input
//   row (p, j) = [features of neighbour q | coordinate channels of (q, p)]        (pointnet2_utils.py:383-430, :497-524)
// (first_mlp, res_connect, grouped_feat_conv: pointnet2_modules.py:119-176, attention.py:70-96).  A 1x1 convolution is
// linear, and every coordinate channel is a linear function of xyz[q] and xyz[p], so its output separates:
//   y(p, j) = a[q] + b[p] (+ d2(p, j) vd + w(p, j) vw for group_knn's two per-slot scalars)
// with a[q] = Wf feat[q] + (W_rel + W_abs) xyz[q] + bias and b[p] = (W_ctr - W_rel) xyz[p] -- a 16-row GEMM per sample
// instead of a 256- / 128-row one (1/16 or 1/8 of the MACs: 31 % of the feature denoiser's FLOPs), and, more important,
// the K-expanded first-layer outputs (1056 channels x 256 rows per sample at SA1: the largest tensors of a step) are never
// written to or read from HBM:
//   * pair_norm_kernel turns the per-point products into the tables (a, b): GroupNorm statistics over the sample's pairs
//     (exact, fp32) are folded INTO the tables (a g + shift, b g), or emitted as sums for the attention's joint GroupNorm;
//   * gemm_gx_kernel is the consumer GEMM of such a layer: a workgroup keeps its sample's tables in LDS and GENERATES the
//     MFMA B fragments (x = max(a[q] + b[p], 0) + t-embedding, or max(.)*scale + shift) between LDS and MFMA -- three packed
//     fp16 VALU ops per fragment register -- while only the weights stream through an LDS-DMA ring;
//   * the res_connect output re-enters as a PAIR residual in the common epilogue (gemm_common.h).
// In the SA blocks every point is a neighbour of every point (K = N = 16) and everything downstream (GroupNorm statistics,
// softmax-weighted sum over the neighbours) is invariant to the neighbour ORDER, so rows run in natural order (q = j): no
// index table, and the a-fragment is shared by all row blocks of a wave.
#include <cstdlib>

#include "gemm_common.h"
#include "gemm_small.h"
#include "pair_norm.h"

namespace {

constexpr int SLIDE_MAX_DEVICES = 64;

// ------------------------------------------------------------------------------------------------ generated-X GEMM
// Tile: 256 rows (one 16x16 sample, or two 16x8 samples) x 128 channels; four waves, wave w owns rows 64 w .. 64 w + 63 and
// all 128 channels (acc[4][2] = 128 registers).  W: chunk-major [k / 32][n_cob * 32][32]; a ring stage is two 32-deep
// chunk images ([128 rows][64 B], source-side XOR swizzle as in the ring kernels of engine.hip) = 16 KB; NST stages.
// Tables in LDS: [16-byte piece of the row][table row][8 halves] -- the 16 rows a ds_read_b128 lane group touches for one
// piece are 256 consecutive bytes (conflict-free for any neighbour permutation), filled by LDS-DMA (per-lane source rows).
// The K loop is software-pipelined by hand over its 16-deep steps: the LDS reads of step s + 1 (weight fragments, table
// rows, vectors) are issued before the B fragments of step s are generated and its 8 MFMAs issued, so the reads' latency and
// the packed-fp16 generation of the next fragments run under the matrix pipe's 256 busy cycles; the ring's wait + barrier of
// the next stage sits one step early for the same reason.
template <int NPXL, int NST, int MODE, int CBW>
__device__ __forceinline__ void gemm_gx_body(const GemmArgs &a, const int bid) {
  using T = _Float16;
  constexpr int NPX = 1 << NPXL;
  constexpr bool FP = NPXL == 7;
  constexpr int NSAMP = TM >> NPXL;            // 1 or 2
  constexpr int NR = 16 * NSAMP;               // table rows in LDS
  constexpr int CH_B = 32 * CBW * 64;                      // a weight chunk image: [32 CBW weight rows][64 B]
  constexpr int NJ = CBW / 2;                              // weight DMA instructions per wave and chunk image (16 rows each)
  // the sample's pair tables STREAM with the weights (round 3): a stage also carries, per 32-deep chunk and table, the image
  // [4 sixteen-byte pieces][NR rows][8 halves] = 64 NR bytes -- instead of both tables staged whole before the K loop
  // (2 x 16 x k_pad x 2 B per sample: 69 KB on the FP1 layer, which pinned that launch to one workgroup per CU, and a 2 - 3 us
  // prologue on every workgroup)
  constexpr int TBL_B = 64 * NR;                           // one (chunk, table) image
  constexpr int STAGE_B = 2 * CH_B + 4 * TBL_B;
  constexpr int NTI = NSAMP;                               // table DMA instructions per wave and stage (1 KB each)
  constexpr int NVEC = (MODE ? 2 : 1) + (FP ? 2 : 0);  // fp16 vectors per sample in LDS: [add | scale, shift][vd, vw][k_pad]
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int ntc = (a.n_cob + CBW - 1) / CBW;
  const int ntr = (a.rows + TM - 1) / TM;
  const int xcd = bid & 7, q0 = bid >> 3;
  const int tc = q0 % ntc, tr = (q0 / ntc) * 8 + xcd;  // the column tiles of a row tile share one XCD's L2 (weights + tables)
  if (tr >= ntr) return;
  const int row0 = tr * TM, cob0 = tc * CBW;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, col = lane & 31;
  const int nk32 = a.k_pad >> 5, nks = (nk32 + 1) >> 1;
  unsigned char *const ring = smem_raw;
  SLIDE_STAMP(a, 0);
  uint32_t *const epi_lds = reinterpret_cast<uint32_t *>(smem_raw + (size_t)NST * STAGE_B);
  float *const vec_lds = reinterpret_cast<float *>(epi_lds + CBW * EPI_DW + (CBW * EPI_DW) % 4);
  T *const vv_l = reinterpret_cast<T *>(vec_lds + CBW * 96);
  const int nsm = a.rows >> NPXL, smp0 = row0 >> NPXL;

  // PROLOGUE ORDER (round 6): descriptor / vector DMAs and the first ring stages are ISSUED before anything waits -- the plain loads
  // below (per-sample vectors, neighbour slots) then share ONE memory round trip with them instead of preceding them (the
  // per-workgroup timeline's "tables" phase: 1.0 - 1.7 us of a 7 - 17 us workgroup; tools/ab/op_timeline.py)
  stage_epilogue_tables<CBW, 256>(a, cob0, tid, epi_lds, vec_lds);
  // ---- table stream: this wave's NTI instructions per stage; instruction id = wave + 4 n -> (chunk of the stage, table, piece half)
  const T *tsrc[NTI];
  int tdst[NTI];
#pragma unroll
  for (int n = 0; n < NTI; ++n) {
    const int id = wave + 4 * n;                       // SA: 0..3 = (c2, t); FP: 0..7 = (c2, t, h)
    const int c2 = FP ? (id >> 2) & 1 : (id >> 1) & 1, t = FP ? (id >> 1) & 1 : id & 1, h = FP ? id & 1 : 0;
    const int piece = FP ? 2 * h + (lane >> 5) : lane >> 4, r = lane & (NR - 1);
    int smp = smp0 + (r >> 4);
    smp = smp < nsm ? smp : nsm - 1;
    tsrc[n] = reinterpret_cast<const T *>(t ? a.gx_tb : a.gx_ta) + ((size_t)smp * 16 + (r & 15)) * a.gx_ld + piece * 8;
    tdst[n] = 2 * CH_B + (c2 * 2 + t) * TBL_B + (FP ? 2 * h * NR * 16 : 0);
  }
  // ---- weight ring: this lane's source piece of the wave's two DMA instructions per 32-deep chunk
  const T *wsrc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int trow = 16 * (j * 4 + wave) + (lane >> 2);
    const int piece = (lane & 3) ^ ((trow >> 2) & 3);
    int gco = cob0 * 32 + trow;
    gco = gco < a.n_cob * 32 ? gco : a.n_cob * 32 - 1;  // rows beyond the matrix: clamp (their channels are never stored)
    wsrc[j] = reinterpret_cast<const T *>(a.W) + (size_t)gco * 32 + piece * 8;
  }
  const size_t w_cs = (size_t)a.n_cob * 32 * 32;  // elements between consecutive chunks
  auto issue = [&](int st) __attribute__((always_inline)) {  // stage st -> slot st % NST (2 NJ DMA instructions per wave)
    unsigned char *dst = ring + (size_t)(st % NST) * STAGE_B;
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
      int kc = st * 2 + c2;
      kc = kc < nk32 ? kc : nk32 - 1;  // odd chunk count: the last stage's second image is a dummy (never read)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)(wsrc[j] + (size_t)kc * w_cs),
                                         (__attribute__((address_space(3))) void *)(dst + c2 * CH_B + (j * 4 + wave) * 1024),
                                         16, 0, 0);
    }
#pragma unroll
    for (int n = 0; n < NTI; ++n) {
      int kc = st * 2 + (FP ? n : (wave >> 1));  // the chunk of the stage this instruction serves (id = wave + 4 n)
      kc = kc < nk32 ? kc : nk32 - 1;
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)(tsrc[n] + kc * 32),
                                       (__attribute__((address_space(3))) void *)(dst + tdst[n]), 16, 0, 0);
    }
  };
#pragma unroll
  for (int s0 = 0; s0 < NST - 1; ++s0)
    if (s0 < nks) issue(s0);
  // ---- per-sample vectors (plain loads, converted to fp16)
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
      T *dst = vv_l + (size_t)sl * NVEC * a.k_pad + k;
      store4<T>(dst, v0);
      if (MODE) store4<T>(dst + a.k_pad, v1);
      if (FP) {
        store4<T>(dst + (MODE ? 2 : 1) * a.k_pad, vd);
        store4<T>(dst + (MODE ? 3 : 2) * a.k_pad, vw);
      }
    }
  }
  // this lane's two row blocks: table rows of the neighbour (a) and of the centre point (b), per-slot scalars
  int aoff[2], boff[2];
  f16x2 d2s[2], ws[2];
  const int sl_w = FP ? (wave >> 1) : 0;
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    int row = row0 + wave * 64 + rb * 32 + col;
    row = row < a.rows ? row : a.rows - 1;
    const int smp = row >> NPXL, pxl = row & (NPX - 1);
    int p, q;
    float d2 = 0.f, w = 0.f;
    if (!FP) { p = pxl >> 4; q = pxl & 15; }
    else {
      p = pxl >> 3;
      const int slot = (smp * 16 + p) * 16 + (pxl & 7);
      q = a.gidx[slot];
      d2 = a.gx_d2[slot]; w = a.gx_w[slot];
    }
    aoff[rb] = (half * NR + sl_w * 16 + q) * 16;
    boff[rb] = (half * NR + sl_w * 16 + p) * 16;
    d2 = fminf(d2, 65504.f);  // (fp16 range: beyond it the conversion gives inf)
    d2s[rb] = f16x2{(T)d2, (T)d2};
    ws[rb] = f16x2{(T)w, (T)w};
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every plain load above has landed (and the primed stages with them)

  SLIDE_STAMP(a, 7);

  const unsigned char *const vbase = reinterpret_cast<const unsigned char *>(vv_l + (size_t)sl_w * NVEC * a.k_pad) + half * 16;
  const int vstr = a.k_pad * 2;  // bytes between the vectors of a sample

  f32x16 acc[CBW][2];
#pragma unroll
  for (int i = 0; i < CBW; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  int wrow[CBW], wkey[CBW];
#pragma unroll
  for (int cb = 0; cb < CBW; ++cb) {
    const int trow = cb * 32 + col;
    wrow[cb] = trow * 64; wkey[cb] = (trow >> 2) & 3;
  }

  // operands of one 16-deep step: weight fragments + the table rows its B fragments are generated from (the per-channel
  // vectors are uniform reads, fetched where they are used: keeping them out of the double buffer is what fits 256 registers)
  struct Step { f16x8 af[CBW], av[FP ? 2 : 1], bv[2]; int kb; };
  auto load_step = [&](Step &o, const unsigned char *sb, int c2, int st2, int kc) __attribute__((always_inline)) {
    const int piece = st2 * 2 + half;
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb)
      o.af[cb] = *reinterpret_cast<const f16x8 *>(sb + c2 * CH_B + wrow[cb] + ((piece ^ wkey[cb]) << 4));
    const unsigned char *ta_s = sb + 2 * CH_B + (c2 * 2) * TBL_B + st2 * 2 * NR * 16;  // this step's pieces of the a image
    const unsigned char *tb_s = ta_s + TBL_B;
    o.kb = (kc * 32 + st2 * 16) * 2;              // byte offset of this step inside a vector
    o.av[0] = *reinterpret_cast<const f16x8 *>(ta_s + aoff[0]);
    if (FP) o.av[FP ? 1 : 0] = *reinterpret_cast<const f16x8 *>(ta_s + aoff[1]);  // (natural order: q is the same in both blocks)
    o.bv[0] = *reinterpret_cast<const f16x8 *>(tb_s + boff[0]);
    o.bv[1] = *reinterpret_cast<const f16x8 *>(tb_s + boff[1]);
  };
  auto compute_step = [&](const Step &o) __attribute__((always_inline)) {
    const f16x2 zero2 = {0, 0};
    const f16x8 v0 = *reinterpret_cast<const f16x8 *>(vbase + o.kb);
    f16x8 v1, vd, vw;
    if (MODE) v1 = *reinterpret_cast<const f16x8 *>(vbase + vstr + o.kb);
    if (FP) {
      vd = *reinterpret_cast<const f16x8 *>(vbase + (MODE ? 2 : 1) * vstr + o.kb);
      vw = *reinterpret_cast<const f16x8 *>(vbase + (MODE ? 3 : 2) * vstr + o.kb);
    }
    f16x8 bf[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const f16x8 av = o.av[FP ? rb : 0], bv = o.bv[rb];
#pragma unroll
      for (int i = 0; i < 4; ++i) {  // packed fp16 math on register pieces; the per-slot scalars are ONE register each
        f16x2 y = f16x2{av[2 * i], av[2 * i + 1]} + f16x2{bv[2 * i], bv[2 * i + 1]};
        if (FP) {
          y = __builtin_elementwise_fma(d2s[rb], f16x2{vd[2 * i], vd[2 * i + 1]}, y);
          y = __builtin_elementwise_fma(ws[rb], f16x2{vw[2 * i], vw[2 * i + 1]}, y);
        }
        y = __builtin_elementwise_max(y, zero2);
        if (MODE) y = __builtin_elementwise_fma(y, f16x2{v0[2 * i], v0[2 * i + 1]}, f16x2{v1[2 * i], v1[2 * i + 1]});
        else y = y + f16x2{v0[2 * i], v0[2 * i + 1]};
        bf[rb][2 * i] = y[0]; bf[rb][2 * i + 1] = y[1];
      }
    }
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
        acc[cb][rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.af[cb], bf[rb], acc[cb][rb], 0, 0, 0);
  };
  // stage st must have landed before anyone reads it; the next one (2 NJ instructions per wave) may stay in flight
  auto stage_ready = [&](int st) __attribute__((always_inline)) {
    if (st + 1 < nks && NST > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NJ + NTI) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // (first pass: also publishes the tables and vectors staged above)
    if (st + NST - 1 < nks) issue(st + NST - 1);  // overwrites the stage consumed at st - 1
  };

  SLIDE_STAMP(a, 1);
  Step cur, nxt;
  stage_ready(0);
  load_step(cur, ring, 0, 0, 0);
  for (int st = 0; st < nks; ++st) {
    const unsigned char *sb = ring + (size_t)(st % NST) * STAGE_B;
    const bool two = st * 2 + 1 < nk32;  // the stage's second 32-deep chunk exists
    // step 0 of chunk 0
    load_step(nxt, sb, 0, 1, st * 2);
    compute_step(cur);
    // step 1 of chunk 0
    if (two) {
      load_step(cur, sb, 1, 0, st * 2 + 1);
      compute_step(nxt);
      load_step(nxt, sb, 1, 1, st * 2 + 1);
      compute_step(cur);
    }
    // last step of the stage: the next stage's barrier and first reads go ahead of its MFMAs
    if (st + 1 < nks) {
      stage_ready(st + 1);
      load_step(cur, ring + (size_t)((st + 1) % NST) * STAGE_B, 0, 0, st * 2 + 2);
    }
    compute_step(nxt);
  }
  __syncthreads();  // every wave is done with the ring before `red` reuses it
  SLIDE_STAMP(a, 2);
  gemm_epilogue<SLIDE_PREC_F16, NPXL, CBW, 2, MODE == 0, false, false, true>(a, acc, row0, cob0, wave, half, col, epi_lds, vec_lds,  // (mode 0 = the Mlp layers: PAIR residual)
                                           reinterpret_cast<float *>(smem_raw));
  SLIDE_STAMP(a, 5);
#ifdef SLIDE_TIMELINE
  if (a.dbg) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SLIDE_STAMP(a, 6);
  }
#endif
}

template <int NPXL, int NST, int MODE>
__global__ __launch_bounds__(256, 2) void gemm_gx_kernel(GemmArgs a) {
  gemm_gx_body<NPXL, NST, MODE, 4>(a, blockIdx.x);
}

// 256 x 64 tiles inside the 168-register budget (64 accumulator registers, 8 KB ring stages): THREE workgroups per CU and
// twice the workgroups per launch -- the K loops of these layers are short (K = 64 .. 544) and each workgroup spends most of
// its life in the table prologue and the epilogue, which a third resident workgroup overlaps
template <int NPXL, int NST, int MODE>
__global__ __launch_bounds__(256, 3) void gemm_gx_n64_kernel(GemmArgs a) {
  gemm_gx_body<NPXL, NST, MODE, 2>(a, blockIdx.x);
}

// the same 64-channel tile at two workgroups per CU (256 registers: also mode 0 with its PAIR residual): for launches whose
// 128-channel grid does not cover the chip -- the FP blocks at 88 samples: 96 workgroups -- half-size tiles double the
// workgroups in flight and halve each one's K loop and epilogue
template <int NPXL, int NST, int MODE>
__global__ __launch_bounds__(256, 2) void gemm_gx_n64w_kernel(GemmArgs a) {
  gemm_gx_body<NPXL, NST, MODE, 2>(a, blockIdx.x);
}

// TWO independent generated-X GEMMs of one block in ONE launch (SLIDE_OP_GEMM_GX_DUAL): the keys -> u layer (mode 1) and the
// first Mlp layer (mode 0) of an FP block read the same pair tables and nothing of each other; as one grid they cost one
// launch gap instead of two and their workgroups fill the chip together (64-channel tiles, two workgroups per CU)
// FP blocks (NPXL = 7; round 6): THREE workgroups per CU -- a two-stage ring (42 KB with the per-sample vectors of K = 544 instead of 58)
// inside the 168-register budget.  The K loops of these layers are bound by the LDS reads and packed-fp16 generation of their fragments
// (ten ds_read_b128 and ~40 VALU per four MFMAs), their workgroups by prologue and epilogue latency: a third resident workgroup -- of
// this launch or of another chain's -- fills what two leave idle: 415.2 -> 419.7 shapes/s in bench.py's arrangement (four alternating
// pairs in one call, tools/ab/r06_libab.sh dual3).  The mode-0 body (PAIR residual) needs 178 registers when left alone: at 168 the
// compiler parks 15 values in scratch -- addressing terms of the prologue / epilogue and ONE 8-byte reload per ring stage (none in the
// MFMA steps) -- which is the one deliberate exception to the product build's no-spill rule (slide_amd/build.py SPILL_OPT_IN).
template <int NPXL>
constexpr int GX_DUAL_NST = NPXL == 7 ? 2 : 3;
template <int NPXL>
__global__ __launch_bounds__(256, NPXL == 7 ? 3 : 2) void gemm_gx_dual_kernel(GemmArgs a1, GemmArgs a0, int grid1) {
  if ((int)blockIdx.x < grid1) gemm_gx_body<NPXL, GX_DUAL_NST<NPXL>, 1, 2>(a1, blockIdx.x);
  else gemm_gx_body<NPXL, GX_DUAL_NST<NPXL>, 0, 2>(a0, blockIdx.x - grid1);
}

// ------------------------------------------------------------------------------------------------ pair-table normalisation
// One workgroup per (sample, 256-channel chunk), one channel per thread: the thread builds its channel's a[0..15] and
// b[0..15] in registers / LDS, walks the sample's pairs for the statistics of its 32-channel block's mode, and writes the
// fp16 tables the generated-X GEMMs and the PAIR residual read.  Cost: 16 K pair evaluations per channel and sample.
template <bool FP>
__global__ __launch_bounds__(256) void pair_norm_kernel(int ld, const float *__restrict__ y, const float *__restrict__ xyz,
                                                        const float *__restrict__ wa, const float *__restrict__ wb,
                                                        const SlideEpi *__restrict__ epi, _Float16 *__restrict__ ta,
                                                        _Float16 *__restrict__ tb, const int *__restrict__ nbr,
                                                        const float *__restrict__ d2t, const float *__restrict__ wt,
                                                        const float *__restrict__ vv_in, float *__restrict__ vv_out) {
  __shared__ float sx[48];
  __shared__ int sq[16 * 8];
  __shared__ float sd[16 * 8], sw[16 * 8];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int c = blockIdx.x * 256 + tid;
  if (tid < 48) sx[tid] = xyz[(size_t)b * 48 + tid];
  if (FP && tid < 128) {
    const int slot = (b * 16 + (tid >> 3)) * 16 + (tid & 7);
    sq[tid] = nbr[slot]; sd[tid] = d2t[slot]; sw[tid] = wt[slot];
  }
  __syncthreads();
  if (c >= ld) return;  // (whole waves: ld is a multiple of 32 and group sizes divide 32)
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
    if (!FP) {
#pragma unroll
      for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          float v = av[q] + bv[p];
          if (pre_relu) v = fmaxf(v, 0.f);
          s += v; ss = fmaf(v, v, ss);
        }
    } else {
      __shared__ float sa[16][257];
#pragma unroll
      for (int p = 0; p < 16; ++p) sa[p][tid] = av[p];  // (a thread reads back only its own column: no barrier needed)
#pragma unroll
      for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int sl = p * 8 + j;
          float v = sa[sq[sl]][tid] + bv[p] + sd[sl] * vd + sw[sl] * vw;
          if (pre_relu) v = fmaxf(v, 0.f);
          s += v; ss = fmaf(v, v, ss);
        }
    }
    if (e.mode == SLIDE_EPI_STATS) {
      e.stats_sum[(size_t)b * e.stats_bs + cl] = s * e.stats_scale;
      e.stats_sq[(size_t)b * e.stats_bs + cl] = ss * e.stats_scale;
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
    ta[((size_t)b * 16 + p) * ld + c] = (_Float16)(av[p] * g + sh);
    tb[((size_t)b * 16 + p) * ld + c] = (_Float16)(bv[p] * g);
  }
  if (FP) {
    vv_out[(size_t)b * 2 * ld + c] = vd * g;
    vv_out[(size_t)b * 2 * ld + ld + c] = vw * g;
  }
}

template <bool FP, typename TT = _Float16, int NT = 1024>
__global__ __launch_bounds__(NT) void pair_norm2_kernel(int ld, const float *__restrict__ y, const float *__restrict__ xyz,
                                                          const float *__restrict__ wa, const float *__restrict__ wb,
                                                          const SlideEpi *__restrict__ epi, TT *__restrict__ ta,
                                                          TT *__restrict__ tb, const int *__restrict__ nbr,
                                                          const float *__restrict__ d2t, const float *__restrict__ wt,
                                                          const float *__restrict__ vv_in, float *__restrict__ vv_out,
                                                          const SlideGnFin *__restrict__ finp) {
  extern __shared__ __attribute__((aligned(16))) float dyn_l[];  // FP: a-values [16][NT + 1]
  pair_norm2_body<FP, TT, NT>(ld, y, xyz, wa, wb, epi, ta, tb, nbr, d2t, wt, vv_in, vv_out, finp, blockIdx.x, blockDim.x, dyn_l);
}

template <int NPXL, int NST, int MODE, int CBW = 4, bool OCC3 = true>
int launch_gx(const GemmArgs &a, hipStream_t s) {
  constexpr int NSAMP = TM >> NPXL, NVEC = (MODE ? 2 : 1) + (NPXL == 7 ? 2 : 0);
  const size_t shm = (size_t)NST * (4096 * CBW + 4 * 64 * 16 * NSAMP) + (CBW * EPI_DW + (CBW * EPI_DW) % 4 + CBW * 96) * 4 +
                     (size_t)NSAMP * NVEC * a.k_pad * 2 + 16;
  // the OCC3 form needs three workgroups per CU, the others two (the tables stream through the ring, so only the per-sample
  // vectors grow with K)
  if (shm > (CBW == 2 && OCC3 ? 53 : 80) * 1024) return -8;
  const int ntc = (a.n_cob + CBW - 1) / CBW, ntr = (a.rows + TM - 1) / TM;
  const int grid = ((ntr + 7) / 8) * 8 * ntc;
  static bool attr_done[SLIDE_MAX_DEVICES] = {};
  int d = 0;
  (void)hipGetDevice(&d);
  bool &attr_set = attr_done[d >= 0 && d < SLIDE_MAX_DEVICES ? d : 0];
  GemmArgs b = a;
  b.shm_bytes = (int)shm;
  if constexpr (CBW == 2 && !OCC3) {
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_gx_n64w_kernel<NPXL, NST, MODE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_set = true;
    }
    hipLaunchKernelGGL((gemm_gx_n64w_kernel<NPXL, NST, MODE>), dim3(grid), dim3(256), shm, s, b);
  } else if constexpr (CBW == 2) {
    // (red / gsh of the common epilogue live in the dead ring: 256 CBW + 128 CBW floats = 3 KB < the ring)
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_gx_n64_kernel<NPXL, NST, MODE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
      attr_set = true;
    }
    hipLaunchKernelGGL((gemm_gx_n64_kernel<NPXL, NST, MODE>), dim3(grid), dim3(256), shm, s, b);
  } else {
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_gx_kernel<NPXL, NST, MODE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_set = true;
    }
    hipLaunchKernelGGL((gemm_gx_kernel<NPXL, NST, MODE>), dim3(grid), dim3(256), shm, s, b);
  }
  return (int)hipGetLastError();
}

}  // namespace

int slide_launch_gemm_gx(const SlideOp &o, hipStream_t s);

static int gx_args_from_op(const SlideOp &o, GemmArgs &a) {
  a = GemmArgs();
  a.gx_ta = o.p[0]; a.W = o.p[1]; a.epi = (const SlideEpi *)o.p[2];
  a.in_scale = (const float *)o.p[3]; a.in_shift = (const float *)o.p[4];
  a.gx_tb = o.p[5]; a.in_add = (const float *)o.p[6]; a.gx_add_idx = (const int *)o.p[7];
  a.gidx = (const int *)o.p[8]; a.gx_d2 = (const float *)o.p[9]; a.gx_w = (const float *)o.p[10];
  a.gx_vv = (const float *)o.p[11];
  a.dbg = (unsigned long long *)o.p[12];
  a.rows = o.i[0]; a.gx_ld = o.i[1]; a.k_pad = o.i[2]; a.n_cob = o.i[3]; a.in_bs = o.i[5];
  a.gx_mode = o.i[6]; a.add_bs = o.i[7]; a.gx_add_idx_stride = o.i[8]; a.gx_vbs = o.i[9];
  a.w_cm = 1; a.x_ld = 32;
  resolve_epi(a);
  const int npxl = o.i[4];
  if (a.k_pad % 32 || a.k_pad <= 0 || a.gx_ld % 8 || a.rows <= 0 || a.n_cob <= 0 || !a.gx_ta || !a.gx_tb) return -3;
  if (a.gx_mode != 0 && (!a.in_scale || !a.in_shift)) return -3;
  if (a.in_add && ((uintptr_t)a.in_add % 16 || a.add_bs % 4 || a.gx_add_idx_stride % 4)) return -3;  // 16-byte vector loads
  if (a.gx_mode != 0 && ((uintptr_t)a.in_scale % 16 || (uintptr_t)a.in_shift % 16 || a.in_bs % 4)) return -3;
  if (a.gx_vv && ((uintptr_t)a.gx_vv % 16 || a.gx_vbs % 8)) return -3;
  if (npxl == 7 && (!a.gidx || !a.gx_d2 || !a.gx_w)) return -3;
  return 0;
}

template <int NPXL>
static int launch_gx_dual(const GemmArgs &a1, const GemmArgs &a0, hipStream_t s) {
  constexpr int NSAMP = TM >> NPXL;
  auto lds = [&](const GemmArgs &a, int nvec) {
    return (size_t)GX_DUAL_NST<NPXL> * (8192 + 4 * 64 * 16 * NSAMP) + (2 * EPI_DW + (2 * EPI_DW) % 4 + 2 * 96) * 4 + (size_t)NSAMP * nvec * a.k_pad * 2 + 16;
  };
  const size_t s1 = lds(a1, 2 + (NPXL == 7 ? 2 : 0)), s0 = lds(a0, 1 + (NPXL == 7 ? 2 : 0));
  const size_t shm = s1 > s0 ? s1 : s0;
  if (shm > (NPXL == 7 ? 53 : 80) * 1024) return -8;  // (three / two workgroups per CU)
  const int ntr = (a1.rows + TM - 1) / TM;
  const int g1 = ((ntr + 7) / 8) * 8 * ((a1.n_cob + 1) / 2), g0 = ((ntr + 7) / 8) * 8 * ((a0.n_cob + 1) / 2);
  static bool attr_done[SLIDE_MAX_DEVICES] = {};
  int d = 0;
  (void)hipGetDevice(&d);
  bool &attr_set = attr_done[d >= 0 && d < SLIDE_MAX_DEVICES ? d : 0];
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_gx_dual_kernel<NPXL>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    attr_set = true;
  }
  GemmArgs b1 = a1, b0 = a0;
  b1.shm_bytes = b0.shm_bytes = (int)shm;
  hipLaunchKernelGGL((gemm_gx_dual_kernel<NPXL>), dim3(g1 + g0), dim3(256), shm, s, b1, b0, g1);
  return (int)hipGetLastError();
}

// SLIDE_OP_GEMM_GX_DUAL: p[0] = HOST pointer to two SlideOp (SLIDE_OP_GEMM_GX: mode 1, then mode 0) of the same block
int slide_launch_gemm_gxs_dual(const SlideOp *pr, hipStream_t s);  // gemm_gxs.hip (split arithmetic)
int slide_launch_gemm_gx_dual(const SlideOp &o, hipStream_t s) {
  const SlideOp *pr = (const SlideOp *)o.p[0];
  if (!pr || pr[0].kind != SLIDE_OP_GEMM_GX || pr[1].kind != SLIDE_OP_GEMM_GX) return -3;
  if ((int)pr[0].f[0] == 3 && (int)pr[1].f[0] == 3) return slide_launch_gemm_gxs_dual(pr, s);
  GemmArgs a1, a0;
  int st = gx_args_from_op(pr[0], a1);
  if (st == 0) st = gx_args_from_op(pr[1], a0);
  if (st != 0) return st;
  if (a1.gx_mode == 0 || a0.gx_mode != 0 || a1.rows != a0.rows || pr[0].i[4] != pr[1].i[4]) return -3;
  if (pr[0].i[4] == 8) st = launch_gx_dual<8>(a1, a0, s);
  else if (pr[0].i[4] == 7) st = launch_gx_dual<7>(a1, a0, s);
  else return -4;
  if (st == -8) {  // (LDS of the dual form does not fit: two launches)
    st = slide_launch_gemm_gx(pr[0], s);
    if (st == 0) st = slide_launch_gemm_gx(pr[1], s);
  }
  return st;
}

int slide_launch_gemm_gx(const SlideOp &o, hipStream_t s) {
  GemmArgs a;
  const int ast = gx_args_from_op(o, a);
  if (ast != 0) return ast;
  const int npxl = o.i[4];
  const bool m1 = a.gx_mode != 0;
  // f[0] != 0 (the plan's default, SLIDE_GX_N64): 256 x 64 tiles, three workgroups per CU (128-channel tiles when the LDS does
  // not fit): measured 385 vs 374 shapes/s in bench.py's arrangement
  // f[0] == 2: the 64-channel tile at two workgroups per CU (both modes; the plan asks for it when the 128-channel grid would
  // leave CUs empty)
  const int n64 = (int)o.f[0];
  // (mode 1 only -- the keys -> u layers: with mode 0's PAIR residual the epilogue does not fit 168 registers)
  int st = -8;
  if (n64 == 1 && m1) {  // 64-channel tiles at three workgroups per CU (a two-stage ring of this form spills at 168 registers)
    if (npxl == 8) st = launch_gx<8, 3, 1, 2>(a, s);
    else if (npxl == 7) st = launch_gx<7, 3, 1, 2>(a, s);
    if (st != -8) return st;
  }
  if (n64 == 2) {
    if (npxl == 8) st = m1 ? launch_gx<8, 3, 1, 2, false>(a, s) : launch_gx<8, 3, 0, 2, false>(a, s);
    else if (npxl == 7) st = m1 ? launch_gx<7, 3, 1, 2, false>(a, s) : launch_gx<7, 3, 0, 2, false>(a, s);
    if (st != -8) return st;
  }
  // 128-channel tiles at two workgroups per CU: three ring stages, else two
  if (npxl == 8) {
    st = m1 ? launch_gx<8, 3, 1>(a, s) : launch_gx<8, 3, 0>(a, s);
    if (st == -8) st = m1 ? launch_gx<8, 2, 1>(a, s) : launch_gx<8, 2, 0>(a, s);
    return st;
  }
  if (npxl == 7) {
    st = m1 ? launch_gx<7, 3, 1>(a, s) : launch_gx<7, 3, 0>(a, s);
    if (st == -8) st = m1 ? launch_gx<7, 2, 1>(a, s) : launch_gx<7, 2, 0>(a, s);
    return st;
  }
  return -4;
}

// SLIDE_OP_PAIR_NORM: the two-launch form of the pair-table pass (SLIDE_PAIR_FUSED=0) and its one-workgroup-per-sample version
// (SLIDE_PAIR_NORM_V2=1) -- experiments build only; the default plans run SLIDE_OP_PAIR_FIRST (engine.hip)
// version 2 with FLOAT tables (i[4] == 1): the pair-table pass of the split-arithmetic plans (product build)
static int launch_pair_norm2_f32(const SlideOp &o, hipStream_t s) {
  const int B = o.i[0], ld = o.i[1], K = o.i[2];
  if (B <= 0 || ld <= 0 || ld % 32 || ld > 2048) return -3;
  const SlideGnFin *fin = (const SlideGnFin *)o.p[12];
  // (512-thread workgroups: 256 registers per thread -- with 1024 the 16 x 16 pair loop over two 16-entry register tables spills)
  const int npass = (ld + 511) / 512, nthr = ((ld + npass - 1) / npass + 63) / 64 * 64;
  const dim3 grid(B), blk(nthr < 512 ? nthr : 512);
  if (K == 16)
    hipLaunchKernelGGL((pair_norm2_kernel<false, float, 512>), grid, blk, 0, s, ld, (const float *)o.p[0], (const float *)o.p[1],
                       (const float *)o.p[2], (const float *)o.p[3], (const SlideEpi *)o.p[4], (float *)o.p[5], (float *)o.p[6],
                       (const int *)nullptr, (const float *)nullptr, (const float *)nullptr, (const float *)nullptr,
                       (float *)nullptr, fin);
  else if (K == 8) {
    if (!o.p[7] || !o.p[8] || !o.p[9] || !o.p[10] || !o.p[11]) return -3;
    static bool attr_done[64] = {};
    int d = 0;
    (void)hipGetDevice(&d);
    d = d >= 0 && d < 64 ? d : 0;
    if (!attr_done[d]) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&pair_norm2_kernel<true, float, 512>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 513 * 4);
      attr_done[d] = true;
    }
    hipLaunchKernelGGL((pair_norm2_kernel<true, float, 512>), grid, blk, (size_t)16 * 513 * 4, s, ld, (const float *)o.p[0],
                       (const float *)o.p[1], (const float *)o.p[2], (const float *)o.p[3], (const SlideEpi *)o.p[4],
                       (float *)o.p[5], (float *)o.p[6], (const int *)o.p[7], (const float *)o.p[8], (const float *)o.p[9],
                       (const float *)o.p[10], (float *)o.p[11], fin);
  } else return -5;
  return (int)hipGetLastError();
}

#ifndef SLIDE_EXPERIMENTS
int slide_launch_pair_norm(const SlideOp &o, hipStream_t s) {
  if (o.i[3] == 2 && o.i[4] == 1) return launch_pair_norm2_f32(o, s);
  return -20;
}
#else
int slide_launch_pair_norm(const SlideOp &o, hipStream_t s) {
  if (o.i[3] == 2 && o.i[4] == 1) return launch_pair_norm2_f32(o, s);
  const int B = o.i[0], ld = o.i[1], K = o.i[2];
  if (B <= 0 || ld <= 0 || ld % 32) return -3;
  const SlideGnFin *fin = (const SlideGnFin *)o.p[12];
  if (o.i[3] == 2) {  // version 2: one workgroup per sample, joint [query | key] GroupNorm finalised here (p[12], may be NULL)
    if (ld > 2048) return -3;
    // passes of equal size: every thread handles ceil(ld / nthr) channels (1056 channels: 2 x 544, not 1024 + 32)
    const int npass = (ld + 1023) / 1024, nthr = ((ld + npass - 1) / npass + 63) / 64 * 64;
    const dim3 grid(B), blk(nthr < 1024 ? nthr : 1024);
    if (K == 16)
      hipLaunchKernelGGL(pair_norm2_kernel<false>, grid, blk, 0, s, ld, (const float *)o.p[0], (const float *)o.p[1],
                         (const float *)o.p[2], (const float *)o.p[3], (const SlideEpi *)o.p[4], (_Float16 *)o.p[5],
                         (_Float16 *)o.p[6], (const int *)nullptr, (const float *)nullptr, (const float *)nullptr,
                         (const float *)nullptr, (float *)nullptr, fin);
    else if (K == 8) {
      if (!o.p[7] || !o.p[8] || !o.p[9] || !o.p[10] || !o.p[11]) return -3;
      static bool attr_done[64] = {};
      int d = 0;
      (void)hipGetDevice(&d);
      d = d >= 0 && d < 64 ? d : 0;
      if (!attr_done[d]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&pair_norm2_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 1025 * 4);
        attr_done[d] = true;
      }
      hipLaunchKernelGGL(pair_norm2_kernel<true>, grid, blk, (size_t)16 * 1025 * 4, s, ld, (const float *)o.p[0],
                         (const float *)o.p[1], (const float *)o.p[2], (const float *)o.p[3], (const SlideEpi *)o.p[4],
                         (_Float16 *)o.p[5], (_Float16 *)o.p[6], (const int *)o.p[7], (const float *)o.p[8],
                         (const float *)o.p[9], (const float *)o.p[10], (float *)o.p[11], fin);
    } else return -5;
    return (int)hipGetLastError();
  }
  const dim3 grid((ld + 255) / 256, B), blk(256);
  if (K == 16)
    hipLaunchKernelGGL(pair_norm_kernel<false>, grid, blk, 0, s, ld, (const float *)o.p[0], (const float *)o.p[1],
                       (const float *)o.p[2], (const float *)o.p[3], (const SlideEpi *)o.p[4], (_Float16 *)o.p[5],
                       (_Float16 *)o.p[6], (const int *)nullptr, (const float *)nullptr, (const float *)nullptr,
                       (const float *)nullptr, (float *)nullptr);
  else if (K == 8) {
    if (!o.p[7] || !o.p[8] || !o.p[9] || !o.p[10] || !o.p[11]) return -3;
    hipLaunchKernelGGL(pair_norm_kernel<true>, grid, blk, 0, s, ld, (const float *)o.p[0], (const float *)o.p[1],
                       (const float *)o.p[2], (const float *)o.p[3], (const SlideEpi *)o.p[4], (_Float16 *)o.p[5],
                       (_Float16 *)o.p[6], (const int *)o.p[7], (const float *)o.p[8], (const float *)o.p[9],
                       (const float *)o.p[10], (float *)o.p[11]);
  } else return -5;
  return (int)hipGetLastError();
}
#endif

namespace {

// ================================================================================================ fused SA Mlp chain
// second_mlp -> rest_mlp of an SA block's Mlp_plus_t_emb (pointnet2_modules.py:119-176) in ONE launch, one workgroup per
// sample (16 x 16 rows, natural neighbour order), eight waves x 32 rows:
//   stage 1: h2 = relu(GN(W1 . h1 + b1)) + class-embedding vector, h1 GENERATED from the pair tables (mode 0 above);
//            every wave owns ALL n1 channels of its 32 rows, so after the GroupNorm (statistics across the waves through
//            LDS) its accumulators, converted to fp16 and re-paired with v_permlane32_swap, ARE the MFMA B fragments of
//   stage 2: mo = relu(GN(W2 . h2 + b2)) + pair residual, in slabs of 256 channels, stored chunk-major for the attention tail.
// h2 never exists in memory; one weight ring (16 KB chunk images of 256 channels x 32 K) streams W1 then the slabs of W2 and
// keeps running ahead across the epilogues.  GroupNorm group sizes 4 / 8 / 16 (widths 128 / 256 / 512).
struct F1Args {
  const void *ta, *tb, *ra, *rb;  // fp16 pair tables [B*16][t_ld]: first_mlp columns (pre-normalised), res_connect columns
  const void *W1, *W2;            // chunk-major [k1/32][n1][32], [n1/32][n2][32]
  const float *vec1, *vec2;       // [bias | gamma | beta][n]
  const float *add0;              // h1 add vector (t-embedding): add0[idx * add0_stride + b * add0_bs + k] or NULL
  const int *add0_idx;
  const float *add1;              // h2 add vector (class embedding): add1[b * add1_bs + c] or NULL
  void *out;                      // mo, chunk-major [n2/32][B*256][32] fp16
  unsigned long long *dbg;
  int B, t_ld, k1, n1, n2, gs1, gs2, add0_stride, add0_bs, add1_bs;
  float inv1, inv2;
  int out_fm;  // round 6: out is fragment-major (SLIDE_F_OUT_FM)
  int nsplit;  // round 6: workgroups per sample (1 | 2): each takes n2 / 256 / nsplit of the stage-2 slabs (stage 1 is computed by all)
};

// sum over the 32 lanes of a half wave; the total is valid in the half's UPPER 16 lanes (col >= 16).  Pure DPP: four steps
// inside the 16-lane rows, then row_bcast15 carries row 0 / 2's total into row 1 / 3 (no LDS swizzle round trip).
__device__ __forceinline__ float half_wave_sum_hi(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));   // xor 1
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));   // xor 2
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));  // row_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false)); // row_bcast15 -> rows 1, 3
  return v;
}

// workgroup barrier that orders LDS traffic only: __syncthreads() would also drain the VM counter, i.e. wait for the weight
// ring's DMA in flight (and for the epilogue's own stores)
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int NB1>  // 32-channel blocks of h2 (4 or 8)
__device__ __forceinline__ void sa_chain_body(const F1Args &a, const int bid) {
  using T = _Float16;
  constexpr int NB2 = 8;                       // blocks per stage-2 slab
  constexpr int CH_B = 256 * 64, STAGE_B = 2 * CH_B, NST = 2;
  constexpr int NR = 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // SLAB SPLIT (round 6): with few samples per launch (one 8-wave workgroup per sample: 88 of 256 CUs at bench.py's sub-batch size)
  // a sample's stage-2 slabs go to nsplit workgroups -- each repeats stage 1 (a third of the chain's FLOPs at n2 = 512) and runs
  // its own slabs: 1.5x the matrix work per sample on twice the CUs, a shorter critical path for the chain that waits on this launch
  const int b = bid / a.nsplit, sl0 = (bid % a.nsplit) * ((a.n2 >> 8) / a.nsplit);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, col = lane & 31;
  const int nk1 = a.k1 >> 5, nk2 = (NB1 * 32) >> 5, nslab = (a.n2 >> 8) / a.nsplit;
  const int nchunks = nk1 + nslab * nk2, nks = nchunks >> 1;  // (nk1, nk2 even)
  unsigned char *const ring = smem_raw;
  float *const vec1_l = reinterpret_cast<float *>(smem_raw + (size_t)NST * STAGE_B);  // [3][n1]
  float *const vec2_l = vec1_l + 3 * a.n1;                                            // [3][n2]
  float *const add1_l = vec2_l + 3 * a.n2;                                            // [n1]
  float *const red = add1_l + a.n1;                                                   // [8 waves][8 cb][2 halves][4][2]
  float *const gsh = red + 8 * 8 * 2 * 4 * 2;                                         // [8 cb][2][32]
  T *const add0_l = reinterpret_cast<T *>(gsh + 8 * 2 * 32);                          // [k1] fp16
  unsigned char *const ta_l = reinterpret_cast<unsigned char *>(add0_l + a.k1);
  const int tab_b = (a.k1 >> 3) * NR * 16;
  unsigned char *const tb_l = ta_l + tab_b;
  unsigned char *const ra_l = tb_l + tab_b;                 // res_connect tables, same image: [piece][16 rows][16 B]
  unsigned char *const rb_l = ra_l + (a.n2 >> 3) * NR * 16;
#ifdef SLIDE_TIMELINE
  if (a.dbg && tid == 0) a.dbg[(size_t)b * 16 + 0] = wall_clock64();
#define F1_STAMP(k) do { if (a.dbg && tid == 0) a.dbg[(size_t)b * 16 + (k)] = wall_clock64(); } while (0)
#else
#define F1_STAMP(k) do { } while (0)
#endif

  // ---- pair tables by LDS-DMA: [piece][16 rows][16 B]; one instruction = 4 pieces x 16 rows
  {
    const int r = lane & 15, pl = lane >> 4;
    const size_t grow = ((size_t)b * 16 + r) * a.t_ld;
    const int nins = (a.k1 >> 3) / 4;
    for (int i = wave; i < 2 * nins; i += 8) {
      const int t = i >= nins, ii = t ? i - nins : i;
      const T *src = reinterpret_cast<const T *>(t ? a.tb : a.ta) + grow + (ii * 4 + pl) * 8;
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)src,
                                       (__attribute__((address_space(3))) void *)((t ? tb_l : ta_l) + ii * 1024), 16, 0, 0);
    }
    const int rins = (a.n2 >> 3) / 4;
    for (int i = wave; i < 2 * rins; i += 8) {
      const int t = i >= rins, ii = t ? i - rins : i;
      const T *src = reinterpret_cast<const T *>(t ? a.rb : a.ra) + grow + (ii * 4 + pl) * 8;
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)src,
                                       (__attribute__((address_space(3))) void *)((t ? rb_l : ra_l) + ii * 1024), 16, 0, 0);
    }
  }
  // ---- weight ring: chunk g = W1 chunk g (g < nk1) or chunk (g - nk1) % nk2 of slab (g - nk1) / nk2 of W2; image [256][64 B]
  // 16 DMA instructions per chunk, 2 per wave: instruction j * 8 + wave carries image rows 16 (j * 8 + wave) ..
  int wtrow[2], wpiece[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    wtrow[j] = 16 * (j * 8 + wave) + (lane >> 2);
    wpiece[j] = (lane & 3) ^ ((wtrow[j] >> 2) & 3);
  }
  auto issue = [&](int st) __attribute__((always_inline)) {
    unsigned char *dst = ring + (size_t)(st % NST) * STAGE_B;
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
      const int g = st * 2 + c2;
      const T *base;
      int nrow, kc, r0;
      if (g < nk1) { base = reinterpret_cast<const T *>(a.W1); nrow = a.n1; kc = g; r0 = 0; }
      else {
        const int h = g - nk1, sl = h / nk2;
        base = reinterpret_cast<const T *>(a.W2); nrow = a.n2; kc = h - sl * nk2; r0 = (sl0 + sl) * 256;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int row = r0 + wtrow[j];
        row = row < nrow ? row : nrow - 1;  // (stage 1 with n1 = 128: the image's upper half is never read)
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)(base + ((size_t)kc * nrow + row) * 32 + wpiece[j] * 8),
                                         (__attribute__((address_space(3))) void *)(dst + c2 * CH_B + (j * 8 + wave) * 1024),
                                         16, 0, 0);
      }
    }
  };
  issue(0);
  // ---- vectors.  PROLOGUE ORDER (round 6): the table DMAs and the first ring stage above are in flight before anything waits; the fp32
  // vectors go by LDS-DMA as well (four bytes per lane), only the t-embedding row (fp16 in LDS, behind a device-side row index) takes
  // plain loads -- one memory round trip + the index instead of four in a row (timeline "prologue": 1.8 - 2.3 us, and the first stage's
  // wait behind it)
  {
    const int w64 = wave * 64;
    auto dma4 = [&](const float *src, float *dst, int n) __attribute__((always_inline)) {
      for (int i0 = w64; i0 < n; i0 += 512)
        if (i0 + lane < n)
          __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)(src + i0 + lane), (__attribute__((address_space(3))) void *)(dst + i0), 4, 0, 0);
    };
    dma4(a.vec1, vec1_l, 3 * a.n1);
    dma4(a.vec2, vec2_l, 3 * a.n2);
    if (a.add1) dma4(a.add1 + (size_t)b * a.add1_bs, add1_l, a.n1);
    else for (int i = tid; i < a.n1; i += 512) add1_l[i] = 0.f;
  }
  {
    const float *addp = a.add0;
    if (addp && a.add0_idx) addp += (size_t)a.add0_idx[0] * a.add0_stride;
    for (int i = tid; i < a.k1; i += 512) add0_l[i] = (T)(addp ? addp[(size_t)b * a.add0_bs + i] : 0.f);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  F1_STAMP(14);
  // stage st must have landed before anyone reads it (NST = 2: nothing else is in flight)
  auto stage_ready = [&](int st) __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (st + 1 < nks) issue(st + 1);  // overwrites the stage consumed at st - 1
  };

  // this wave's 32 rows: points 2 wave, 2 wave + 1; neighbour q = col & 15
  const int aoff = (half * NR + (col & 15)) * 16, boff = (half * NR + 2 * wave + (col >> 4)) * 16;
  int wrow[NB2], wkey[NB2];
#pragma unroll
  for (int cb = 0; cb < NB2; ++cb) {
    const int trow = cb * 32 + col;
    wrow[cb] = trow * 64; wkey[cb] = (trow >> 2) & 3;
  }

  // ================================================================================================ stage 1
  // accumulators start from the bias: the LDS reads land in the accumulator registers, the epilogue has no bias pass
  auto init_acc = [&](auto nb_tag, f32x16 *v, const float *bias_l) __attribute__((always_inline)) {
    constexpr int NB = decltype(nb_tag)::value;
#pragma unroll
    for (int cb = 0; cb < NB; ++cb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 bia = *reinterpret_cast<const float4 *>(bias_l + cb * 32 + 8 * q + 4 * half);
        v[cb][4 * q] = bia.x; v[cb][4 * q + 1] = bia.y; v[cb][4 * q + 2] = bia.z; v[cb][4 * q + 3] = bia.w;
      }
  };
  f32x16 acc1[NB1];
  struct Step1 { f16x8 af[NB1], av, bv; int kb; };
  auto load1 = [&](Step1 &o, const unsigned char *sb, int c2, int st2, int kc) __attribute__((always_inline)) {
    const int piece = st2 * 2 + half;
#pragma unroll
    for (int cb = 0; cb < NB1; ++cb)
      o.af[cb] = *reinterpret_cast<const f16x8 *>(sb + c2 * CH_B + wrow[cb] + ((piece ^ wkey[cb]) << 4));
    const int pb = (kc * 4 + st2 * 2) * NR * 16;
    o.kb = (kc * 32 + st2 * 16 + half * 8) * 2;
    o.av = *reinterpret_cast<const f16x8 *>(ta_l + pb + aoff);
    o.bv = *reinterpret_cast<const f16x8 *>(tb_l + pb + boff);
  };
  auto compute1 = [&](const Step1 &o) __attribute__((always_inline)) {
    const f16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    const f16x8 v0 = *reinterpret_cast<const f16x8 *>(reinterpret_cast<const unsigned char *>(add0_l) + o.kb);
    const f16x8 bf = __builtin_elementwise_max(o.av + o.bv, zero) + v0;
#pragma unroll
    for (int cb = 0; cb < NB1; ++cb) acc1[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.af[cb], bf, acc1[cb], 0, 0, 0);
  };
  const int nks1 = nk1 >> 1;
  {
    Step1 cur, nxt;
    stage_ready(0);  // (its barrier publishes the vectors staged above)
    init_acc(std::integral_constant<int, NB1>(), acc1, vec1_l);
    load1(cur, ring, 0, 0, 0);
    for (int st = 0; st < nks1; ++st) {
      const unsigned char *sb = ring + (size_t)(st % NST) * STAGE_B;
      load1(nxt, sb, 0, 1, st * 2);
      compute1(cur);
      load1(cur, sb, 1, 0, st * 2 + 1);
      compute1(nxt);
      load1(nxt, sb, 1, 1, st * 2 + 1);
      compute1(cur);
      if (st + 1 < nks1) {
        stage_ready(st + 1);
        load1(cur, ring + (size_t)((st + 1) % NST) * STAGE_B, 0, 0, st * 2 + 2);
      }
      compute1(nxt);
    }
  }
  F1_STAMP(1);
  // (the ring runs one stage ahead: stage_ready(s) issued stage s + 1, so the first chunks of W2 land under the epilogue below)

  // GroupNorm of a stage: acc blocks v[NB] (D layout: reg r -> channel (r & 3) + 8 (r >> 2) + 4 half of the block, lane col ->
  // row; bias included since the accumulator init); statistics over the sample's 256 rows x gs channels.  Leaves per-channel
  // scale / shift in gsh.  Written on register PAIRS (packed fp32 VALU ops): these epilogues are VALU-issue bound.
  auto group_stats = [&](auto nb_tag, f32x16 *v, const float *vec_l, int cb_base, int n, int gs, float inv_count) __attribute__((always_inline)) {
    constexpr int NB = decltype(nb_tag)::value;
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
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
        s[q] = half_wave_sum_hi(s[q]);
        ss[q] = half_wave_sum_hi(ss[q]);
      }
      if (col == 31) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<f32x2 *>(red + ((((wave * 8 + cb) * 2 + half) * 4 + q) * 2)) = f32x2{s[q], ss[q]};
      }
    }
    lds_barrier();
    // one wave per channel block: channel c = lane (lower half), sums its group's partials over the 8 waves
    if (wave < NB && half == 0) {
      const int cb = wave, c = col, q = c >> 3, hh = (c >> 2) & 1;
      f32x2 t = {0.f, 0.f};
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        const float *pw = red + (((w * 8 + cb) * 2) * 4) * 2;
        if (gs == 4) t += *reinterpret_cast<const f32x2 *>(pw + ((hh * 4 + q) * 2));
        else if (gs == 8) t += *reinterpret_cast<const f32x2 *>(pw + (q * 2)) + *reinterpret_cast<const f32x2 *>(pw + ((4 + q) * 2));
        else t += *reinterpret_cast<const f32x2 *>(pw + ((q & 2) * 2)) + *reinterpret_cast<const f32x2 *>(pw + ((4 + (q & 2)) * 2));
      }
      const float mean = t[0] * inv_count;
      const float var = fmaxf(t[1] * inv_count - mean * mean, 0.f);
      const float g = vec_l[n + (cb_base + cb) * 32 + c] * __builtin_amdgcn_rsqf(var + GN_EPS);
      gsh[(cb * 2 + 0) * 32 + c] = g;
      gsh[(cb * 2 + 1) * 32 + c] = vec_l[2 * n + (cb_base + cb) * 32 + c] - mean * g;
    }
    lds_barrier();
  };
  // normalise block cb (fp32, packed), convert to fp16, ReLU on the packed halves, [+ addp: 16 fp32 per lane in the D layout's
  // channel order], and re-pair between the lane halves: o[p] = the 8 consecutive channels 16 p + 8 half of the lane's row
  auto norm_pack = [&](const f32x16 &v, int cb, const float *addp, f16x8 (&o)[2]) __attribute__((always_inline)) {
    uint32_t u[8];
    const f16x2 zero2 = {0, 0};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 g = *reinterpret_cast<const float4 *>(gsh + (cb * 2 + 0) * 32 + 8 * q + 4 * half);
      const float4 sh = *reinterpret_cast<const float4 *>(gsh + (cb * 2 + 1) * 32 + 8 * q + 4 * half);
      f32x2 lo = __builtin_elementwise_fma(f32x2{v[4 * q], v[4 * q + 1]}, f32x2{g.x, g.y}, f32x2{sh.x, sh.y});
      f32x2 hi = __builtin_elementwise_fma(f32x2{v[4 * q + 2], v[4 * q + 3]}, f32x2{g.z, g.w}, f32x2{sh.z, sh.w});
      f16x2 l2 = __builtin_elementwise_max(__builtin_convertvector(lo, f16x2), zero2);
      f16x2 h2 = __builtin_elementwise_max(__builtin_convertvector(hi, f16x2), zero2);
      if (addp) {
        const float4 ad = *reinterpret_cast<const float4 *>(addp + 8 * q + 4 * half);
        l2 += __builtin_convertvector(f32x2{ad.x, ad.y}, f16x2);
        h2 += __builtin_convertvector(f32x2{ad.z, ad.w}, f16x2);
      }
      u[2 * q] = __builtin_bit_cast(uint32_t, l2);
      u[2 * q + 1] = __builtin_bit_cast(uint32_t, h2);
    }
    // quads 2p (u[4p], u[4p+1]) <-> 2p + 1 (u[4p+2], u[4p+3]) across the lane halves: four swaps behind one hazard pad
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\t"
                 "v_permlane32_swap_b32 %4, %6\n\tv_permlane32_swap_b32 %5, %7\n\ts_nop 1"
                 : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]));
    o[0] = __builtin_bit_cast(f16x8, u32x4{u[0], u[1], u[2], u[3]});
    o[1] = __builtin_bit_cast(f16x8, u32x4{u[4], u[5], u[6], u[7]});
  };

  group_stats(std::integral_constant<int, NB1>(), acc1, vec1_l, 0, a.n1, a.gs1, a.inv1);
  f16x8 h2b[2 * NB1];  // B fragments of stage 2: K block kb of 16 channels, this lane's row
#pragma unroll
  for (int cb = 0; cb < NB1; ++cb) {
    f16x8 o[2];
    norm_pack(acc1[cb], cb, add1_l + cb * 32, o);
    h2b[2 * cb] = o[0]; h2b[2 * cb + 1] = o[1];
  }
  F1_STAMP(2);

  // ================================================================================================ stage 2
  const size_t R = (size_t)a.B * 256;
  const int row = b * 256 + wave * 32 + col;
  const int raoff = (col & 15) * 16, rboff = (2 * wave + (col >> 4)) * 16;  // this lane's rows inside a piece of the residual tables
  for (int sli = 0; sli < nslab; ++sli) {
    const int sl = sl0 + sli;  // the slab's position in the layer; sli: its position in this workgroup's ring sequence
    f32x16 acc2[NB2];
    init_acc(std::integral_constant<int, NB2>(), acc2, vec2_l + sl * 256);
    const int st0 = nks1 + sli * (nk2 >> 1);  // first ring stage of this slab
    f16x8 afc[NB2], afn[NB2];
    auto loadw = [&](f16x8 (&o)[NB2], const unsigned char *sb, int c2, int st2) __attribute__((always_inline)) {
      const int piece = st2 * 2 + half;
#pragma unroll
      for (int cb = 0; cb < NB2; ++cb)
        o[cb] = *reinterpret_cast<const f16x8 *>(sb + c2 * CH_B + wrow[cb] + ((piece ^ wkey[cb]) << 4));
    };
    stage_ready(st0);
    loadw(afc, ring + (size_t)(st0 % NST) * STAGE_B, 0, 0);
#pragma unroll
    for (int s2 = 0; s2 < NB1 / 2; ++s2) {  // ring stages of the slab: 4 K steps of 16 each (K blocks 4 s2 .. 4 s2 + 3)
      const unsigned char *sb = ring + (size_t)((st0 + s2) % NST) * STAGE_B;
      loadw(afn, sb, 0, 1);
#pragma unroll
      for (int cb = 0; cb < NB2; ++cb) acc2[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afc[cb], h2b[4 * s2], acc2[cb], 0, 0, 0);
      loadw(afc, sb, 1, 0);
#pragma unroll
      for (int cb = 0; cb < NB2; ++cb) acc2[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afn[cb], h2b[4 * s2 + 1], acc2[cb], 0, 0, 0);
      loadw(afn, sb, 1, 1);
#pragma unroll
      for (int cb = 0; cb < NB2; ++cb) acc2[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afc[cb], h2b[4 * s2 + 2], acc2[cb], 0, 0, 0);
      if (s2 + 1 < NB1 / 2) {
        stage_ready(st0 + s2 + 1);
        loadw(afc, ring + (size_t)((st0 + s2 + 1) % NST) * STAGE_B, 0, 0);
      }
#pragma unroll
      for (int cb = 0; cb < NB2; ++cb) acc2[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afn[cb], h2b[4 * s2 + 3], acc2[cb], 0, 0, 0);
    }
    F1_STAMP(3 + 3 * sli);
    group_stats(std::integral_constant<int, NB2>(), acc2, vec2_l, sl * NB2, a.n2, a.gs2, a.inv2);
    F1_STAMP(4 + 3 * sli);
#pragma unroll
    for (int cb = 0; cb < NB2; ++cb) {
      const int cg = sl * 256 + cb * 32;
      f16x8 ra4[2], rb4[2];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int pc = ((cg + 16 * p) >> 3) + half;  // 16-byte piece of channels cg + 16 p + 8 half ..
        ra4[p] = *reinterpret_cast<const f16x8 *>(ra_l + pc * NR * 16 + raoff);
        rb4[p] = *reinterpret_cast<const f16x8 *>(rb_l + pc * NR * 16 + rboff);
      }
      f16x8 o[2];
      norm_pack(acc2[cb], cb, nullptr, o);
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const f16x8 y = o[p] + (ra4[p] + rb4[p]);
        // (chunk-major rows, or FRAGMENT-major: SLIDE_F_OUT_FM, include/slide_engine.h -- 1 KB of consecutive memory per instruction)
        const size_t o16 = a.out_fm ? (size_t)(row & ~31) * 32 + p * 512 + half * 256 + (row & 31) * 8 : (size_t)row * 32 + 16 * p + 8 * half;
        *reinterpret_cast<u32x4 *>(reinterpret_cast<T *>(a.out) + (size_t)(cg >> 5) * R * 32 + o16) = __builtin_bit_cast(u32x4, y);
      }
    }
    F1_STAMP(5 + 3 * sli);
  }
#ifdef SLIDE_TIMELINE
  if (a.dbg) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    F1_STAMP(15);
  }
#endif
}

template <int NB1>
__global__ __launch_bounds__(512, 2) void sa_chain_kernel(F1Args a) {
  sa_chain_body<NB1>(a, blockIdx.x);
}

// The fused Mlp chain of an SA block AND the per-point query GEMM of its attention (weight_conv.2's query half with the joint
// GroupNorm finalisation: gemm_small.h, AFF form) in ONE launch (SLIDE_OP_SA_CHAIN_P): both depend on the block's pair tables
// only and nothing on each other.  Blocks [0, grid_sa) run the chain (eight waves); the others run one 64 x 64 tile of the
// query GEMM on their first four waves (the other four exit).  The query GEMM's ~10 us launch + gap disappear under the chain.
template <int NB1>
__global__ __launch_bounds__(512, 2) void sa_chain_p_kernel(F1Args a, GemmArgs g, int grid_sa) {
  if ((int)blockIdx.x < grid_sa) {
    sa_chain_body<NB1>(a, blockIdx.x);
    return;
  }
  if (threadIdx.x >= 256) return;
  small_body<2, true, 0>(g, PairArgs(), (int)blockIdx.x - grid_sa);
}

}  // namespace

// SLIDE_OP_SA_CHAIN (include/slide_engine.h)
static int sa_args_from_op(const SlideOp &o, F1Args &a, size_t &shm) {
  a = F1Args();
  a.ta = o.p[0]; a.tb = o.p[1]; a.ra = o.p[2]; a.rb = o.p[3]; a.W1 = o.p[4]; a.W2 = o.p[5];
  a.vec1 = (const float *)o.p[6]; a.vec2 = (const float *)o.p[7];
  a.add0 = (const float *)o.p[8]; a.add0_idx = (const int *)o.p[9]; a.add1 = (const float *)o.p[10]; a.out = o.p[11];
  a.dbg = (unsigned long long *)o.p[12];
  a.B = o.i[0]; a.t_ld = o.i[1]; a.k1 = o.i[2]; a.n1 = o.i[3]; a.n2 = o.i[4]; a.gs1 = o.i[5]; a.gs2 = o.i[6];
  a.add0_stride = o.i[7]; a.add0_bs = o.i[8]; a.add1_bs = o.i[9];
  a.inv1 = o.f[0]; a.inv2 = o.f[1];
  a.out_fm = o.f[2] != 0.f;
  // OPT-IN (SLIDE_SA_SPLIT_MAX=<samples>: two workgroups per sample for launches of at most that many samples).  Measured in bench.py's
  // arrangement (tools/ab/r06_split.sh, three alternating pairs, --steps 300): 379.1 shapes/s with the split at 88 samples per launch
  // against 387.5 without -- the launch itself gets shorter, but the arrangement is bound by CU-time, not by this launch's latency,
  // and the repeated stage 1 is 1.33x the chain's matrix work.
  static const bool no_split = getenv("SLIDE_SA_SPLIT") && getenv("SLIDE_SA_SPLIT")[0] == '0';
  static const int split_max = getenv("SLIDE_SA_SPLIT_MAX") ? atoi(getenv("SLIDE_SA_SPLIT_MAX")) : 0;
  a.nsplit = (!no_split && (a.n2 >> 8) % 2 == 0 && a.B <= split_max) ? 2 : 1;
  if (a.B <= 0 || a.k1 % 64 || a.k1 <= 0 || (a.n1 != 128 && a.n1 != 256) || a.n2 % 256 || a.n2 <= 0 || a.t_ld % 8) return -3;
  auto okgs = [](int g) { return g == 4 || g == 8 || g == 16; };
  if (!okgs(a.gs1) || !okgs(a.gs2)) return -3;
  shm = (size_t)2 * 32768 + (size_t)(3 * a.n1 + 3 * a.n2 + a.n1 + 8 * 8 * 2 * 4 * 2 + 8 * 2 * 32) * 4 +
        (size_t)a.k1 * 2 + (size_t)2 * (a.k1 >> 3) * 16 * 16 + (size_t)2 * (a.n2 >> 3) * 16 * 16 + 64;
  if (shm > 160 * 1024) return -8;
  return 0;
}

// SLIDE_OP_SA_CHAIN_P: p[0] = HOST pointer to two SlideOp: the SLIDE_OP_SA_CHAIN and the SLIDE_OP_GEMM of the per-point query
// layer (16 rows per sample, fp16, input affine + statistics finalisation: the small-launch kernel's AFF form)
int slide_launch_sa_chain_p(const SlideOp &o, hipStream_t s) {
  const SlideOp *pr = (const SlideOp *)o.p[0];
  if (!pr || pr[0].kind != SLIDE_OP_SA_CHAIN || pr[1].kind != SLIDE_OP_GEMM) return -3;
  F1Args a;
  size_t shm_sa = 0;
  const int st = sa_args_from_op(pr[0], a, shm_sa);
  if (st != 0) return st;
  const SlideOp &q = pr[1];
  GemmArgs g = GemmArgs();
  g.X = q.p[0]; g.W = q.p[1]; g.epi = (const SlideEpi *)q.p[2];
  g.in_scale = (const float *)q.p[3]; g.in_shift = (const float *)q.p[4];
  g.gn_fin = (const SlideGnFin *)q.p[6];
  g.aff_tps = 1;
  g.rows = q.i[0]; g.x_ld = q.i[1]; g.k_pad = q.i[2]; g.n_cob = q.i[3]; g.in_bs = q.i[5];
  resolve_epi(g);
  if (q.i[4] != 4 || q.i[6] != SLIDE_PREC_F16 || (q.i[8] & 6) || !g.in_scale || !g.in_shift || q.p[8] || q.p[9] ||
      q.p[10] || q.p[11] || g.k_pad % 32 || g.x_ld % 8 || g.rows != a.B * 16 || g.n_cob <= 0)
    return -3;
  const int grid_p = ((g.rows + 63) / 64) * ((g.n_cob + 1) / 2);
  const size_t shm_p = (size_t)4 * 2 * 6144 + 2 * (sizeof(SlideEpi) + 96 * 4) + 32 + (size_t)4 * 2 * g.k_pad * 2 + 1024;
  const size_t shm = shm_sa > shm_p ? shm_sa : shm_p;
  static bool attr_done[2][64] = {};
  int d = 0;
  (void)hipGetDevice(&d);
  d = d >= 0 && d < 64 ? d : 0;
  if (a.n1 == 128) {
    if (!attr_done[0][d]) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sa_chain_p_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_done[0][d] = true;
    }
    hipLaunchKernelGGL((sa_chain_p_kernel<4>), dim3(a.B * a.nsplit + grid_p), dim3(512), shm, s, a, g, a.B * a.nsplit);
  } else {
    if (!attr_done[1][d]) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sa_chain_p_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_done[1][d] = true;
    }
    hipLaunchKernelGGL((sa_chain_p_kernel<8>), dim3(a.B * a.nsplit + grid_p), dim3(512), shm, s, a, g, a.B * a.nsplit);
  }
  return (int)hipGetLastError();
}

int slide_launch_sa_chain(const SlideOp &o, hipStream_t s) {
  F1Args a;
  size_t shm = 0;
  const int ast = sa_args_from_op(o, a, shm);
  if (ast != 0) return ast;
  static bool attr_done[2][64] = {};
  int d = 0;
  (void)hipGetDevice(&d);
  d = d >= 0 && d < 64 ? d : 0;
  if (a.n1 == 128) {
    if (!attr_done[0][d]) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sa_chain_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_done[0][d] = true;
    }
    hipLaunchKernelGGL((sa_chain_kernel<4>), dim3(a.B * a.nsplit), dim3(512), shm, s, a);
  } else {
    if (!attr_done[1][d]) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sa_chain_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_done[1][d] = true;
    }
    hipLaunchKernelGGL((sa_chain_kernel<8>), dim3(a.B * a.nsplit), dim3(512), shm, s, a);
  }
  return (int)hipGetLastError();
}
