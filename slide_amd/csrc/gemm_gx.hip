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
#include "gemm_common.h"

namespace {

constexpr int SLIDE_MAX_DEVICES = 64;

// ------------------------------------------------------------------------------------------------ generated-X GEMM
// Tile: 256 rows (one 16x16 sample, or two 16x8 samples) x 128 channels; four waves, wave w owns rows 64 w .. 64 w + 63 and
// all 128 channels (acc[4][2] = 128 registers).  W: chunk-major [k / 32][n_cob * 32][32]; a ring stage is two 32-deep
// chunk images ([128 rows][64 B], source-side XOR swizzle as in the ring kernels of engine.hip) = 16 KB; NST stages.
template <int NPXL, int NST>
__global__ __launch_bounds__(256, 2) void gemm_gx_kernel(GemmArgs a) {
  using T = _Float16;
  constexpr int CBW = 4;
  constexpr int NPX = 1 << NPXL;
  constexpr bool FP = NPXL == 7;
  constexpr int NSAMP = TM >> NPXL;            // 1 or 2
  constexpr int CH_B = 128 * 64, STAGE_B = 2 * CH_B;
  constexpr int NVEC = FP ? 4 : 2;             // fp16 vectors per sample in LDS: [v0 | v1 | vd | vw][k_pad]
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int ntc = (a.n_cob + CBW - 1) / CBW;
  const int ntr = (a.rows + TM - 1) / TM;
  const int xcd = blockIdx.x & 7, q0 = blockIdx.x >> 3;
  const int tc = q0 % ntc, tr = (q0 / ntc) * 8 + xcd;  // the column tiles of a row tile share one XCD's L2 (weights + tables)
  if (tr >= ntr) return;
  const int row0 = tr * TM, cob0 = tc * CBW;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, col = lane & 31;
  const int nk32 = a.k_pad >> 5, nks = (nk32 + 1) >> 1;
  const int pitch = a.k_pad * 2 + 16;          // bytes between table rows in LDS: an ODD number of 16-byte pieces, so the 16
                                               // rows a ds_read_b128 lane group touches fall on 16 different bank slots
  unsigned char *const ring = smem_raw;
  SLIDE_STAMP(a, 0);
  uint32_t *const epi_lds = reinterpret_cast<uint32_t *>(smem_raw + (size_t)NST * STAGE_B);
  float *const vec_lds = reinterpret_cast<float *>(epi_lds + CBW * EPI_DW + (CBW * EPI_DW) % 4);
  unsigned char *const ta_l = reinterpret_cast<unsigned char *>(vec_lds + CBW * 96);
  unsigned char *const tb_l = ta_l + (size_t)NSAMP * 16 * pitch;
  T *const vv_l = reinterpret_cast<T *>(tb_l + (size_t)NSAMP * 16 * pitch);

  // ---- weight ring: this lane's source piece of the wave's two DMA instructions per 32-deep chunk
  const T *wsrc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int trow = 16 * (j * 4 + wave) + (lane >> 2);
    const int piece = (lane & 3) ^ ((trow >> 2) & 3);
    int gco = cob0 * 32 + trow;
    gco = gco < a.n_cob * 32 ? gco : a.n_cob * 32 - 1;  // rows beyond the matrix: clamp (their channels are never stored)
    wsrc[j] = reinterpret_cast<const T *>(a.W) + (size_t)gco * 32 + piece * 8;
  }
  const size_t w_cs = (size_t)a.n_cob * 32 * 32;  // elements between consecutive chunks
  auto issue = [&](int st) __attribute__((always_inline)) {  // stage st -> slot st % NST (4 DMA instructions per wave)
    unsigned char *dst = ring + (size_t)(st % NST) * STAGE_B;
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
      int kc = st * 2 + c2;
      kc = kc < nk32 ? kc : nk32 - 1;  // odd chunk count: the last stage's second image is a dummy (never read)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)(wsrc[j] + (size_t)kc * w_cs),
                                         (__attribute__((address_space(3))) void *)(dst + c2 * CH_B + (j * 4 + wave) * 1024),
                                         16, 0, 0);
    }
  };
#pragma unroll
  for (int s0 = 0; s0 < NST - 1; ++s0)
    if (s0 < nks) issue(s0);

  // ---- tables, vectors, epilogue descriptors (plain loads; the ring's first stages are in flight meanwhile)
  stage_epilogue_tables<CBW, 256>(a, cob0, tid, epi_lds, vec_lds);
  const int nsm = a.rows >> NPXL, smp0 = row0 >> NPXL;
  {
    const int ppr = a.k_pad >> 3;               // 16-byte pieces per table row
    const int cnt = NSAMP * 16 * ppr;
    for (int i = tid; i < cnt; i += 256) {
      const int r = i / ppr, pc = i - r * ppr;  // r = sample slot * 16 + point
      int smp = smp0 + (r >> 4);
      smp = smp < nsm ? smp : nsm - 1;
      const size_t g = ((size_t)smp * 16 + (r & 15)) * a.gx_ld + pc * 8;
      const u32x4 va = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const T *>(a.gx_ta) + g);
      const u32x4 vb = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const T *>(a.gx_tb) + g);
      *reinterpret_cast<u32x4 *>(ta_l + (size_t)r * pitch + pc * 16) = va;
      *reinterpret_cast<u32x4 *>(tb_l + (size_t)r * pitch + pc * 16) = vb;
    }
    const float *addp = a.in_add;
    if (addp && a.gx_add_idx) addp += (size_t)a.gx_add_idx[0] * a.gx_add_idx_stride;  // row t of a per-timestep table
    for (int i = tid; i < NSAMP * a.k_pad; i += 256) {
      const int sl = i / a.k_pad, k = i - sl * a.k_pad;
      int smp = smp0 + sl;
      smp = smp < nsm ? smp : nsm - 1;
      float v0 = 0.f, v1 = 0.f;
      if (a.gx_mode == 0) {
        if (addp) v0 = addp[(size_t)smp * a.add_bs + k];
      } else {
        v0 = a.in_scale[(size_t)smp * a.in_bs + k];
        v1 = a.in_shift[(size_t)smp * a.in_bs + k];
      }
      vv_l[(sl * NVEC + 0) * a.k_pad + k] = (T)v0;
      vv_l[(sl * NVEC + 1) * a.k_pad + k] = (T)v1;
      if (FP) {
        float d = 0.f, w = 0.f;
        if (a.gx_vv) { d = a.gx_vv[(size_t)smp * a.gx_vbs + k]; w = a.gx_vv[(size_t)smp * a.gx_vbs + (a.gx_vbs >> 1) + k]; }
        vv_l[(sl * NVEC + 2) * a.k_pad + k] = (T)d;
        vv_l[(sl * NVEC + 3) * a.k_pad + k] = (T)w;
      }
    }
  }
  SLIDE_STAMP(a, 7);
  // this lane's two row blocks: table rows of the neighbour (a) and of the centre point (b), per-slot scalars
  int aoff[2], boff[2];
  f16x8 d2v[2], wv[2];
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
    aoff[rb] = (sl_w * 16 + q) * pitch + half * 16;
    boff[rb] = (sl_w * 16 + p) * pitch + half * 16;
    const T dh = (T)d2, wh = (T)w;
    d2v[rb] = f16x8{dh, dh, dh, dh, dh, dh, dh, dh};
    wv[rb] = f16x8{wh, wh, wh, wh, wh, wh, wh, wh};
  }
  const unsigned char *const vbase = reinterpret_cast<const unsigned char *>(vv_l + (size_t)sl_w * NVEC * a.k_pad) + half * 16;
  const int vstr = a.k_pad * 2;  // bytes between the vectors of a sample
  const bool mode1 = a.gx_mode != 0;

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

  SLIDE_STAMP(a, 1);
  for (int st = 0; st < nks; ++st) {
    // stage st must have landed; the next one (4 instructions per wave) may stay in flight
    if (st + 1 < nks && NST > 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // (first pass: also publishes the tables staged above)
    if (st + NST - 1 < nks) issue(st + NST - 1);  // overwrites the stage consumed at st - 1
    const unsigned char *sb = ring + (size_t)(st % NST) * STAGE_B;
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
      const int kc = st * 2 + c2;
      if (kc >= nk32) break;
#pragma unroll
      for (int st2 = 0; st2 < 2; ++st2) {
        f16x8 af[CBW], bf[2];
        const int piece = st2 * 2 + half;
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb)
          af[cb] = *reinterpret_cast<const f16x8 *>(sb + c2 * CH_B + wrow[cb] + ((piece ^ wkey[cb]) << 4));
        const int kb = (kc * 32 + st2 * 16) * 2;  // byte offset of this K step inside a table row / vector
        const f16x8 v0 = *reinterpret_cast<const f16x8 *>(vbase + kb);
        f16x8 v1, vd, vw;
        if (mode1) v1 = *reinterpret_cast<const f16x8 *>(vbase + vstr + kb);
        if (FP) {
          vd = *reinterpret_cast<const f16x8 *>(vbase + 2 * vstr + kb);
          vw = *reinterpret_cast<const f16x8 *>(vbase + 3 * vstr + kb);
        }
        const f16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
        f16x8 a0 = *reinterpret_cast<const f16x8 *>(ta_l + aoff[0] + kb);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          f16x8 av = a0;
          if (FP && rb) av = *reinterpret_cast<const f16x8 *>(ta_l + aoff[1] + kb);  // (natural order: q is the same in both blocks)
          const f16x8 bv = *reinterpret_cast<const f16x8 *>(tb_l + boff[rb] + kb);
          f16x8 y = av + bv;
          if (FP) {
            y = __builtin_elementwise_fma(d2v[rb], vd, y);
            y = __builtin_elementwise_fma(wv[rb], vw, y);
          }
          y = __builtin_elementwise_max(y, zero);
          bf[rb] = mode1 ? __builtin_elementwise_fma(y, v0, v1) : y + v0;
        }
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
          for (int rb = 0; rb < 2; ++rb)
            acc[cb][rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[cb], bf[rb], acc[cb][rb], 0, 0, 0);
      }
    }
  }
  __syncthreads();  // every wave is done with the ring before `red` reuses it
  SLIDE_STAMP(a, 2);
  gemm_epilogue<SLIDE_PREC_F16, NPXL, CBW>(a, acc, row0, cob0, wave, half, col, epi_lds, vec_lds,
                                           reinterpret_cast<float *>(smem_raw));
  SLIDE_STAMP(a, 5);
#ifdef SLIDE_TIMELINE
  if (a.dbg) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SLIDE_STAMP(a, 6);
  }
#endif
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

template <int NPXL, int NST>
int launch_gx(const GemmArgs &a, hipStream_t s) {
  constexpr int NSAMP = TM >> NPXL, NVEC = NPXL == 7 ? 4 : 2;
  const size_t shm = (size_t)NST * 16384 + (4 * EPI_DW + (4 * EPI_DW) % 4 + 4 * 96) * 4 +
                     (size_t)2 * NSAMP * 16 * (a.k_pad * 2 + 16) + (size_t)NSAMP * NVEC * a.k_pad * 2 + 16;
  if (shm > 160 * 1024) return -8;
  const int ntc = (a.n_cob + 3) / 4, ntr = (a.rows + TM - 1) / TM;
  const int grid = ((ntr + 7) / 8) * 8 * ntc;
  static bool attr_done[SLIDE_MAX_DEVICES] = {};
  int d = 0;
  (void)hipGetDevice(&d);
  bool &attr_set = attr_done[d >= 0 && d < SLIDE_MAX_DEVICES ? d : 0];
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_gx_kernel<NPXL, NST>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  GemmArgs b = a;
  b.shm_bytes = (int)shm;
  hipLaunchKernelGGL((gemm_gx_kernel<NPXL, NST>), dim3(grid), dim3(256), shm, s, b);
  return (int)hipGetLastError();
}

}  // namespace

int slide_launch_gemm_gx(const SlideOp &o, hipStream_t s) {
  GemmArgs a = {};
  a.gx_ta = o.p[0]; a.W = o.p[1]; a.epi = (const SlideEpi *)o.p[2];
  a.in_scale = (const float *)o.p[3]; a.in_shift = (const float *)o.p[4];
  a.gx_tb = o.p[5]; a.in_add = (const float *)o.p[6]; a.gx_add_idx = (const int *)o.p[7];
  a.gidx = (const int *)o.p[8]; a.gx_d2 = (const float *)o.p[9]; a.gx_w = (const float *)o.p[10];
  a.gx_vv = (const float *)o.p[11];
  a.dbg = (unsigned long long *)o.p[12];
  a.rows = o.i[0]; a.gx_ld = o.i[1]; a.k_pad = o.i[2]; a.n_cob = o.i[3]; a.in_bs = o.i[5];
  a.gx_mode = o.i[6]; a.add_bs = o.i[7]; a.gx_add_idx_stride = o.i[8]; a.gx_vbs = o.i[9];
  a.w_cm = 1; a.x_ld = 32;
  const int npxl = o.i[4];
  if (a.k_pad % 32 || a.k_pad <= 0 || a.gx_ld % 8 || a.rows <= 0 || a.n_cob <= 0 || !a.gx_ta || !a.gx_tb) return -3;
  if (a.gx_mode != 0 && (!a.in_scale || !a.in_shift)) return -3;
  if (npxl == 8) return launch_gx<8, 3>(a, s);
  if (npxl == 7) {
    if (!a.gidx || !a.gx_d2 || !a.gx_w) return -3;
    const int st = launch_gx<7, 3>(a, s);
    return st == -8 ? launch_gx<7, 2>(a, s) : st;
  }
  return -4;
}

int slide_launch_pair_norm(const SlideOp &o, hipStream_t s) {
  const int B = o.i[0], ld = o.i[1], K = o.i[2];
  if (B <= 0 || ld <= 0 || ld % 32) return -3;
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
