// gemm_xs.hip -- the X-stationary GEMM of the fused denoiser plan (dispatched by engine.hip's run_gemm).
#include "gemm_common.h"

namespace {

// ------------------------------------------------------------------------------------------------ X-stationary GEMM
// Sample-wide layers at the model's K (64 .. 288) were bound by L2 -> LDS traffic and per-tile fixed costs: every 256 x 64
// tile re-streams its X rows (256 x K) for 64 output channels, a DMA ring + one barrier per 32-deep chunk paces it, and
// three such workgroups share a CU.  Here ONE workgroup owns a row tile (a whole sample at 256 rows) for ALL output
// channels of the layer:
//   * its X tile is DMA-read ONCE (same swizzled chunk image as the ring kernels; gathered rows / the input affine are
//     applied once, not per column tile) and stays in LDS: <= 144 KB, one workgroup per CU, the full 512-register file;
//   * the weights are used as MFMA A operands only: they are read as host-packed fragments (1 KB per wave load, fully
//     coalesced, L2-resident) straight into VGPRs through a small rolling register ring -- no LDS staging, no barrier in
//     the K loop, the compiler's own vmcnt bookkeeping;
//   * the common epilogue runs per column tile on the same accumulator layout (GroupNorm statistics stay in the
//     workgroup exactly as before), its descriptor tables double-buffered.
// L2 -> LDS bytes per sample and layer fall from N/64 x (256 + 64) K x 2 to 256 K x 2 (5x at N = 512).
template <int NPXL, int CBW, bool AFF, bool GAT>
__global__ __launch_bounds__(256, 1) void gemm_xs_kernel(GemmArgs a, int nstw) {
  using T = _Float16;
  constexpr int ROWB = 64, PPR = 4, RPI = 16, SWS = 2, LPW = 4;  // chunk image: [rows][4 pieces of 16 B]; X: 4 DMA / wave / chunk
  constexpr int CH_B = TM * ROWB;                                 // 16 KB of X per 32-deep chunk
  constexpr int TN = 32 * CBW, WST_B = TN * ROWB, LPWW = TN / 64; // weight stage: TN rows, LPWW DMA instructions / wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, col = lane & 31;
  const int nk = a.k_pad / 32;
  const int ntc = (a.n_cob + CBW - 1) / CBW;
  const int row0 = blockIdx.x * TM;
  constexpr int TAB_DW = CBW * EPI_DW + (CBW * EPI_DW) % 4;  // one set of descriptor tables: dwords, then CBW * 96 floats
  constexpr int TAB_B = (TAB_DW + CBW * 96) * 4;
  constexpr int NSAMP = (1 << NPXL) >= TM ? 1 : TM >> NPXL;
  unsigned char *const wring = smem_raw + (size_t)nk * CH_B;
  unsigned char *const tabs = wring + (size_t)nstw * WST_B;
  _Float16 *const aff_lds = reinterpret_cast<_Float16 *>(tabs + 2 * TAB_B);  // [sample][scale | shift][k_pad]
  float *const red = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(aff_lds) + (AFF ? (size_t)NSAMP * 2 * a.k_pad * 2 : 0));
  SLIDE_STAMP(a, 0);

  // every workgroup walks the column tiles in its own rotation (they all start together: spread the weight panels they
  // ask the L2 for)
  const int rot = blockIdx.x % ntc;
  auto tile_of = [&](int t) { const int v = t + rot; return v >= ntc ? v - ntc : v; };
  // weight ring: chunk g = (tile step tt, chunk kc) -> stage g % nstw; this lane's source row / swizzled piece
  const int G = ntc * nk;
  int wtrow[LPWW], wpiece[LPWW];  // this lane's source row inside a TN-row weight panel and its (swizzled) 16-byte piece
#pragma unroll
  for (int j = 0; j < LPWW; ++j) {
    wtrow[j] = RPI * (j * 4 + wave) + lane / PPR;  // 0 .. TN-1
    wpiece[j] = (lane % PPR) ^ ((wtrow[j] >> SWS) & (PPR - 1));
  }
  // the next chunk to issue is tracked incrementally (tile step wt, chunk wk, stage ws): no division in the K loop
  const T *wsrc[LPWW];
  int wt = 0, wk = 0, ws = 0;
  auto wretile = [&]() __attribute__((always_inline)) {
    const int co0 = tile_of(wt < ntc ? wt : ntc - 1) * TN;
#pragma unroll
    for (int j = 0; j < LPWW; ++j) {
      int co = co0 + wtrow[j];
      co = co < a.n_cob * 32 ? co : a.n_cob * 32 - 1;  // rows beyond the weight matrix: clamp (their channels are never stored)
      wsrc[j] = reinterpret_cast<const T *>(a.W) + (size_t)co * a.k_pad + wpiece[j] * 8;
    }
  };
  wretile();
  auto wissue = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < LPWW; ++j)
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)(wsrc[j] + wk * 32),
                                       (__attribute__((address_space(3))) void *)(wring + (size_t)ws * WST_B + (j * 4 + wave) * 1024),
                                       16, 0, 0);
    ws = ws + 1 == nstw ? 0 : ws + 1;
    if (++wk == nk) { wk = 0; ++wt; wretile(); }
  };

  // ---- X tile: nk chunks, this wave's 4 DMA instructions per chunk
  {
    const T *gp[LPW];
    const T *ga[GAT ? LPW : 1];
#pragma unroll
    for (int j = 0; j < LPW; ++j) {
      const int trow = RPI * (j * 4 + wave) + lane / PPR;
      const int piece = (lane % PPR) ^ ((trow >> SWS) & (PPR - 1));
      int grow = row0 + trow;
      grow = grow < a.rows ? grow : a.rows - 1;
      gp[j] = reinterpret_cast<const T *>(a.X) + (size_t)grow * a.x_ld + piece * 8;
      if (GAT) {
        const int smp = grow >> NPXL, pxl = grow & ((1 << NPXL) - 1);
        const int nb = a.gidx[(smp * 16 + (pxl >> a.g_klog2)) * 16 + (pxl & ((1 << a.g_klog2) - 1))];
        ga[j] = reinterpret_cast<const T *>(a.gfeat) + (size_t)(smp * 16 + nb) * a.g_ldf + piece * 8;
        gp[j] -= (size_t)a.g_nsplit * 32;
      }
    }
    for (int kc = 0; kc < nk; ++kc) {
#pragma unroll
      for (int j = 0; j < LPW; ++j) {
        const T *src = (GAT && kc < a.g_nsplit) ? ga[GAT ? j : 0] : gp[j];
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)(src + kc * 32),
                                         (__attribute__((address_space(3))) void *)(smem_raw + (size_t)kc * CH_B + (j * 4 + wave) * 1024),
                                         16, 0, 0);
      }
    }
  }
  for (int g = 0; g < nstw - 1 && g < G; ++g) wissue();
  if (AFF) {  // per-sample scale / shift of the consumer-side GroupNorm, staged as fp16 beside the tables
    const int nb = a.rows >> NPXL, n_aff = NSAMP * a.k_pad;
    for (int i = tid; i < n_aff; i += 256) {
      const int sm = i / a.k_pad, k = i - sm * a.k_pad;
      int b = (row0 >> NPXL) + sm;
      b = b < nb ? b : nb - 1;
      aff_lds[(sm * 2 + 0) * a.k_pad + k] = (_Float16)a.in_scale[(size_t)b * a.in_bs + k];
      aff_lds[(sm * 2 + 1) * a.k_pad + k] = (_Float16)a.in_shift[(size_t)b * a.in_bs + k];
    }
  }
  stage_epilogue_tables<CBW>(a, tile_of(0) * CBW, tid, reinterpret_cast<uint32_t *>(tabs), reinterpret_cast<float *>(tabs + TAB_DW * 4));
  SLIDE_STAMP(a, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // X (and the first weight chunks) have landed
  __syncthreads();
  SLIDE_STAMP(a, 2);
  if (AFF) {  // x <- x * scale + shift, once, in place (the ring kernels redo this for every column tile)
    const int npieces = nk * TM * PPR;
    for (int i = tid; i < npieces; i += 256) {
      const int kc = i / (TM * PPR), rem = i - kc * (TM * PPR), trow = rem >> 2, slot = rem & 3;
      const int piece = slot ^ ((trow >> SWS) & (PPR - 1));
      const _Float16 *aw = aff_lds + (size_t)(trow >> (NPXL < 8 ? NPXL : 8)) * 2 * a.k_pad + kc * 32 + piece * 8;
      f16x8 *px = reinterpret_cast<f16x8 *>(smem_raw + (size_t)kc * CH_B + trow * ROWB + slot * 16);
      const f16x8 sc = *reinterpret_cast<const f16x8 *>(aw), sh = *reinterpret_cast<const f16x8 *>(aw + a.k_pad);
      *px = __builtin_elementwise_fma(*px, sc, sh);
    }
    __syncthreads();
  }
  int xrow[2], xkey[2], wrw[CBW], wkey[CBW];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const int trow = wave * 64 + rb * 32 + col;
    xrow[rb] = trow * ROWB; xkey[rb] = (trow >> SWS) & (PPR - 1);
  }
#pragma unroll
  for (int cb = 0; cb < CBW; ++cb) {
    const int trow = cb * 32 + col;
    wrw[cb] = trow * ROWB; wkey[cb] = (trow >> SWS) & (PPR - 1);
  }
  int cs = 0;  // stage of the chunk being consumed
  for (int tt = 0; tt < ntc; ++tt) {
    const int tc = tile_of(tt);
    f32x16 acc[CBW][2];
#pragma unroll
    for (int i = 0; i < CBW; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int kc = 0; kc < nk; ++kc) {
      const int g = tt * nk + kc;
      // chunk g must have landed.  Younger weight DMAs may stay in flight; across a tile boundary the epilogue's own
      // loads / stores sit in the same counter, so the first chunk of a tile drains it (the chunks issued before the
      // epilogue have long landed by then)
      const int younger = G - 1 - g < nstw - 2 ? G - 1 - g : nstw - 2;
      if (kc == 0 || younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPWW) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPWW) : "memory");
      __builtin_amdgcn_s_barrier();
      if (g + nstw - 1 < G) wissue();  // chunk g + nstw - 1 overwrites the stage consumed at g - 1
      const unsigned char *sx = smem_raw + (size_t)kc * CH_B;
      const unsigned char *sw = wring + (size_t)cs * WST_B;
      cs = cs + 1 == nstw ? 0 : cs + 1;
#pragma unroll
      for (int st2 = 0; st2 < 2; ++st2) {
        f16x8 af[CBW], bf[2];
        const int piece = st2 * 2 + half;
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb) af[cb] = *reinterpret_cast<const f16x8 *>(sw + wrw[cb] + ((piece ^ wkey[cb]) << 4));
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) bf[rb] = *reinterpret_cast<const f16x8 *>(sx + xrow[rb] + ((piece ^ xkey[rb]) << 4));
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
          for (int rb = 0; rb < 2; ++rb)
            acc[cb][rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[cb], bf[rb], acc[cb][rb], 0, 0, 0);
      }
    }
    if (tt == 0) SLIDE_STAMP(a, 6);
    if (tt == ntc - 1) SLIDE_STAMP(a, 13);
    // descriptor tables of the NEXT column tile into the other table set (visible after the next tile's barrier)
    unsigned char *const tb = tabs + (size_t)(tt & 1) * TAB_B;
    if (tt + 1 < ntc) {
      unsigned char *const tn = tabs + (size_t)((tt + 1) & 1) * TAB_B;
      stage_epilogue_tables<CBW>(a, tile_of(tt + 1) * CBW, tid, reinterpret_cast<uint32_t *>(tn), reinterpret_cast<float *>(tn + TAB_DW * 4));
    }
    __syncthreads();  // every wave is past the previous tile's epilogue: `red` and the other table set are free
    gemm_epilogue<SLIDE_PREC_F16, NPXL, CBW>(a, acc, row0, tc * CBW, wave, half, col, reinterpret_cast<const uint32_t *>(tb),
                                             reinterpret_cast<const float *>(tb + TAB_DW * 4), red);
    if (tt == 0) SLIDE_STAMP(a, 7);
    if (tt == ntc - 1) SLIDE_STAMP(a, 14);
  }
}

constexpr int SLIDE_MAX_DEVICES = 64;
inline int current_device_slot() {
  int d = 0;
  (void)hipGetDevice(&d);
  return d >= 0 && d < SLIDE_MAX_DEVICES ? d : 0;
}

template <int NPXL, int CBW, bool AFF, bool GAT>
int launch_gemm_xs(const GemmArgs &a, hipStream_t s) {
  constexpr int NSAMP = (1 << NPXL) >= TM ? 1 : TM >> NPXL;
  constexpr int TAB_DW = CBW * EPI_DW + (CBW * EPI_DW) % 4;
  const size_t fixed = (size_t)(a.k_pad / 32) * TM * 64 + 2 * (size_t)(TAB_DW + CBW * 96) * 4 +
                       (AFF ? (size_t)NSAMP * 2 * a.k_pad * 2 : 0) + (size_t)(256 * CBW + 64 * CBW * (NSAMP > 4 ? NSAMP : 4)) * 4 + 64;
  // weight ring: as many 32-deep stages as the LDS leaves (2 .. 8; 3 keep two chunks in flight behind the one consumed)
  int nstw = (int)((160 * 1024 - (long)fixed) / (32 * CBW * 64));
  nstw = nstw > 8 ? 8 : nstw;
  if (nstw < 3) return -8;
  const size_t shm = fixed + (size_t)nstw * 32 * CBW * 64;
  const int ntr = (a.rows + TM - 1) / TM;
  GemmArgs b = a;
  b.sched = nullptr;
  b.shm_bytes = (int)shm;
  static bool attr_done[SLIDE_MAX_DEVICES] = {};
  bool &attr_set = attr_done[current_device_slot()];
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_xs_kernel<NPXL, CBW, AFF, GAT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_xs_kernel<NPXL, CBW, AFF, GAT>), dim3(ntr), dim3(256), shm, s, b, nstw);
  return (int)hipGetLastError();
}

}  // namespace

int slide_launch_gemm_xs(const GemmArgs &a, int npxl, int cbw, bool aff, bool gat, const void *wfrag, hipStream_t s) {
#define XCASE(L, A, G) if (npxl == L && aff == A && gat == G) return cbw == 4 ? launch_gemm_xs<L, 4, A, G>(a, s) : launch_gemm_xs<L, 2, A, G>(a, s)
  XCASE(8, false, false); XCASE(8, true, false); XCASE(8, false, true);
  XCASE(7, false, false); XCASE(7, true, false); XCASE(7, false, true);
#undef XCASE
  return -4;
}
