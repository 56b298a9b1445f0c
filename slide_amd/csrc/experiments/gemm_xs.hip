// gemm_xs.hip -- the X-stationary GEMM of the fused denoiser plan (dispatched by engine.hip's run_gemm).
#include "gemm_common.h"

#include <cstdlib>

namespace {

// ------------------------------------------------------------------------------------------------ X-stationary GEMM
// What the timelines of the ring kernels (gemm_glds_*) show at the model's K (64 .. 544): a 256 x 64 tile re-streams its X
// rows (256 x K) for every 64 output channels -- L2 -> LDS is the saturated resource of the K loops, and the queueing it
// causes (~2 us per DMA chunk or global load with three workgroups per CU streaming) is what every other phase of a tile
// then waits on.  This kernel removes the re-streaming:
//   * a workgroup owns a row tile (whole samples) and a SUBSET of the layer's column tiles (column split `nsplit`: the
//     workgroups of one row tile sit on one XCD and share its L2); its input stays in LDS for all of them;
//   * GATHERED first layers (the grouped input of an SA / FP block: row (point, neighbour) = [feature row of the neighbour |
//     coordinate channels]) keep only the 16-row POINT TABLE of the sample + the coordinate chunk: the B fragments of the
//     feature K steps are read from the neighbour's table row (per-lane LDS addresses are free) -- 24 KB instead of a
//     144 KB X tile at K = 288, so three such workgroups fit a CU;
//   * plain inputs are DMA-read once in the ring kernels' swizzled chunk image (16 KB per 32-deep chunk); the consumer-side
//     GroupNorm affine is applied once, in place;
//   * only the weights stream: a small LDS-DMA ring of [32 CBW rows][32 K] chunks, counted vmcnt + one barrier per chunk,
//     running ahead across column tiles;
//   * the common epilogue (gemm_common.h) runs per column tile on the same accumulator layout, descriptor tables double
//     buffered: outputs are bit-identical to the ring kernels'.
// OCC = workgroups per CU the register budget is compiled for (LDS decides at launch which variant applies).
template <int NPXL, int CBW, bool AFF, bool GAT, int OCC>
__global__ __launch_bounds__(256, OCC) void gemm_xs_kernel(GemmArgs a, int nstw, int nsplit, int ntr8) {
  using T = _Float16;
  constexpr int ROWB = 64, PPR = 4, RPI = 16, SWS = 2, LPW = 4;  // chunk image: [rows][4 pieces of 16 B]; X: 4 DMA / wave / chunk
  constexpr int CH_B = TM * ROWB;                                 // 16 KB of X per 32-deep chunk
  constexpr int TN = 32 * CBW, WST_B = TN * ROWB, LPWW = TN / 64; // weight stage: TN rows, LPWW DMA instructions / wave
  constexpr int NPX = 1 << NPXL;
  constexpr int NSAMP = NPX >= TM ? 1 : TM >> NPXL;
  constexpr int TB_B = NSAMP * 16 * ROWB;  // bytes of one 32-deep chunk of the point table(s) of the tile's samples
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, col = lane & 31;
  const int nk = a.k_pad / 32;
  const int ng = GAT ? a.g_nsplit : 0;  // leading chunks served by the point table
  const int nx = nk - ng;               // chunks held as X rows
  const int ntc_all = (a.n_cob + CBW - 1) / CBW;
  const int tr = blockIdx.x % ntr8, js = blockIdx.x / ntr8;  // row tile, column-split index
  if (tr * TM >= a.rows || js >= ntc_all) return;
  const int ntc = (ntc_all - js + nsplit - 1) / nsplit;      // this workgroup's column tiles: js, js + nsplit, ...
  const int row0 = tr * TM;
  constexpr int TAB_DW = CBW * EPI_DW + (CBW * EPI_DW) % 4;  // one set of descriptor tables: dwords, then CBW * 96 floats
  constexpr int TAB_B = (TAB_DW + CBW * 96) * 4;
  unsigned char *const xbase = smem_raw + (size_t)ng * TB_B;
  unsigned char *const wring = xbase + (size_t)nx * CH_B;
  unsigned char *const tabs = wring + (size_t)nstw * WST_B;
  _Float16 *const aff_lds = reinterpret_cast<_Float16 *>(tabs + 2 * TAB_B);  // [sample][scale | shift][k_pad]
  float *const red = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(aff_lds) + (AFF ? (size_t)NSAMP * 2 * a.k_pad * 2 : 0));
  SLIDE_STAMP(a, 0);

  // every workgroup walks its column tiles in its own rotation (all start together: spread the weight panels they ask for)
  const int rot = tr % ntc;
  auto tile_of = [&](int t) { int v = t + rot; v = v >= ntc ? v - ntc : v; return js + v * nsplit; };
  const int G = ntc * nk;
  int wtrow[LPWW], wpiece[LPWW];  // this lane's source row inside a TN-row weight panel and its (swizzled) 16-byte piece
#pragma unroll
  for (int j = 0; j < LPWW; ++j) {
    wtrow[j] = RPI * (j * 4 + wave) + lane / PPR;  // 0 .. TN-1
    wpiece[j] = (lane % PPR) ^ ((wtrow[j] >> SWS) & (PPR - 1));
  }
  // the next weight chunk to issue is tracked incrementally (tile step wt, chunk wk, stage ws): no division in the K loop
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

  // ---- resident input.  Point table (GAT): chunk c = [NSAMP * 16 table rows][64 B], one DMA instruction per 16 rows
  if (GAT) {
    const int n_ins = ng * NSAMP;  // 1-KB instructions, dealt round-robin to the waves
    const int smp0 = row0 >> NPXL, nsm = a.rows >> NPXL;
    for (int i = wave; i < n_ins; i += 4) {
      const int c = i / NSAMP, sl = i - c * NSAMP;
      const int trow = sl * 16 + lane / PPR;  // table row inside the tile
      const int piece = (lane % PPR) ^ ((trow >> SWS) & (PPR - 1));
      int smp = smp0 + sl;
      smp = smp < nsm ? smp : nsm - 1;
      const T *src = reinterpret_cast<const T *>(a.gfeat) + (size_t)(smp * 16 + (lane / PPR)) * a.g_ldf + c * 32 + piece * 8;
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)src,
                                       (__attribute__((address_space(3))) void *)(smem_raw + (size_t)c * TB_B + sl * 1024), 16, 0, 0);
    }
  }
  {  // X rows: nx chunks, this wave's 4 DMA instructions per chunk
    const T *gp[LPW];
#pragma unroll
    for (int j = 0; j < LPW; ++j) {
      const int trow = RPI * (j * 4 + wave) + lane / PPR;
      const int piece = (lane % PPR) ^ ((trow >> SWS) & (PPR - 1));
      int grow = row0 + trow;
      grow = grow < a.rows ? grow : a.rows - 1;
      gp[j] = reinterpret_cast<const T *>(a.X) + (size_t)grow * a.x_ld + piece * 8;
    }
    for (int kc = 0; kc < nx; ++kc) {
#pragma unroll
      for (int j = 0; j < LPW; ++j)
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)(gp[j] + kc * 32),
                                         (__attribute__((address_space(3))) void *)(xbase + (size_t)kc * CH_B + (j * 4 + wave) * 1024),
                                         16, 0, 0);
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
  // B-fragment addressing of this lane's two row blocks: own X rows, and (GAT) the neighbour's table row
  int xrow[2], xkey[2], trw[2], tkey[2], wrw[CBW], wkey[CBW];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const int trow = wave * 64 + rb * 32 + col;
    xrow[rb] = trow * ROWB; xkey[rb] = (trow >> SWS) & (PPR - 1);
    trw[rb] = 0; tkey[rb] = 0;
    if (GAT) {
      int grow = row0 + trow;
      grow = grow < a.rows ? grow : a.rows - 1;
      const int smp = grow >> NPXL, pxl = grow & (NPX - 1);
      const int nb = a.gidx[(smp * 16 + (pxl >> a.g_klog2)) * 16 + (pxl & ((1 << a.g_klog2) - 1))];
      const int tt = (trow >> NPXL) * 16 + nb;  // table row inside the tile
      trw[rb] = tt * ROWB; tkey[rb] = (tt >> SWS) & (PPR - 1);
    }
  }
#pragma unroll
  for (int cb = 0; cb < CBW; ++cb) {
    const int trow = cb * 32 + col;
    wrw[cb] = trow * ROWB; wkey[cb] = (trow >> SWS) & (PPR - 1);
  }
  SLIDE_STAMP(a, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the resident input (and the first weight chunks) have landed
  __syncthreads();
  SLIDE_STAMP(a, 2);
  if (AFF) {  // x <- x * scale + shift, once, in place (the ring kernels redo this for every column tile)
    const int npieces = nx * TM * PPR;
    for (int i = tid; i < npieces; i += 256) {
      const int kc = i / (TM * PPR), rem = i - kc * (TM * PPR), trow = rem >> 2, slot = rem & 3;
      const int piece = slot ^ ((trow >> SWS) & (PPR - 1));
      const _Float16 *aw = aff_lds + (size_t)(trow >> (NPXL < 8 ? NPXL : 8)) * 2 * a.k_pad + (ng + kc) * 32 + piece * 8;
      f16x8 *px = reinterpret_cast<f16x8 *>(xbase + (size_t)kc * CH_B + trow * ROWB + slot * 16);
      const f16x8 sc = *reinterpret_cast<const f16x8 *>(aw), sh = *reinterpret_cast<const f16x8 *>(aw + a.k_pad);
      *px = __builtin_elementwise_fma(*px, sc, sh);
    }
    __syncthreads();
  }
  int cs = 0;  // stage of the weight chunk being consumed
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
      // weight chunk g must have landed.  Younger weight DMAs may stay in flight; across a tile boundary the epilogue's own
      // loads / stores sit in the same counter, so the first chunk of a tile drains it (the chunks issued before the
      // epilogue have long landed by then)
      const int younger = G - 1 - g < nstw - 2 ? G - 1 - g : nstw - 2;
      if (kc == 0 || younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPWW) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPWW) : "memory");
      __builtin_amdgcn_s_barrier();
      if (g + nstw - 1 < G) wissue();  // chunk g + nstw - 1 overwrites the stage consumed at g - 1
      const bool tab = GAT && kc < ng;
      const unsigned char *sx = tab ? smem_raw + (size_t)kc * TB_B : xbase + (size_t)(kc - ng) * CH_B;
      const unsigned char *sw = wring + (size_t)cs * WST_B;
      cs = cs + 1 == nstw ? 0 : cs + 1;
#pragma unroll
      for (int st2 = 0; st2 < 2; ++st2) {
        f16x8 af[CBW], bf[2];
        const int piece = st2 * 2 + half;
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb) af[cb] = *reinterpret_cast<const f16x8 *>(sw + wrw[cb] + ((piece ^ wkey[cb]) << 4));
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
          bf[rb] = *reinterpret_cast<const f16x8 *>(sx + (tab ? trw[rb] + ((piece ^ tkey[rb]) << 4) : xrow[rb] + ((piece ^ xkey[rb]) << 4)));
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

template <int NPXL, int CBW, bool AFF, bool GAT, int OCC>
int launch_xs(const GemmArgs &b, size_t shm, int nstw, int nsplit, int ntr8, hipStream_t s) {
  static bool attr_done[SLIDE_MAX_DEVICES] = {};
  bool &attr_set = attr_done[current_device_slot()];
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_xs_kernel<NPXL, CBW, AFF, GAT, OCC>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_xs_kernel<NPXL, CBW, AFF, GAT, OCC>), dim3(ntr8 * nsplit), dim3(256), shm, s, b, nstw, nsplit, ntr8);
  return (int)hipGetLastError();
}

// LDS footprint decides how many workgroups share a CU (1, 2 or 3), that decides the register budget (the OCC variant) and how
// many workgroups split a row tile's column tiles.  `want_occ` (0 = automatic) forces a lower occupancy (A/B timing).
template <int NPXL, bool AFF, bool GAT>
int launch_gemm_xs(const GemmArgs &a, int cbw_req, int want_occ, hipStream_t s) {
  constexpr int NSAMP = (1 << NPXL) >= TM ? 1 : TM >> NPXL;
  const int nk = a.k_pad / 32, ng = GAT ? a.g_nsplit : 0;
  auto fixed_of = [&](int cbw) {
    const int tab_dw = cbw * EPI_DW + (cbw * EPI_DW) % 4;
    return (size_t)ng * NSAMP * 16 * 64 + (size_t)(nk - ng) * TM * 64 + 2 * (size_t)(tab_dw + cbw * 96) * 4 +
           (AFF ? (size_t)NSAMP * 2 * a.k_pad * 2 : 0) + (size_t)(256 * cbw + 64 * cbw * (NSAMP > 4 ? NSAMP : 4)) * 4 + 64;
  };
  const int ntr = (a.rows + TM - 1) / TM, ntr8 = (ntr + 7) / 8 * 8;
  // occupancy: the most workgroups per CU whose LDS (>= 3 ring stages each) fits; 64-channel tiles for 2 / 3 per CU
  int occ = 1, cbw = 2, nstw = 0;
  for (int o = 3; o >= 1; --o) {
    if (want_occ && o > want_occ) continue;
    const int c = (o == 1 && cbw_req == 4) ? 4 : 2;
    const long room = (long)(160 * 1024 / o) - (long)fixed_of(c) - (o > 1 ? 256 : 0);
    const int st = (int)(room / (32 * c * 64));
    if (st >= 3) { occ = o; cbw = c; nstw = st > 8 ? 8 : st; break; }
  }
  if (nstw < 3) return -8;
  const int ntc = (a.n_cob + cbw - 1) / cbw;
  // column split: enough workgroups to fill occ x 256 slots, never more than the column tiles
  int nsplit = (occ * 256 + ntr - 1) / ntr;
  static const int split_cap = [] { const char *e = getenv("SLIDE_XS_NSPLIT"); return e ? atoi(e) : 0; }();  // experiment knob
  if (split_cap > 0 && nsplit > split_cap) nsplit = split_cap;
  nsplit = nsplit > ntc ? ntc : nsplit;
  nsplit = nsplit < 1 ? 1 : nsplit;
  GemmArgs b = a;
  b.sched = nullptr;
  const size_t shm = fixed_of(cbw) + (size_t)nstw * 32 * cbw * 64;
  b.shm_bytes = (int)shm;
  if (occ == 3) return launch_xs<NPXL, 2, AFF, GAT, 3>(b, shm, nstw, nsplit, ntr8, s);
  if (occ == 2) return launch_xs<NPXL, 2, AFF, GAT, 2>(b, shm, nstw, nsplit, ntr8, s);
  if (cbw == 4) return launch_xs<NPXL, 4, AFF, GAT, 1>(b, shm, nstw, nsplit, ntr8, s);
  return launch_xs<NPXL, 2, AFF, GAT, 1>(b, shm, nstw, nsplit, ntr8, s);
}

}  // namespace

int slide_launch_gemm_xs(const GemmArgs &a, int npxl, int cbw, bool aff, bool gat, int want_occ, hipStream_t s) {
#define XCASE(L, A, G) if (npxl == L && aff == A && gat == G) return launch_gemm_xs<L, A, G>(a, cbw, want_occ, s)
  XCASE(8, false, false); XCASE(8, true, false); XCASE(8, false, true);
  XCASE(7, false, false); XCASE(7, true, false); XCASE(7, false, true);
#undef XCASE
  return -4;
}
