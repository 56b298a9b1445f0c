// gemm_small.h -- the split-K small-launch GEMM body (16 rows per sample), shared by engine.hip (gemm_small_kernel,
// pair_first_kernel) and gemm_gx.hip (sa_chain_p_kernel: the per-point query GEMM of an SA block riding on the launch of its
// fused Mlp chain).  Anonymous namespace: every translation unit gets its own copy.
#pragma once
#include "gemm_common.h"

namespace {

// Small launches (the 16-row per-point GEMMs: rows = 16 x batch, so a 256-row tiling yields a handful of workgroups
// whose cost is their own serial latency).  Tile 64 rows x 64 channels; the four waves are (K half, channel block): wave
// (kh, cbw) takes the 32-deep chunks kh, kh+2, ... of channel block cbw through a private LDS-DMA ring (no workgroup barrier
// in the loop) -- then the two K halves of a block meet through LDS and every wave finishes ONE 32x32 output block (channel
// block cbw, row block kh) through the common epilogue: the K loop and the epilogue are each ~4x shorter per wave than
// on a 256-row tile and the grid is 4x larger.  48 KB of rings: three workgroups per CU.
// (168-VGPR budget: a wave of this kernel then shares a SIMD with two waves of the 64-channel GEMM tiles of the other chains)
// PAIR (pair_first_kernel below): 0 = none, 1 / 2 = the channel blocks from pa.pair_cob0 on are the per-point products of an
// SA / FP block's pair decomposition and leave through the pair-table epilogue instead of the common one
template <int NST, bool AFF, int PAIR>
__device__ __forceinline__ void small_body(const GemmArgs &a, const PairArgs &pa, const int bid) {
  using T = _Float16;
  constexpr int NPXL = 4;
  // (round 3: waves = (K half kh, channel block cbw) instead of four K quarters over both blocks: a stage is 64 X rows + the
  //  wave's 32 W rows = 6 KB, the rings 48 KB instead of 64 and the exchange 16 KB -- THREE workgroups fit a CU's LDS)
  constexpr int STAGE_B = 96 * 64;  // 64 X rows + 32 W rows, 64 bytes each
  constexpr int NJ = 6;             // LDS-DMA instructions per stage (16 rows each)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  SLIDE_STAMP(a, 0);
  const int ntc = (a.n_cob + 1) / 2;
  const int tc = bid % ntc, tr = bid / ntc;
  const int row0 = tr * 64, cob0 = tc * 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, col = lane & 31;
  const int kh = wave >> 1, cbw = wave & 1;
  unsigned char *const ring = smem_raw + (size_t)wave * NST * STAGE_B;
  uint32_t *const epi_lds = reinterpret_cast<uint32_t *>(smem_raw + (size_t)4 * NST * STAGE_B);
  float *const vec_lds = reinterpret_cast<float *>(epi_lds + 2 * EPI_DW + (2 * EPI_DW) % 4);
  const T *gp[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int trow = 16 * j + (lane >> 2);
    const int piece = (lane & 3) ^ ((trow >> 2) & 3);
    if (trow < 64) {
      int grow = row0 + trow;
      grow = grow < a.rows ? grow : a.rows - 1;
      gp[j] = reinterpret_cast<const T *>(a.X) + (size_t)grow * a.x_ld + piece * 8;
    } else {
      int gco = (cob0 + cbw) * 32 + (trow - 64);
      gco = gco < a.n_cob * 32 ? gco : a.n_cob * 32 - 1;
      gp[j] = reinterpret_cast<const T *>(a.W) + (size_t)gco * a.k_pad + piece * 8;
    }
  }
  auto issue = [&](int kc, int st) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)(gp[j] + kc * 32),
                                       (__attribute__((address_space(3))) void *)(ring + (size_t)st * STAGE_B + j * 1024),
                                       16, 0, 0);
  };
  f32x16 acc[2];  // row blocks 0 / 1 of channel block cbw, this wave's K half
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  int xrow[2], xkey[2];
  const int wrow = (64 + col) * 64, wkey = ((64 + col) >> 2) & 3;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int tx = i * 32 + col;
    xrow[i] = tx * 64; xkey[i] = (tx >> 2) & 3;
  }
  const int nk = a.k_pad / 32;
  const int mine = nk > kh ? (nk - kh + 1) / 2 : 0;  // chunks kh, kh + 2, ...
#pragma unroll
  for (int s0 = 0; s0 < NST - 1; ++s0)
    if (s0 < mine) issue(kh + 2 * s0, s0);
  // tables (and the affine vectors) are staged behind the primed rings: their latency overlaps the first chunks'
  stage_epilogue_tables<2>(a, cob0, tid, epi_lds, vec_lds);
  // PAIR: the four samples' coordinates [4][48]; (FP) the neighbour / squared-distance / weight slots [4][16 x 8] each: pair_ext
  float *const pair_lds = vec_lds + 2 * 96;
  float *const pair_ext = reinterpret_cast<float *>(smem_raw + 20480);  // [nbr | d2 | w][4 samples][16 x 8]: behind `part` / `tr` (18 KB)
  int ext_n[2] = {0, 0};
  float ext_d[2] = {0.f, 0.f}, ext_w[2] = {0.f, 0.f};
  if constexpr (PAIR != 0) {
    const int nb = a.rows >> NPXL, b0 = row0 >> NPXL;
    if (tid < 192) {
      const int bb = b0 + tid / 48 < nb ? b0 + tid / 48 : nb - 1;
      pair_lds[tid] = pa.xyz[(size_t)bb * 48 + tid % 48];
    }
    if (PAIR == 2) {
      // (round 6) the slot tables wait in SIX registers across the K loop and go to the DEAD ring area behind it (pair_ext below):
      // staged next to the coordinates they made the workgroup 57 KB -- two per CU; 51 KB fit three
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int i = tid + 256 * u;
        const int bb = b0 + (i >> 7) < nb ? b0 + (i >> 7) : nb - 1, t = i & 127;
        const int slot = (bb * 16 + (t >> 3)) * 16 + (t & 7);
        ext_n[u] = pa.nbr[slot]; ext_d[u] = pa.d2t[slot]; ext_w[u] = pa.wt[slot];
      }
    }
  }
  // AFF: consumer-side GroupNorm affine of the four samples of this tile, fp16 [sample][scale | shift][k_pad]
  _Float16 *const aff_lds = reinterpret_cast<_Float16 *>(vec_lds + 2 * 96);
  if (AFF) {
    const int nb = a.rows >> NPXL;
    if (a.gn_fin) {
      // finalise the statistics of this tile's four samples (finalize_gn_kernel's arithmetic), keep the own channel
      // slice for the fragments, publish the full rows for the later consumers (column tile 0 only)
      const SlideGnFin f = *a.gn_fin;
      const int off = (int)(a.in_scale - f.scale);
      float *const grp = reinterpret_cast<float *>(aff_lds + (size_t)4 * 2 * a.k_pad);  // [4 samples][32 groups][mean, rstd]
      if (tid < 128) {  // one (sample, group) per thread
        const int sm = tid >> 5, g = tid & 31;
        int b = (row0 >> NPXL) + sm;
        b = b < nb ? b : nb - 1;
        float mean = 0.f, rstd = 0.f;
        if (g < f.G) {
          float S = 0.f, SS = 0.f;
          // sixteen channels per trip as four 16-byte loads per array (dword-aligned: a group starts at any channel), all
          // issued together: the 128 threads' scattered 4-byte loads made a trip cost ~1.2 us (request rate), and a 24-channel
          // group three trips
          typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
          const int c_end = f.gend[g];
          const float *ps = f.sum + (size_t)b * f.bs, *pq = f.sq + (size_t)b * f.bs;
          for (int cc = f.gstart[g]; cc < c_end; cc += 16) {
            f4u s4[4], q4[4];
            int base[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              base[u] = cc + 4 * u + 4 <= f.bs ? cc + 4 * u : f.bs - 4;  // (stay inside the row; elements are masked below)
              s4[u] = *reinterpret_cast<const f4u *>(ps + base[u]);
              q4[u] = *reinterpret_cast<const f4u *>(pq + base[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int c = base[u] + e;
                if (c >= cc + 4 * u && c < c_end) { S += s4[u][e]; SS += q4[u][e]; }
              }
          }
          mean = S * f.inv_count;
          const float var = fmaxf(SS * f.inv_count - mean * mean, 0.f);
          rstd = 1.0f / sqrtf(var + GN_EPS);
        }
        grp[tid * 2] = mean;
        grp[tid * 2 + 1] = rstd;
      }
      __syncthreads();
      // column tile 0 publishes all C channels; the other tiles only need their own K slice [off, off + k_pad)
      const int c_lo = tc == 0 ? 0 : off, c_n = tc == 0 ? f.C : (a.k_pad < f.C - off ? a.k_pad : f.C - off);
      // four channels per thread and trip: their table loads are issued together
      for (int i0 = tid; i0 < 4 * c_n; i0 += 1024) {
        int gq[4];
        float gm[4], bt[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + 256 * u < 4 * c_n ? i0 + 256 * u : i0;
          const int c = c_lo + i % c_n;
          gq[u] = f.gid[c]; gm[u] = f.gamma[c]; bt[u] = f.beta[c];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + 256 * u;
          if (i >= 4 * c_n) break;
          const int sm = i / c_n, c = c_lo + (i - sm * c_n);
          int b = (row0 >> NPXL) + sm;
          b = b < nb ? b : nb - 1;
          const int g = gq[u];
          float sc = 1.f, sh = 0.f;
          if (g >= 0) {
            sc = gm[u] * grp[(sm * 32 + g) * 2 + 1];
            sh = bt[u] - grp[(sm * 32 + g) * 2] * sc;
          }
          if (tc == 0) {
            f.scale[(size_t)b * f.bs + c] = sc;
            f.shift[(size_t)b * f.bs + c] = sh;
          }
          const int kk = c - off;
          if (kk >= 0 && kk < a.k_pad) {
            aff_lds[(sm * 2 + 0) * a.k_pad + kk] = (_Float16)sc;
            aff_lds[(sm * 2 + 1) * a.k_pad + kk] = (_Float16)sh;
          }
        }
      }
    } else {
      for (int i = tid; i < 4 * a.k_pad; i += 256) {
        const int sm = i / a.k_pad, k = i - sm * a.k_pad;
        int b = (row0 >> NPXL) + sm;
        b = b < nb ? b : nb - 1;
        aff_lds[(sm * 2 + 0) * a.k_pad + k] = (_Float16)a.in_scale[(size_t)b * a.in_bs + k];
        aff_lds[(sm * 2 + 1) * a.k_pad + k] = (_Float16)a.in_shift[(size_t)b * a.in_bs + k];
      }
    }
    __syncthreads();
  }

  SLIDE_STAMP(a, 1);
  for (int i = 0; i < mine; ++i) {
    // this wave's own DMA: a counted wait orders it for this wave's reads, no barrier involved
    if (i + NST - 2 < mine) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * NJ) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned char *sb = ring + (size_t)(i % NST) * STAGE_B;
    f16x8 af[2], bf[2][2];
#pragma unroll
    for (int st2 = 0; st2 < 2; ++st2) {
      const int piece = st2 * 2 + half;
      af[st2] = *reinterpret_cast<const f16x8 *>(sb + wrow + ((piece ^ wkey) << 4));
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) bf[st2][rb] = *reinterpret_cast<const f16x8 *>(sb + xrow[rb] + ((piece ^ xkey[rb]) << 4));
      if (AFF) {
        const int kc = kh + 2 * i;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          const _Float16 *ap = aff_lds + (size_t)((rb * 2 + (col >> 4)) * 2) * a.k_pad + kc * 32 + piece * 8;
          bf[st2][rb] = __builtin_elementwise_fma(bf[st2][rb], *reinterpret_cast<const f16x8 *>(ap),
                                                  *reinterpret_cast<const f16x8 *>(ap + a.k_pad));
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the stage is free again before it is re-armed below
    if (i + NST - 1 < mine) issue(kh + 2 * (i + NST - 1), (i + NST - 1) % NST);
#pragma unroll
    for (int st2 = 0; st2 < 2; ++st2)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[st2], bf[st2][rb], acc[rb], 0, 0, 0);
  }
  __syncthreads();  // rings are dead, tables are visible
  SLIDE_STAMP(a, 2);
  if constexpr (PAIR == 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = tid + 256 * u;
      reinterpret_cast<int *>(pair_ext)[i] = ext_n[u];
      pair_ext[512 + i] = ext_d[u];
      pair_ext[1024 + i] = ext_w[u];
    }
  }
  // wave (kh, cbw) finishes the block (channel block cbw, row block kh): it keeps its own partial of that block in registers
  // and takes the other K half's from its partner (1 - kh, cbw) through LDS [wave][reg][lane]
  float *const part = reinterpret_cast<float *>(smem_raw);
#pragma unroll
  for (int r = 0; r < 16; ++r) part[((wave * 16 + r) << 6) + lane] = kh ? acc[0][r] : acc[1][r];  // the block it does NOT own
  __syncthreads();
  f32x16 one[1][1];
  {
    const int pw = ((1 - kh) << 1) | cbw;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float own = kh ? acc[1][r] : acc[0][r], oth = part[((pw * 16 + r) << 6) + lane];
      one[0][0][r] = kh ? oth + own : own + oth;  // K half 0 first, whichever wave adds
    }
  }
  SLIDE_STAMP(a, 3);
  SLIDE_STAMP(a, 4);
  const int cb = cbw, rb = kh;
  if (PAIR == 0 || cob0 + cb < pa.pair_cob0)
    gemm_epilogue<SLIDE_PREC_F16, NPXL, 1, 1>(a, one, row0 + rb * 32, cob0 + cb, 0, half, col, epi_lds + cb * EPI_DW,
                                              vec_lds + cb * 96, nullptr);
  if constexpr (PAIR != 0) {
    // ---- pair-table epilogue (the arithmetic of pair_norm_kernel, gemm_gx.hip, on the accumulators instead of a stored y).
    // The wave's block D[channel][row] is transposed through LDS so that a lane owns ONE channel of ONE of the block's two
    // samples and its registers run over the sample's 16 points.
    __syncthreads();  // every wave has summed its block: the partial-sum area is free
    const int cobi = cob0 + cb;
    if (cobi >= pa.pair_cob0 && cobi < a.n_cob) {
      float *const tr = reinterpret_cast<float *>(smem_raw) + wave * 1152;  // [32 channels][33] transposed block, then [16][65] columns
#pragma unroll
      for (int r = 0; r < 16; ++r) tr[((r & 3) + 8 * (r >> 2) + 4 * half) * 33 + col] = one[0][0][r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int smp = half, cl = col;                    // lane = (sample of the block, channel)
      const int sl4 = rb * 2 + smp;                      // sample slot of the tile (0..3)
      const int nb = a.rows >> NPXL;
      const int b = (row0 >> NPXL) + sl4;
      const int c = (cobi - pa.pair_cob0) * 32 + cl;     // pair channel
      const int ld = pa.ld;
      const uint32_t *ed = epi_lds + cb * EPI_DW;
      const int e_mode = (int)ed[0], e_flags = (int)ed[1], e_gs = (int)ed[2], e_n_norm = (int)ed[3];
      const float e_inv_count = __uint_as_float(ed[4]), e_stats_scale = __uint_as_float(ed[5]);
      const int e_stats_bs = (int)ed[9];
      float *const e_sum = reinterpret_cast<float *>((uint64_t)ed[30] | ((uint64_t)ed[31] << 32));
      float *const e_sq = reinterpret_cast<float *>((uint64_t)ed[32] | ((uint64_t)ed[33] << 32));
      const float *sxs = pair_lds + sl4 * 48;
      const float bias = vec_lds[cb * 96 + cl];
      const float4 ca = *reinterpret_cast<const float4 *>(pa.wa + (size_t)c * 4), cbv = *reinterpret_cast<const float4 *>(pa.wb + (size_t)c * 4);
      float av[16], bv[16];
#pragma unroll
      for (int p = 0; p < 16; ++p) {
        const float x0 = sxs[p * 3], x1 = sxs[p * 3 + 1], x2 = sxs[p * 3 + 2];
        av[p] = (tr[cl * 33 + smp * 16 + p] + bias) + (ca.x * x0 + ca.y * x1 + ca.z * x2);
        bv[p] = cbv.x * x0 + cbv.y * x1 + cbv.z * x2;
      }
      float vd = 0.f, vw = 0.f;
      if (PAIR == 2) { vd = pa.vv_in[c]; vw = pa.vv_in[ld + c]; }
      float g = 1.f, sh = 0.f;
      if (e_mode != SLIDE_EPI_RAW) {
        const bool pre_relu = (e_flags & SLIDE_F_PRE_RELU) != 0;
        float s = 0.f, ss = 0.f;
        if (PAIR == 1) {
#pragma unroll
          for (int p = 0; p < 16; ++p)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
              float v = av[q] + bv[p];
              if (pre_relu) v = fmaxf(v, 0.f);
              s += v; ss = fmaf(v, v, ss);
            }
        } else {
          const int *sqs = reinterpret_cast<const int *>(pair_ext) + sl4 * 128;
          const float *sds = pair_ext + 512 + sl4 * 128, *sws = pair_ext + 1024 + sl4 * 128;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the transposed block has been read: its area takes the columns
          float *const sa = tr;  // [16][65]
#pragma unroll
          for (int p = 0; p < 16; ++p) sa[p * 65 + lane] = av[p];  // (a lane reads back only its own column)
#pragma unroll
          for (int p = 0; p < 16; ++p)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const int sl = p * 8 + j;
              float v = sa[sqs[sl] * 65 + lane] + bv[p] + sds[sl] * vd + sws[sl] * vw;
              if (pre_relu) v = fmaxf(v, 0.f);
              s += v; ss = fmaf(v, v, ss);
            }
        }
        if (e_mode == SLIDE_EPI_STATS) {
          if (b < nb) {
            e_sum[(size_t)b * e_stats_bs + cl] = s * e_stats_scale;
            e_sq[(size_t)b * e_stats_bs + cl] = ss * e_stats_scale;
          }
        } else {  // NORM: groups of e_gs physical channels (a power of two <= 32: lanes of one half)
          for (int m = 1; m < e_gs; m <<= 1) {
            s += __shfl_xor(s, m, 64);
            ss += __shfl_xor(ss, m, 64);
          }
          const float mean = s * e_inv_count;
          const float var = fmaxf(ss * e_inv_count - mean * mean, 0.f);
          g = vec_lds[cb * 96 + 32 + cl] * __builtin_amdgcn_rsqf(var + GN_EPS);
          sh = vec_lds[cb * 96 + 64 + cl] - mean * g;
          if (cl >= e_n_norm) { g = 1.f; sh = 0.f; }  // MyGroupNorm leaves the last C % G channels as they are
        }
      }
      if (b < nb) {
#pragma unroll
        for (int p = 0; p < 16; ++p) {
          pa.ta[((size_t)b * 16 + p) * ld + c] = (_Float16)(av[p] * g + sh);
          pa.tb[((size_t)b * 16 + p) * ld + c] = (_Float16)(bv[p] * g);
        }
        if (PAIR == 2) {
          pa.vv_out[(size_t)b * 2 * ld + c] = vd * g;
          pa.vv_out[(size_t)b * 2 * ld + ld + c] = vw * g;
        }
      }
    }
  }
  SLIDE_STAMP(a, 5);
#ifdef SLIDE_TIMELINE
  if (a.dbg) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SLIDE_STAMP(a, 6);
  }
#endif
}


}  // namespace
