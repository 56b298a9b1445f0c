/*
 * slide_engine.h -- C-ABI of the fused latent-DDPM denoiser engine in libslide_hip.so.
 *
 * What it replaces: one reverse-diffusion step of the reference =
 *   PointNet2CloudCondition.forward (pointnet2/models/pointnet2_with_pcld_condition.py:286-489; ~80 torch
 *   module launches: Mlp_plus_t_emb pointnet2_ops/pointnet2_modules.py:119-176, AttentionModule
 *   pointnet2_ops/attention.py:70-96, QueryAndGroup / group_knn pointnet2_ops/pointnet2_utils.py:368-524)
 *   + the DDPM update (pointnet2/util.py:247-253, pointnet2/diffusion_utils/diffusion.py:58-95).
 * Here a step is a short list of `SlideOp` launches (a "plan") that the host builds once and replays,
 * eagerly or from a captured hipGraph.  All activations are channel-minor fp32 matrices
 * [B*npx][ld] resident in HBM; 1x1 convolutions are MFMA contractions with GroupNorm / ReLU /
 * t-embedding / residual fused into the epilogue.
 *
 * All pointers are DEVICE pointers unless stated.  Status: 0 = ok, else hipError_t (or <0 for bad args).
 */
#ifndef SLIDE_ENGINE_H
#define SLIDE_ENGINE_H

#include <stdint.h>

#include "slide_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* epilogue modes of one 32-channel output block */
enum { SLIDE_EPI_RAW = 0, SLIDE_EPI_NORM = 1, SLIDE_EPI_STATS = 2 };
enum { SLIDE_F_PRE_RELU = 1, SLIDE_F_POST_RELU = 2, SLIDE_F_OUT_F32 = 4,
       /* PAIR residual (fp16 plans): the residual of row (sample, point p, slot j) is NOT a stored row but the sum of two
        * per-point table rows, residual[sample*16 + q] + res_b[sample*16 + p] -- the first layer of an SA / FP block is linear
        * in [neighbour features | coordinates], so its res_connect output separates into a neighbour term and a centre term.
        * RES_PAIR: 16 x 16-row samples in NATURAL neighbour order (q = j; every point is a neighbour of every point).
        * RES_PAIR_NBR: 16 x 8-row samples, q = nbr[(sample*16 + p)*16 + j] (the GEMM op's neighbour table) and the two
        * per-row scalars of group_knn enter as + d2 * res_vd[c] + w * res_vw[c]. */
       SLIDE_F_RES_PAIR = 8, SLIDE_F_RES_PAIR_NBR = 16,
       /* FRAGMENT-MAJOR output (round 6; fp16 storage, 128- / 256-row samples): like chunk-major the block's 32 channels of
        * the rows are one [rows][32] slab (out = slab base, out_ld == 32), but inside every 32-row group the 2 KB are ordered
        * as the two MFMA operand fragments a consumer wave loads for that group: element (row, c) at
        * (row & ~31)*32 + (c >> 4)*512 + ((c >> 3) & 1)*256 + (row & 31)*8 + (c & 7)  halves -- [k16 step][k half][row][8].
        * A wave's global_load_dwordx4 of one fragment then reads 1 KB of CONSECUTIVE memory (chunk-major: 32 B of each of 32
        * rows 64 B apart -- the request-bound pattern: 28 vs 44-49 B/clk/CU L2 -> VGPR, tools/lds_fill.hip XP / XF), and one
        * epilogue store instruction writes 1 KB of it.  Consumer: SLIDE_OP_ATTN_TAIL with f[1] bit 4 (attn_tail_rx_kernel). */
       SLIDE_F_OUT_FM = 32 };
/* MFMA precision of a GEMM: exact fp32 (v_mfma_f32_32x32x2_f32) or fp16 inputs / fp32 accumulate
 * (v_mfma_f32_32x32x16_f16) */
/* SLIDE_PREC_SPLIT (round 4): fp32 STORAGE (the fp32 plan: same ops, buffers and epilogues as SLIDE_PREC_F32), the contractions
 * on the fp16 matrix pipe with BOTH operands as two-term fp16 splits x = hi + 2^-11 lo (hi = fp16(x), lo = fp16(2^11 (x - hi))),
 * three products hi*hi + 2^-11 (hi*lo + lo*hi) accumulated in fp32: ~2^-22 relative per product (fp32-grade; tools/prec_emul.py:
 * 1e-6 on a forward where fp16 operands give 1.4e-3) at 3/16 of the fp32 MFMA's matrix-pipe time. */
enum { SLIDE_PREC_F32 = 0, SLIDE_PREC_F16 = 1, SLIDE_PREC_SPLIT = 2 };

/* One per 32 output channels of a GEMM; an array of these lives in DEVICE memory.
 *   y = acc + bias + pre_add[row >> pre_add_shift];  PRE_RELU;
 *   [NORM: GroupNorm over (group of `gs` physical channels x the npx rows of one sample) for the first n_norm
 *    channels of the block];  POST_RELU;  + addvec[b] (row addvec_idx[0] of a table when addvec_idx != NULL);
 *   + residual[row];  stored to out (activation storage type, or fp32 with SLIDE_F_OUT_F32).
 * STATS instead emits per-sample channel sums / sums of squares (x stats_scale) for a later FINALIZE_GN and stores
 * the un-normalised value.  Activation pointers (pre_add, residual, out) are fp32 or fp16 according to the GEMM's
 * precision; bias / gamma / beta / addvec / stats are always fp32. */
/* PACKED VECTORS (round 6).  A descriptor array passed to a GEMM-type op (p[2]; p[13] of a chained split layer) may be FOLLOWED in the
 * same device buffer by the blocks' vectors as values, [n_cob][bias 32 | gamma 32 | beta 32] fp32 (zeros where a pointer is NULL);
 * the caller says so by setting BIT 0 of the pointer it passes (the array itself is at least 8-byte aligned).  The kernels then stage
 * descriptors and vectors by LDS-DMA straight from that buffer -- without it every workgroup reads the three pointers of each block
 * and then the vectors behind them: two dependent memory round trips in its prologue.  bias / gamma / beta below stay valid
 * pointers either way (the pair-table pass reads gamma / beta from them). */
#define SLIDE_EPI_PACKED_VECS ((uintptr_t)1)
typedef struct SlideEpi {
  int32_t mode, flags, gs, n_norm;
  float inv_count, stats_scale;
  int32_t out_ld, res_ld;
  int32_t addvec_bs, stats_bs, pre_add_ld, pre_add_shift;
  int32_t addvec_idx_stride, pad0;
  const float *bias;          /* [32] or NULL */
  const float *gamma;         /* [32] (NORM) */
  const float *beta;          /* [32] (NORM) */
  const float *addvec;        /* addvec[idx*addvec_idx_stride + b*addvec_bs + c] or NULL */
  const int32_t *addvec_idx;  /* device scalar (the current timestep) or NULL */
  const void *residual;       /* residual[row*res_ld + c] or NULL */
  const void *pre_add;        /* pre_add[(row >> pre_add_shift)*pre_add_ld + c] or NULL.  pre_add_shift = -k (k = 3, 4): the row of
                               * the NEIGHBOUR point instead, sample*16 + nbr[(sample*16 + point)*16 + j] for row (sample, point, j)
                               * of a 16 x 2^k-row sample, nbr = the GEMM op's p[9] (per-point partial products of a block's first
                               * layer, gathered per neighbour) */
  void *out;                  /* out[row*out_ld + c] */
  float *stats_sum;           /* stats_sum[b*stats_bs + c] (STATS) */
  float *stats_sq;
  const void *res_b;          /* pair residual: centre-term table [B*16][res_ld] (activation type) */
  const float *res_vd;        /* RES_PAIR_NBR: [32] coefficients of the squared distance ... */
  const float *res_vw;        /* ... and of the interpolation weight for this block's channels */
} SlideEpi;

enum {
  SLIDE_OP_GEMM = 1,        /* ([12] / [13]: squared-distance / interpolation-weight tables fp32 [B*16][16] of a SLIDE_F_RES_PAIR_NBR residual, with [9] the neighbour table)  p: X, W, epi, in_scale, in_shift, [5] timeline buffer (instrumented builds only, else NULL), [6] SlideGnFin* (16-row launches with input affine: finalise the statistics in this launch), [7] 9 zeroed ints for the optional persistent tile scheduler (NULL = one tile per workgroup), [8] point-feature table + [9] neighbour table of the GATHER mode (first GEMM of an SA / FP block: the first f[1] 32-column chunks of X row (b, p, k) are read from row b*16 + idx[(b*16+p)*16+k] of the table with row length f[2], neighbours per point 2^f[3]; p[0] / x_ld then describe only the remaining columns; with p[8] NULL, p[9] is the neighbour table of gathered pre_add terms, SlideEpi.pre_add_shift < 0); [11] + f[1..3] with p[3] / p[4] set and no gather: DEFERRED NORMALISATION of the module-level path -- x' = relu?(x * scale + shift) + add applied to the X fragments, p[11] = add vectors [sample][f[2]] (or NULL), f[1] = 256-row tiles per sample, f[3] = 2 * (channels of add) + (ReLU ? 1 : 0); [10] non-NULL selects the X-stationary kernel for sample-wide fp16 layers whose 256-row X tile fits the LDS (one workgroup per row tile keeps X resident and computes every column tile, the weights stream through a small LDS-DMA ring; i[9] == 5 keeps the ring kernels); f[0]: start stagger in us for the persistent mode; i[9]: 0 = default ring, 1 = 64-deep chunks, 2 = eight-wave 256x256 tiles   i: rows, x_ld, k_pad, n_cob, npx_log2, in_bs, prec, cbw(2|4), lds_dma (bit 0: fp16 LDS-DMA ring kernels; bit 1: W is CHUNK-MAJOR [k_pad/32][n_cob*32][32] -- ring kernels of the 128- / 256-row samples only; bit 2: a block carries a PAIR residual).  CHUNK-MAJOR X: with x_ld == 32 and k_pad > 32 the ring kernels read X as [k/32][rows][32] (chunk kc of row r at X + (kc*rows + r)*32); outputs / residuals use the same layout through SlideEpi's per-block pointer with out_ld / res_ld == 32 */
  SLIDE_OP_PREP_POINTS = 2, /* p: x, xyz, feat0, knn_idx, knn_d2, [5] optional second copy of feat0, chunk-major [c/32][B*16][32], [6] SlidePrepCopy[i[4]] (device), [7] optional knn_w [B*16][16]: group_knn's interpolation weights of the 8 nearest (pointnet2_utils.py:510-513), 0 beyond   i: B, cx, ldf, prec, n_copies     (16 points / sample) */
  SLIDE_OP_ASSEMBLE_SA = 3, /* p: xyz, feat, knn_idx, g            i: B, C, ldf, ldg, K, prec, c_begin (0 = all columns; else only columns >= c_begin), ld_out */
  SLIDE_OP_ASSEMBLE_FP = 4, /* p: xyz, feat, knn_idx, knn_d2, g    i: B, C, ldf, ldg, K, prec, c_begin, ld_out */
  SLIDE_OP_FINALIZE_GN = 5, /* p: sum, sq, gid, gstart, gend, gamma, beta, scale, shift  i: B, C, bs   f: inv_count */
  SLIDE_OP_ATTN_COMBINE = 6,/* p: S, V, out   i: B*np, C, ldS, ldV, ldo, K, prec */
  SLIDE_OP_COPY_COLS = 7,   /* p: src, dst    i: rows, n, src_ld, dst_ld, src_is_f16, dst_is_f16 */
  SLIDE_OP_TEMB = 8,        /* p: ts(or NULL), t_dev, w1,b1,w2,b2, wfc, bfc, out, freq   i: nsamp, t_dim, n_out  (weights [in][out]) */
  SLIDE_OP_COND = 9,        /* p: label(int64), class_emb, wfc, bfc, out           i: B, dim, n_out */
  SLIDE_OP_UPDATE_POS = 10, /* p: x, eps, noise(or NULL), t_dev, c_eps, sqrt_alpha, sigma  i: n_elem, eps_ld (0 = compact rows of 3), seed_lo, seed_hi */
  SLIDE_OP_UPDATE_FEAT = 11,/* p: x, eps, noise(or NULL), t_dev, keypoint, c_recip, c_recipm1, c1, c2, c_std, [10] complete_x0 (n_pts*C) + [11] keypoint_mask (n_pts) of the local re-sampling mode (both NULL = off), [12] per-point table feat0 [n_pts][i[6]] (fp16 if i[7]) or NULL + [13] SlidePrepCopy[i[8]]: with fixed key points the update also writes the feature columns SLIDE_OP_PREP_POINTS would derive from the new state (the preparation then runs once per chain, not per step)  i: n_pts, C, kdim, seed_lo, seed_hi, eps_ld (0 = C), ldf, feat0 is fp16, n_copies  f: clamp.  In-kernel noise is keyed on (seed, t_dev[3] = chain nonce, t_dev[1] = step, GLOBAL element = element + t_dev[4] * elements per sample; t_dev[4] = global index of the chain's first sample) */
  SLIDE_OP_ADVANCE_T = 12,  /* p: t_dev  (t_dev[0] -= 1; t_dev[1] += 1); t_dev = [t, step, blocks-done counter, chain nonce, global index of the first sample, 3 spare] (8 ints) */
  SLIDE_OP_SYNC = 14,       /* i: from_lane, to_lane -- lane `to` waits for everything issued so far on lane `from` */
  SLIDE_OP_GROUPNORM_NCHW = 13,/* p: x, gamma, beta, y (NCHW fp32)   i: B, C, HW, G, n_norm, relu  (module-level path) */
  SLIDE_OP_ATTN_TAIL = 16,  /* fp16: scores GEMM + values GEMM (GroupNorm, ReLU) + softmax-weighted sum over the neighbours in one launch.  p: u, W5, mo, Wv, out, vec [bias_s | bias_v | gamma | beta][n_cob*32], [6] optional chunk-major copy of out [c/32][points][32] (a gather table of the next block: SLIDE_OP_GEMM f[2] == 32 reads p[8] that way), [7] optional copy of the first f[3] channels into another per-point buffer with leading dimension f[2]   i: rows, u_ld, k1, mo_ld, k2, n_cob, npx_log2, gs, n_norm, out_ld   f: 1 / (gs * rows per sample), [1] bit 0: both weight matrices chunk-major [k/32][n_cob*32][32] (u / mo are chunk-major [k/32][rows][32] when their ld is 32), bit 1: the two-stage-ring form at three workgroups per CU, bit 3 (round 5, csrc/gemm_gxs.hip attn_tail_split_kernel): SPLIT arithmetic -- u, mo and out are FLOAT rows, both weight matrices float row-major [n_cob*32][k], every contraction as two-term fp16 operand splits on one accumulator set (|w| < 32 required), bit 4 (round 6): u and mo are FRAGMENT-major (SLIDE_F_OUT_FM; register-X kernel only: the launcher refuses other forms) */
  SLIDE_OP_GEMM_GX = 17,    /* fp16 "generated-X" GEMM of the pair decomposition (gemm_gx.hip): the first layer of an SA / FP
                             * block is linear in [neighbour features | coordinates], so its output for row (point p, slot j) is ta[q] + tb[p]
                             * (q = the slot's neighbour) -- the 256- / 128-row activation this GEMM consumes is never stored: a workgroup keeps
                             * the sample's two 16-row tables in LDS and builds its MFMA B fragments from them,
                             *   mode 0: x = max(ta[q] + tb[p] + d2 vd + w vw, 0) + add        (tables pre-normalised by SLIDE_OP_PAIR_NORM)
                             *   mode 1: x = max(ta[q] + tb[p] + d2 vd + w vw, 0) * scale + shift
                             * only the weights stream (LDS-DMA ring).  npx_log2 8: 16 x 16 rows in NATURAL neighbour order (q = j); 7: 16 x 8
                             * rows, q = nbr[(b*16+p)*16+j], d2 / w = the slot's squared distance / interpolation weight (vd = vw = 0 for 8).
                             * p: [0] ta, [1] W chunk-major [k_pad/32][n_cob*32][32], [2] epi, [3] scale, [4] shift (fp32 [b*in_bs + k], mode 1),
                             *    [5] tb, [6] add vectors fp32 [idx*add_idx_stride + b*add_bs + k] or NULL (mode 0), [7] idx (device int) or NULL,
                             *    [8] nbr table, [9] d2, [10] w (fp32 [B*16][16]; npx_log2 7), [11] vd | vw fp32 [b*vbs + {0, vbs/2} + k] (SLIDE_OP_PAIR_NORM's vv, column offset applied) or NULL
                             * i: rows, t_ld (elements between table rows), k_pad, n_cob, npx_log2, in_bs, mode, add_bs, add_idx_stride, vbs
                             * f: [0] = 1 (mode 1): 256 x 64 tiles at three workgroups per CU when their LDS fits; 2: 256 x 64 tiles at two per CU
                             *    (launches whose 128-channel grid leaves CUs empty); else 256 x 128 tiles
                             * SPLIT ARITHMETIC (round 5, csrc/gemm_gxs.hip): f[0] = 3 -- ta / tb are FLOAT tables, W float row-major
                             *    [n_cob*32][k_pad], outputs float rows; x is generated in fp32 and split into two fp16 terms like the weights,
                             *    three fp16 MFMAs per product into one accumulator set (|w| < 32 required); the PAIR residual of a mode-0
                             *    block reads float tables.  CHAINED second layer (16 x 16-row samples, mode 0, n_cob <= 2): p[12] = its float
                             *    weights [f[1]*32][f[2]], p[13] = its SlideEpi[f[1]] (GroupNorm epilogue, no add vector), f[2] == n_cob*32 --
                             *    this layer's output stays in registers (its SlideEpi.out is NULL) and only the second layer's is stored */
  SLIDE_OP_PAIR_NORM = 18,  /* per-point tables of the pair decomposition: a[q][c] = y[q][c] + wa[c] . xyz[q], b[p][c] = wb[c] . xyz[p]; for each
                             * 32-channel block by its SlideEpi (mode, gs, n_norm, inv_count, gamma, beta, stats_*; bias already in y):
                             *   NORM : GroupNorm statistics over the sample's (p, slot) pairs of a[q] + b[p] (+ d2 vd + w vw), folded into the
                             *          tables: ta = a g + (beta - mean g), tb = b g, vv = (vd g | vw g) with g = gamma rstd
                             *   STATS: per-channel sums / sums of squares of max(a[q] + b[p] + ..., 0); ta = a, tb = b, vv = (vd | vw)
                             *   RAW  : ta = a, tb = b, vv = (vd | vw)
                             * p: [0] y fp32 [B*16][ld], [1] xyz fp32 [B*16][3], [2] wa fp32 [ld][4], [3] wb fp32 [ld][4], [4] epi [ld/32],
                             *    [5] ta, [6] tb (fp16 [B*16][ld]), [7] nbr table or NULL (natural order, K = 16), [8] d2, [9] w, [10] vd | vw
                             *    fp32 [2][ld] (K = 8), [11] vv fp32 [B][2][ld] out (K = 8), [12] SlideGnFin* or NULL (version 2: the joint GroupNorm over the
                             *    attention's [query | key] concatenation is finalised here: scale / shift rows written, the STATS sums stay on chip)
                             * i: B, ld, K, version (2: one 1024-thread workgroup per sample, ld <= 2048; else one per 256 channels),
                             *    [4] = 1: ta / tb are FLOAT tables (split plans, round 5; version 2 only: 512-thread workgroups) */
  SLIDE_OP_SA_CHAIN = 19,   /* second_mlp -> rest_mlp of an SA block's Mlp_plus_t_emb in one launch, one workgroup per 16 x 16-row sample:
                             * h2 = relu(GN(W1 . h1 + b1)) + add1 with h1 = max(ta[q] + tb[p], 0) + add0 generated from the pair tables, kept in
                             * registers as the B fragments of mo = relu(GN(W2 . h2 + b2)) + ra[q] + rb[p] (pair residual), stored chunk-major.
                             * p: ta, tb, ra, rb (fp16 [B*16][t_ld], column offsets applied), W1 [k1/32][n1][32], W2 [n1/32][n2][32], vec1, vec2
                             *    ([bias | gamma | beta][n] fp32), [8] add0 fp32 [idx*add0_stride + b*add0_bs + k] or NULL, [9] idx or NULL,
                             *    [10] add1 fp32 [b*add1_bs + c] or NULL, [11] out [n2/32][B*256][32] fp16
                             * i: B, t_ld, k1, n1 (128 | 256), n2 (multiple of 256), gs1, gs2 (GroupNorm group sizes 4 | 8 | 16), add0_stride,
                             *    add0_bs, add1_bs      f: 1 / (gs1 * 256), 1 / (gs2 * 256), [2] != 0 (round 6): out is FRAGMENT-major (SLIDE_F_OUT_FM) */
  SLIDE_OP_PAIR_FIRST = 31, /* the per-point GEMM of a block's pair decomposition AND its pair-table pass (SLIDE_OP_PAIR_NORM version 1) in one
                             * launch: y = W . feat + bias never goes through memory.  The channel blocks [0, pair_cob0) are ordinary segments
                             * (the attention queries riding on the launch: common epilogue, SlideEpi as for SLIDE_OP_GEMM); the blocks from
                             * pair_cob0 on are the pair channels: their SlideEpi carries bias + what SLIDE_OP_PAIR_NORM reads.
                             * p: [0] X fp16 [rows][x_ld], [1] W fp16 [n_cob*32][k_pad], [2] epi [n_cob], [3] xyz, [4] wa, [5] wb (fp32 [ld][4]),
                             *    [6] ta, [7] tb (fp16 [rows][ld]), [8] nbr, [9] d2, [10] w, [11] vd | vw fp32 [2][ld], [12] vv out fp32 [B][2][ld]
                             *    (8 .. 12: K = 8 only)
                             * i: rows (= B*16), x_ld, k_pad, n_cob, pair_cob0, ld (= (n_cob - pair_cob0) * 32), K (16 | 8) */
  SLIDE_OP_GEMM_CHAIN = 32, /* consecutive 16-rows-per-sample SLIDE_OP_GEMM launches (fp16, no input affine / gather / statistics finalisation)
                             * as ONE launch (csrc/gemm_chain.hip): a workgroup owns 64 rows and walks the layers, each through the common
                             * epilogue of its SlideEpi descriptors.  p[0]: HOST pointer to SlideChainLayer[n] (device pointers inside), kept
                             * alive by the plan.  i: rows, n (<= 6) */
  SLIDE_OP_GEMM_GX_DUAL = 34,/* two independent SLIDE_OP_GEMM_GX of one block -- the mode-1 keys -> u layer and the mode-0 first Mlp layer of an FP
                             * block -- as ONE launch on 64-channel tiles: p[0] = HOST pointer to the two SlideOp (mode 1 first), kept alive
                             * by the plan; falls back to two launches when the LDS of the dual form does not fit */
  SLIDE_OP_SA_CHAIN_P = 35, /* an SA block's fused Mlp chain (SLIDE_OP_SA_CHAIN) and the per-point query GEMM of its attention (SLIDE_OP_GEMM, 16 rows per
                             * sample, input affine + statistics finalisation) in ONE launch -- both depend on the block's pair tables only:
                             * p[0] = HOST pointer to the two SlideOp (chain first), kept alive by the plan */
  SLIDE_OP_PP_STAGE = 36,   /* split plans (round 5, csrc/gemm_gxs.hip): a run of consecutive per-point launches of one block -- 16-rows-per-sample
                             * SLIDE_OP_GEMM in the split arithmetic (no gather / statistics finalisation / per-point pre_add) and SLIDE_OP_PAIR_NORM
                             * version 2 on float tables -- as ONE launch, one workgroup per sample walking the steps; the dense layers are exact
                             * fp32 FMA chains on the vector ALUs.  p[0] = HOST pointer to {int64 n, B; n records of 16 int64 slots} (device
                             * pointers inside; layout in csrc/gemm_gxs.hip slide_launch_pp_stage), kept alive by the plan.  i: B, n */
  SLIDE_OP_POINT_CHAIN = 37,/* round 5 (csrc/point_chain.hip): the per-point END of a denoiser step as one launch -- the last FP block's second
                             * Mlp (first_mlp + res_connect, second_mlp + fc_condition + residual: pointnet2_modules.py:119-176, :842-855) and the
                             * output head fc_lyaer (pointnet2_with_pcld_condition.py:480-483), four dependent 16-rows-per-sample GEMMs; writes
                             * the prediction eps.  p[0] = HOST pointer to a SlidePointChainArgs block */
  SLIDE_OP_GEMM_ATTEND = 38,/* round 6 (module-level path, fp16 rows): the score convolution of an AttentionModule (weight_conv's last conv,
                             * reference pointnet2_ops attention.py:86-95), the soft-max over the K neighbour rows of each point and the weighted
                             * sum of the value rows in ONE launch -- the score map never reaches memory (SLIDE_OP_GEMM + SLIDE_OP_ROWS_ATTN
                             * stored it and read it back).  p: [0] X fp16 [rows][x_ld], [1] W fp16 [n_cob*32][k_pad], [2] epi (bias only),
                             * [3] in_scale, [4] in_shift (deferred normalisation of X as for SLIDE_OP_GEMM, or NULL), [5] V fp16 [rows][ldv],
                             * [6] out fp16 [rows / K][ldo], [7] counts int32 [rows / K] or NULL, [8] vss fp32 [sample][2][ldv] (deferred
                             * normalisation of V) or NULL, [11] add vectors of X's deferred step.  i: rows, x_ld, k_pad, n_cob, K (4 | 8 | 16 | 32),
                             * in_bs, ldv, ldo, points per sample, ReLU after V's affine, C (logical channels).  f: [1] 256-row tiles per sample,
                             * [2] add_bs, [3] 2 * add_n + relu (X's deferred step, as SLIDE_OP_GEMM) */
  SLIDE_OP_HEAD_UPDATE = 33,/* output head (two per-point GEMMs with the GroupNorm between them) + DDPM update + device-side t -= 1 as one launch
                             * (csrc/engine.hip head_update_kernel): p[0] = HOST pointer to a SlideHeadArgs block */
  SLIDE_OP_BLOCK_BODY = 30, /* the whole K-expanded body of an SA / FP block whose widths are <= 256 channels in one launch (csrc/block_body.hip):
                             * Mlp tail -> mo, generated keys -> u, attention tail; one workgroup per sample, mo / u as MFMA operand fragments in
                             * registers, weights through one LDS-DMA ring.  p[0]: HOST pointer to the BodyArgs block (csrc/block_body.hip;
                             * its members are device pointers), kept alive by the plan.  i: npx_log2 (7 | 8), has_rest_mlp */
  SLIDE_OP_TRANSPOSE = 15,  /* p: in, out (fp32)   i: B, R, C, in_ld, out_ld, in_batch_stride, out_batch_stride, out_is_fp16: out[b][c][r] = in[b][r][c] (module-level path: NCHW <-> row-major) */
  /* Row-major module-level path (rows_ops.hip): an activation is [B * S][ld] (S rows per sample, ld = channels rounded up
   * to 32, pad columns zero), fp32 or -- i[9] = 1 -- fp16.  Replaces the reference's NCHW tensor program of
   * Mlp_plus_t_emb / AttentionModule / QueryAndGroup('nn') / group_knn (pointnet2_modules.py:71-176, attention.py:35-96,
   * pointnet2_utils.py:383-430, :497-524). */
  SLIDE_OP_ROWS_FROM_NCX = 20, /* p: in (B,C,P) fp32, out rows   i: B, C, P, ld */
  SLIDE_OP_ROWS_TO_NCX = 21,   /* p: in rows, out (B,C,P) fp32   i: B, C, P, ld */
  SLIDE_OP_ROWS_GROUP = 22,    /* p: xyz (B,N,3), new_xyz (B,np,3), feat rows [B*N][ldf] (or NULL), idx int64 (B,np,K), d2 (B,np,K) (FP layout), out [B*np*K][ldg]   i: B, N, np, K, C, ldf, ldg, flags (1: group_knn layout [feat|d2|w|abs|rel|centre]; else [feat|rel|abs if 2|centre if 4], 8: no coordinate channels, 16: idx is int32), [6] counts int32 (B,np) or NULL: a centre with count 0 stands in as its own neighbour with zero features */
  SLIDE_OP_ROWS_GN = 23,       /* p: x, gamma, beta, addvec [B][addvec_ld] fp32 (or NULL), residual rows (or NULL), scratch (B*64*ld*2 + B*2*ld floats), y (may be x), [7] / [8] per-tile channel sums / sums of squares [B*i[8]][ld] from the producing GEMM's STATS epilogue (256-row tiles, i[8] tiles per sample) instead of a statistics pass, [10] optional OUT: the statistics [B][64][mean | rstd] fp32 (the training step's backward, slide_train.h)   i: B, S, ld, G (0 = no normalisation), n_norm, flags (1 ReLU before, 2 ReLU after, 4 statistics + scale / shift only -> p[9] [B][2][ld] fp32, for a consumer GEMM with the deferred affine; 8 apply only with p[9]), addvec_ld, res_ld, tiles per sample */
  SLIDE_OP_ROWS_CONCAT_QK = 24,/* p: q [rows/K][ldq], k [rows][ldk], out [rows][ldo] = relu([q | k])   i: rows, K, C1, ldq, C2, ldk, ldo */
  SLIDE_OP_ROWS_ATTN = 25,     /* p: scores [pts*K][lds], values [pts*K][ldv], out [pts][ldo], [3] counts int32 [pts] or NULL (softmax over the first max(1,count) slots), [4] deferred normalisation of the values: scale / shift [sample][2][ldv] fp32 or NULL   i: pts, K, C, lds, ldv, ldo, points per sample, ReLU after the affine */
  SLIDE_OP_ROWS_POOL = 26,     /* p: x [pts*K][ldx], out [pts][ldo], counts int32 [pts] or NULL   i: pts, K, C, ldx, ldo, mode (0 max, 1 mean over the counted slots, 2 max for channels < C/2 and mean for the rest) */
  SLIDE_OP_ROWS_PAIR_EXPAND = 39, /* round 6 (fp16 rows): a 1 x 1 convolution over a GROUPED input without the grouped matrix -- out[(b, p, k)][c] =
                             * A[b*N + idx[b][p][k]][c] + bias[c] + coef[c][0..2] . xyz[neighbour] + coef[c][3..5] . xyz[centre] + coef[c][6] d2 +
                             * coef[c][7] w, A = the convolution's feature columns applied per SOURCE point (fp32 [B*N][ldA], or NULL without
                             * features), coef = (W_rel + W_abs: applied by the host program to A | W_centre - W_rel | w_d2 | w_w) per output channel, fp32 [ld][8]; flags: 1 = ReLU,
                             * 2 = group_knn slot scalars (d2, w = normalised inverse squared distance; p[6] = d2 (B, np, K)), 16 = idx is int32.
                             * p: A, bias [ld], coef, xyz (B,N,3), new_xyz (B,np,3), idx (B,np,K), d2 or NULL, out [B*np*K][ld], [8] / [9] per-256-row-
                             * tile channel sums / sums of squares of the (ReLU'd) output [rows/256][ld] or both NULL.  i: B, N, np, K, ld, ldA, flags */
  SLIDE_OP_ROWS_GN_JOINT = 27  /* round 5: GroupNorm over the VIRTUAL concatenation [q(point) x K | k(point, neighbour)] of an AttentionModule (attention.py:45-47) from the per-256-row-tile channel sums of the two producing GEMMs (STATS epilogue): p: qsum, qsq [B*tq][ldq], ksum, ksq [B*tk][ldk], gamma, beta [n_norm], OUT ssq [B][2][ldq], ssk [B][2][ldk] (scale | shift per sample and channel in the producers' layouts)   i: B, C1, ldq, tq, K, C2, ldk, tk, G   f: 1 / (rows of k per sample), n_norm */
};

/* GroupNorm finalisation folded into the first consumer (fp16 small-launch GEMM with input affine, SlideOp.p[6]):
 * per-(sample, channel) sums of a channel-concatenated tensor whose groups straddle producers -> scale / shift.
 * The kernel computes them for the samples of its tile, uses its own channel slice and writes the full rows to
 * scale / shift for the later consumers (what SLIDE_OP_FINALIZE_GN does as a separate launch). */
typedef struct SlideGnFin {
  const float *sum, *sq;           /* [B][bs] */
  const int *gid, *gstart, *gend;  /* channel -> group (-1 = not normalised); group -> physical channel range */
  const float *gamma, *beta;       /* [C] */
  float *scale, *shift;            /* [B][bs] outputs */
  float inv_count;
  int C, bs, G;                    /* channels, row stride of sum / sq / scale / shift, number of groups (<= 32) */
} SlideGnFin;

/* extra destinations of SLIDE_OP_PREP_POINTS (SlideOp.p[6], i[4] entries): the first n columns of every point's
 * [features | xyz] row (kind 0) or its xyz (kind 1, n = 3) are also written to dst[(b*16 + p)*ld + c] (activation type) --
 * columns of later concatenation buffers, saving their COPY launches */
typedef struct SlidePrepCopy {
  void *dst;
  int32_t ld, kind, n, pad;
} SlidePrepCopy;

/* one layer of SLIDE_OP_GEMM_CHAIN: X fp16 [rows][x_ld] (k_pad columns read), W fp16 row-major [n_cob*32][k_pad], epi SlideEpi[n_cob] */
typedef struct SlideChainLayer {
  const void *X, *W, *epi;
  int32_t x_ld, k_pad, n_cob, pad;
} SlideChainLayer;

/* SLIDE_OP_HEAD_UPDATE (p[0] = HOST pointer to this block, device pointers inside; kept alive by the plan): the output head
 * fc_lyaer (conv 128 -> GroupNorm(32, 128) -> ReLU -> conv) and the sampler's DDPM update in one launch */
typedef struct SlideHeadArgs {
  const void *X, *W0, *W1;     /* fp16: head input [rows][x_ld]; W0 [128][k0] row-major; W1 [n1c*32][128] row-major */
  const float *v0, *b1;        /* [bias | gamma | beta][128] of layer 1; bias [n1c*32] of layer 2 */
  float *eps_out;              /* optional copy of the prediction [rows][eps_ld], or NULL */
  int32_t rows, x_ld, k0, n1c, eps_ld;
  int32_t kind, C, kdim, ldf, half_out, n_copies; /* update: 0 = position (SLIDE_OP_UPDATE_POS), 1 = feature (SLIDE_OP_UPDATE_FEAT) */
  float clamp;
  uint32_t seed_lo, seed_hi;
  float *x;
  const float *noise;
  int32_t *t_dev;
  const float *keypoint, *t0, *t1, *t2, *t3, *t4, *complete_x0, *kmask; /* schedule tables in the update op's order */
  void *feat0;
  const struct SlidePrepCopy *copies;
} SlideHeadArgs;

/* SLIDE_OP_POINT_CHAIN (p[0] = HOST pointer to this block, device pointers inside; kept alive by the plan).  Every layer is 128 channels
 * wide with GroupNorm(32, 128); fp16 operands / activations, fp32 accumulation, statistics and residual. */
typedef struct SlidePointChainArgs {
  const void *Z;        /* fp16 [rows][z_ld]: the block's per-point input [attention output | skip features | xyz], kz (<= 192) columns read */
  const void *Wz;       /* fp16 [256][kz] row-major: first_mlp.0 (rows 0..127) | res_connect (rows 128..255) */
  const void *W2;       /* fp16 [128][128]: second_mlp.0 */
  const void *W0, *W1;  /* fp16 head: [128][k0], [n1c*32][128] */
  const float *vz;      /* [bias | gamma | beta] of first_mlp, bias of res_connect: [4][128] */
  const float *v2;      /* [bias | gamma | beta][128] of second_mlp */
  const float *v0, *b1; /* head: [bias | gamma | beta][128]; bias [n1c*32] */
  const float *tvec;    /* added after first_mlp's ReLU: tvec[(t_idx ? t_idx[0] * t_stride : 0) + sample * t_bs + c] (row t of the per-timestep table; or, t_idx NULL and t_bs > 0, the sample's own t-embedding row), or NULL */
  const int32_t *t_idx;
  const float *cvec;    /* added after second_mlp's ReLU: cvec[sample * c_bs + c], or NULL */
  void *X;              /* fp16 [rows][x_ld], the head's input: columns [128, k0) are READ ([xyz | zero pad]), columns [0, 128) WRITTEN (the block's output) */
  float *eps;           /* OUT fp32 [rows][eps_ld]: the prediction (eps_ld <= 32 n1c, a multiple of 4) */
  const void *Wz_lo, *W2_lo, *W0_lo, *W1_lo; /* round 6, WIDE form (all four set, or all NULL): fp16 LOW fragments of the four weight
                         * matrices, lo' = fp16((w - fp16(w)) * 2^11), in the layouts of Wz / W2 / W0 / W1 -- the chain then runs in the
                         * split arithmetic (fp32-grade products: three fp16 MFMAs per product, activations cross the waves as two
                         * fp16 planes) */
  int32_t rows, z_ld, kz, x_ld, k0, n1c, eps_ld, t_stride, t_bs /* elements between the SAMPLES' rows of tvec (0: one shared row) */, c_bs;
  int32_t fuse_update;  /* 1: the launch also applies the feature DDPM's update (upd.kind == 1) to the state and advances the device-side
                         * timestep -- the lane that holds eps[row][channel] updates x[row][channel]; the noise of the workgroup's elements
                         * is drawn while the chain's loads are in flight.  Of `upd` the update members are read (kind ... copies). */
  SlideHeadArgs upd;
} SlidePointChainArgs;

typedef struct SlideOp {
  int32_t kind;
  int32_t i[11];
  float f[4];
  void *p[14];
} SlideOp;

/* launches ops[0..n) in order on `stream` (HOST array).  Safe inside hipGraph stream capture. */
SLIDE_API int slide_run_ops(const SlideOp *ops, int n, slide_stream_t stream);
/* two-lane variant: op.i[10] (0/1) picks the stream, SLIDE_OP_SYNC orders the lanes; independent branches of a plan
 * (e.g. the query / score branch and the value branch of an attention block) then overlap, eagerly or as parallel
 * branches of a captured hipGraph (capture on stream0; every lane must be joined back into lane 0 at the end). */
SLIDE_API int slide_run_ops2(const SlideOp *ops, int n, slide_stream_t stream0, slide_stream_t stream1);
/* `reps` back-to-back eager replays of one plan (one host thread per chain / stream; no SLIDE_OP_SYNC across threads) */
/* `reps` steps of n_chains independent single-lane plans, issued round-robin from the calling thread (chain c on streams[c]) */
SLIDE_API int slide_run_chains(const SlideOp *const *ops, const int *n, const slide_stream_t *streams, int n_chains, int reps);
/* ... chain c issues a step only on rounds r with r % every[c] == 0 (every[c] >= 1) */
SLIDE_API int slide_run_chains_every(const SlideOp *const *ops, const int *n, const slide_stream_t *streams, const int *every,
                                     int n_chains, int reps);
SLIDE_API int slide_run_ops_repeat(const SlideOp *ops, int n, slide_stream_t stream0, slide_stream_t stream1, int reps);

/* same, eagerly, with a HIP event recorded on `stream` between consecutive launches; ms_out[i] (HOST, n floats)
 * receives the device time of ops[i].  Synchronises.  For per-kernel roofline figures, never on a timed path. */
SLIDE_API int slide_run_ops_timed(const SlideOp *ops, int n, slide_stream_t stream, float *ms_out);

/* hipGraph helpers: capture everything launched on `stream` between begin/end, replay it later. */
SLIDE_API int slide_graph_begin(slide_stream_t stream);
SLIDE_API int slide_graph_end(slide_stream_t stream, void **graph_exec_out);
SLIDE_API int slide_graph_launch(void *graph_exec, slide_stream_t stream);
SLIDE_API int slide_graph_destroy(void *graph_exec);

/* HIP-event timing on an arbitrary stream (torch.cuda.Event only sees torch's current stream) */
SLIDE_API int slide_event_create(void **ev);
SLIDE_API int slide_event_record(void *ev, slide_stream_t stream);
SLIDE_API int slide_event_elapsed_ms(void *start, void *stop, float *ms); /* synchronises on `stop` */
SLIDE_API int slide_event_destroy(void *ev);

/* A stream whose kernels run on a SUBSET of the compute units (hipExtStreamCreateWithCUMask): mask = n_words x 32 bits, bit i = CU i.
 * For a chain that has slack on its own queue but whose wide launches would otherwise take every CU from the latency-critical
 * chains beside it (the position DDPM beside the feature sub-batches). */
SLIDE_API int slide_stream_create_cu_mask(const uint32_t *mask, int n_words, slide_stream_t *stream_out);
SLIDE_API int slide_stream_destroy(slide_stream_t stream);

SLIDE_API int slide_sizeof_epi(void);
SLIDE_API int slide_sizeof_op(void);

#ifdef __cplusplus
}
#endif
#endif
