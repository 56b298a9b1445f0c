/*
 * slide_resident.h -- C-ABI of the LDS-resident denoiser kernel in libslide_hip_exp.so (the EXPERIMENTS build; not in the product library).
 *
 * What it replaces: the same thing include/slide_engine.h's plan replaces -- one reverse-diffusion step of the
 * reference = PointNet2CloudCondition.forward (pointnet2/models/pointnet2_with_pcld_condition.py:286-489) + the DDPM
 * update (pointnet2/util.py:247-253) -- but for networks whose per-sample working set fits ONE compute unit's 160 KB of
 * LDS (the position DDPM: channel widths <= 128): a workgroup owns a sample, keeps every activation in LDS, streams the
 * fp16 weights from L2 straight into MFMA operand registers, and runs ALL layers AND all requested timesteps in a single
 * launch (samples never exchange anything: GroupNorm is per sample).  The engine plan needs 43 launches per step for the
 * same work.
 *
 * A step is a short "program" of ROp records (host-built, device memory); an op covers whole layers:
 *   PREP      x -> xyz, point-feature table, sorted 16x16 neighbour table          (QueryAndGroup 'nn' / group_knn search)
 *   ASSEMBLE  coordinate channels of the grouped input (pointnet2_utils.py:383-430, :497-524); the feature channels are
 *             never materialised -- the GEMM reads them from the neighbour's row of the point-feature table
 *   GEMM      1x1 convolution(s) sharing one input, one 32-channel strip per wave: bias, per-point pre-activation term,
 *             ReLU, GroupNorm, t-/class-embedding add, and optionally a SECOND accumulation phase into the normalised
 *             registers (res_connect of Mlp_plus_t_emb, pointnet2_modules.py:119-176)
 *   FINALIZE  GroupNorm statistics of the virtual concatenation [query | keys] (attention.py weight_conv.1)
 *   AFFINE    x <- relu(x) * scale + shift in place
 *   TAIL      AttentionModule tail (attention.py:86-95): scores GEMM, values GEMM + GroupNorm + ReLU, softmax over the
 *             neighbours, weighted sum -> per-point table
 *   ZFILL     skip-feature / coordinate columns of a concatenation buffer
 *   UPDATE    DDPM update of the state kept in LDS, in-kernel Philox noise, t -= 1
 *
 * All pointers are DEVICE pointers.  Status: 0 = ok, else hipError_t (or < 0 for bad arguments).
 */
#ifndef SLIDE_RESIDENT_H
#define SLIDE_RESIDENT_H

#include <stdint.h>

#include "../slide_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { SLIDE_R_PREP = 1, SLIDE_R_ASSEMBLE = 2, SLIDE_R_GEMM = 3, SLIDE_R_FINALIZE = 4, SLIDE_R_AFFINE = 5, SLIDE_R_TAIL = 6,
       SLIDE_R_ZFILL = 7 };
enum { SLIDE_RS_RAW = 0, SLIDE_RS_NORM = 1, SLIDE_RS_STATS = 2 };
enum { SLIDE_RF_PRE_RELU = 1, SLIDE_RF_POST_RELU = 2, SLIDE_RF_OUT_F32 = 4 };
enum { SLIDE_RO_BARRIER_BEFORE_STORE = 1 };

/* one GEMM input: the first nks_gat 16-channel K steps of row (point, neighbour) are read from row nbr(point, neighbour)
 * of a 16-row table at LDS byte offset gat_off (row length gat_ld halfs), the next nks_x steps from row-major buffer
 * x_off / x_ld.  Offsets are bytes into the workgroup's LDS arena, leading dimensions are in fp16 elements. */
typedef struct RIn {
  int32_t gat_off, gat_ld, nks_gat, x_off, x_ld, nks_x;
} RIn;

typedef struct ROp {
  int32_t type, rows_log2, kshift, n_strips, parts, strip0, flags, pad0;
  RIn a, b;       /* accumulation phase A / B (b.nks_* == 0: no second phase).  TAIL: a = scores input, b = values input */
  int32_t p[8];   /* type specific, see resident.hip */
  float f[4];
} ROp;

/* one 32-channel output strip of a GEMM / TAIL op */
typedef struct RStrip {
  int32_t mode, flags, gs, n_valid, n_store;
  int32_t out_off, out_ld, out_col; /* LDS bytes, halfs (floats with OUT_F32), first column */
  int32_t wfrag;                    /* first 1-KB MFMA A-operand fragment in the weight pool (phase A steps, then phase B) */
  int32_t vec_off;                  /* floats into the vector pool: [bias 32 | gamma 32 | beta 32 | bias2 32] */
  int32_t addvec_kind, addvec_off;  /* 0 none, 1 t-embedding row, 2 class-embedding row; float offset in that row */
  int32_t preadd_off, preadd_ld;    /* per-point pre-activation table in LDS (bytes / halfs), -1 = none */
  int32_t stats_off;                /* STATS: LDS byte offset of this strip's [part][32][2] float sums */
  float inv_count;                  /* NORM: 1 / (group size x rows per sample) */
} RStrip;

typedef struct RArgs {
  const ROp *ops;
  const RStrip *strips;
  const void *wpool;    /* fp16 fragments, 1 KB each: lane l holds W[32 s + (l & 31)][16 f + 8 (l >> 5) .. +7] */
  const float *vpool;
  const float *tvec;    /* [T or B][tvec_ld] */
  const float *cvec;    /* [B][cvec_ld] */
  float *x;             /* [B][16][cx] state, read at launch, written back at the end */
  float *eps_out;       /* [B][16][out_dim] or NULL: the last step's network output (forward API / tests) */
  int32_t *t_dev;       /* [t, step, blocks-done counter, chain nonce] or NULL (forward mode: no update) */
  const float *c_eps, *sqrt_alpha, *sigma; /* position DDPM tables [T] */
  const float *noise;   /* explicit noise [steps][B*16*cx] or NULL (in-kernel Philox) */
  void *dbg;            /* NULL, or [B][lds_bytes]: every workgroup dumps its LDS arena at the end (tests / debugging) */
  uint64_t *timeline;   /* NULL, or [n_ops + 2]: shader-clock stamps of workgroup 0 around every op of its first step */
  int32_t n_ops, n_steps, B, cx, out_dim, per_sample_t, tvec_ld, cvec_ld;
  int32_t xstate_off, xyz_off, knn_off, kd2_off, eps_off, lds_bytes;
  uint32_t seed_lo, seed_hi;
} RArgs;

/* runs n_steps reverse steps (or one forward when t_dev == NULL) for all B samples in ONE launch */
SLIDE_API int slide_resident_run(const RArgs *args, slide_stream_t stream);
SLIDE_API int slide_sizeof_rop(void);
SLIDE_API int slide_sizeof_rstrip(void);
SLIDE_API int slide_sizeof_rargs(void);

#ifdef __cplusplus
}
#endif
#endif
