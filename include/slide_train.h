/*
 * slide_train.h -- C-ABI of the BACKWARD kernels of the row-major layers in libslide_hip.so (the
 * training step of the latent DDPMs).
 *
 * What they differentiate (reference, paths relative to /root/reference):
 *   MyGroupNorm + ReLU                       pointnet2_ops_lib/pointnet2_ops/pointnet2_modules.py:24-69
 *   grouping_operation / knn_gather (feats)  pointnet2_ops_lib/pointnet2_ops/pointnet2_utils.py:222-268, :506-507
 *                                            (the reference's native backward: _ext-src/src/group_points_gpu.cu:30-60)
 *   relu(cat([q.expand, k])), softmax over K + weighted sum    pointnet2_ops_lib/pointnet2_ops/attention.py:78-95
 * and whose forward counterparts are the module path's kernels (slide_engine.h: SLIDE_OP_ROWS_GN / _GROUP / _CONCAT_QK / _ATTN).
 * The reference gets these gradients from torch.autograd over ~80 module launches per forward (pointnet2/train.py,
 * pointnet2/train_latent_ddpm.py; losses: pointnet2/util.py:262-300, pointnet2/diffusion_utils/diffusion.py:319-341).
 *
 * All matrices are row-major fp32 [rows][ld] DEVICE arrays (ld = channels rounded up to 32, pad columns zero); outputs are
 * written in full unless stated.  Status: 0 = ok, else hipError_t (< 0: bad arguments).  Asynchronous on `stream`.
 */
#ifndef SLIDE_TRAIN_H
#define SLIDE_TRAIN_H

#include <stdint.h>

#include "slide_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* y = post_relu?(GroupNorm_G(pre_relu?(x))) on [B*S][ld]: the first n_norm channels in G <= 64 groups over (group x S rows of a
 * sample), the rest pass through (MyGroupNorm); flags: 1 = ReLU before, 2 = ReLU after.  G = 0: no normalisation (ReLUs only;
 * gamma, beta, mean_rstd, dgamma, dbeta, scratch may be NULL).  mean_rstd [B][64][2]: the forward's statistics
 * (SLIDE_OP_ROWS_GN p[10], slide_engine.h).  Writes dx [B*S][ld] and the PER-SAMPLE parameter gradients dgamma, dbeta [B][ld]
 * (zero beyond n_norm; the caller sums over B: a deterministic reduction).  scratch: B * (64 * ld * 2 + 128) floats. */
SLIDE_API int slide_gn_rows_bwd(int B, int S, int ld, int G, int n_norm, int flags, const float *x, const float *gamma,
                                const float *beta, const float *mean_rstd, const float *dy, float *dx, float *dgamma, float *dbeta,
                                float *scratch, slide_stream_t stream);

/* out [ld] = column sums of x [rows][ld] (the bias gradient of a convolution: the sum of dy over the rows; GroupNorm's parameter
 * gradients over the batch).  Two launches: row chunks, then 32-column stripes over the partial rows; deterministic.
 * scratch: 1024 * ld floats (unused when rows < 128). */
SLIDE_API int slide_col_sums(long long rows, int ld, const float *x, float *out, float *scratch, slide_stream_t stream);

/* grouped rows out[(b,p,k)][0..C) = feat[b][idx[b][p][k]][0..C): dfeat [B*N][ldf] += dout [B*np*K][ldg] (atomic; dfeat must be
 * zero-initialised); counts (B*np) int32 or NULL: centres with count 0 carried zero features and receive nothing. */
SLIDE_API int slide_group_rows_bwd(int B, int N, int np, int K, int C, int ldf, int ldg, const int64_t *idx, const int *counts,
                                   const float *dout, float *dfeat, slide_stream_t stream);

/* out = relu([q(point) broadcast over K | k(point, neighbour)]) with q [pts][ldq] (C1 channels), k [pts*K][ldk] (C2), out
 * [pts*K][ldo]: dq, dk from dout and the forward OUTPUT (the ReLU mask); the first C1 / C2 channels of dq / dk are written. */
SLIDE_API int slide_concat_qk_bwd(long long pts, int K, int C1, int ldq, int C2, int ldk, int ldo, const float *out,
                                  const float *dout, float *dq, float *dk, slide_stream_t stream);

/* out[pt][c] = sum_k softmax_k(s)[k][c] v[(pt,k)][c] over the first max(1, count) of the K neighbour rows (counts NULL: all K):
 * ds, dv [pts*K][lds | ldv] (first C channels) from s, v and dout [pts][ldo]. */
SLIDE_API int slide_attn_rows_bwd(long long pts, int K, int C, int lds, int ldv, int ldo, const float *s, const float *v,
                                  const int *counts, const float *dout, float *ds, float *dv, slide_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
