/*
 * oracle/ops_cpu.c -- TEST INFRASTRUCTURE ONLY (never shipped, never measured as product).
 *
 * Plain-C CPU restatement of the reference's pointnet2_ops `_ext` kernels plus the two
 * pytorch3d ops the executed configs call.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library.
 *
 * Each function cites the reference source it follows (paths relative to
 * /root/reference/pointnet2_ops_lib/pointnet2_ops/_ext-src/).  The reference has NO CPU
 * path for these ops (sampling.cpp:34 `AT_ASSERT(false, "CPU not supported")`) and cannot
 * be compiled here (CUDA only), so this restatement is pinned by (a) golden vectors
 * produced by importing the reference *Python* with this library plugged in as `_ext`
 * (tests/golden/, tools/gen_golden.py) and (b) hand-derived known answers in
 * tests/test_oracle_ops.py.  The reference's own tests pin nothing on this path
 * ("parity unpinned by the reference's own tests").
 *
 * Floating-point recipe (shared bit-for-bit with the HIP kernels):
 *   d = fmaf(dz, dz, fmaf(dy, dy, dx * dx))       with compiler contraction OFF.
 * Build with -ffp-contract=off (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORA_API __attribute__((visibility("default")))

static inline float sqdist3(float ax, float ay, float az, float bx, float by, float bz) {
  const float dx = ax - bx, dy = ay - by, dz = az - bz;
  return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

/* include/cuda_utils.h:13-19  opt_n_threads */
ORA_API int ora_opt_n_threads(int work_size) {
  const int pow_2 = (int)(log((double)work_size) / log(2.0));
  int v = 1 << pow_2;
  if (v > 512) v = 512;
  if (v < 1) v = 1;
  return v;
}

/* src/sampling_gpu.cu:8-20  gather_points_kernel: out[b,c,j] = points[b,c,idx[b,j]] */
ORA_API void ora_gather_points(int b, int c, int n, int m, const float *points, const int *idx,
                               float *out) {
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < m; ++j) {
        const int a = idx[i * m + j];
        out[((size_t)i * c + l) * m + j] = points[((size_t)i * c + l) * n + a];
      }
}

/* src/sampling_gpu.cu:34-47  gather_points_grad_kernel (atomicAdd scatter; here sequential) */
ORA_API void ora_gather_points_grad(int b, int c, int n, int m, const float *grad_out,
                                    const int *idx, float *grad_points) {
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < m; ++j) {
        const int a = idx[i * m + j];
        grad_points[((size_t)i * c + l) * n + a] += grad_out[((size_t)i * c + l) * m + j];
      }
}

/*
 * src/sampling_gpu.cu:59-173  furthest_point_sampling_kernel<block_size>, host sampling.cpp:66-87.
 * Literal simulation of the block: `bs` lanes, lane t owns k = t, t+bs, ...; per-lane
 * strict-'>' argmax (lines 108-109), then the shared-memory tree of __update (lines 59-65,
 * 115-168) where a tie keeps the LOWER slot.  The near-origin skip (lines 100-101) compares
 * the float magnitude against the DOUBLE literal 1e-3.  `temp` is read and updated in place,
 * exactly like the kernel does in global memory.
 */
ORA_API void ora_furthest_point_sampling(int b, int n, int m, const float *dataset, float *temp,
                                         int *idxs) {
  if (m <= 0) return;
  const int bs = ora_opt_n_threads(n);
  float *dists = (float *)malloc(sizeof(float) * (size_t)bs);
  int *dists_i = (int *)malloc(sizeof(int) * (size_t)bs);
  for (int bi = 0; bi < b; ++bi) {
    const float *pts = dataset + (size_t)bi * n * 3;
    float *tmp = temp + (size_t)bi * n;
    int *out = idxs + (size_t)bi * m;
    int old = 0;
    out[0] = old;
    for (int j = 1; j < m; ++j) {
      const float x1 = pts[old * 3 + 0], y1 = pts[old * 3 + 1], z1 = pts[old * 3 + 2];
      for (int tid = 0; tid < bs; ++tid) {
        int besti = 0;
        float best = -1.0f;
        for (int k = tid; k < n; k += bs) {
          const float x2 = pts[k * 3 + 0], y2 = pts[k * 3 + 1], z2 = pts[k * 3 + 2];
          const float mag = fmaf(z2, z2, fmaf(y2, y2, x2 * x2));
          if ((double)mag <= 1e-3) continue;
          const float d = sqdist3(x2, y2, z2, x1, y1, z1);
          const float d2 = d < tmp[k] ? d : tmp[k]; /* min(d, temp[k]) */
          tmp[k] = d2;
          besti = d2 > best ? k : besti;
          best = d2 > best ? d2 : best;
        }
        dists[tid] = best;
        dists_i[tid] = besti;
      }
      for (int s = bs / 2; s >= 1; s >>= 1) {
        for (int tid = 0; tid < s; ++tid) {
          const float v1 = dists[tid], v2 = dists[tid + s];
          const int i1 = dists_i[tid], i2 = dists_i[tid + s];
          dists[tid] = v1 > v2 ? v1 : v2; /* max(v1, v2) */
          dists_i[tid] = v2 > v1 ? i2 : i1;
        }
      }
      old = dists_i[0];
      out[j] = old;
    }
  }
  free(dists);
  free(dists_i);
}

/*
 * src/ball_query_gpu.cu:9-47 query_ball_point_kernel; host ball_query.cpp:10-38 (idx and counts
 * zero-initialised by the caller).  Strict `d2 < radius*radius`, scan order k ascending, first hit
 * floods all nsample slots.
 */
ORA_API void ora_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                            const float *xyz, int *idx, int *counts) {
  const float radius2 = radius * radius;
  for (int bi = 0; bi < b; ++bi) {
    const float *p = xyz + (size_t)bi * n * 3;
    const float *q = new_xyz + (size_t)bi * m * 3;
    int *oi = idx + (size_t)bi * m * nsample;
    int *oc = counts + (size_t)bi * m;
    for (int j = 0; j < m; ++j) {
      const float nx = q[j * 3 + 0], ny = q[j * 3 + 1], nz = q[j * 3 + 2];
      for (int k = 0, cnt = 0; k < n && cnt < nsample; ++k) {
        const float d2 = sqdist3(nx, ny, nz, p[k * 3 + 0], p[k * 3 + 1], p[k * 3 + 2]);
        if (d2 < radius2) {
          if (cnt == 0)
            for (int l = 0; l < nsample; ++l) oi[j * nsample + l] = k;
          oi[j * nsample + cnt] = k;
          ++cnt;
          oc[j] = cnt;
        }
      }
    }
  }
}

/* src/group_points_gpu.cu:8-28  out[b,l,j,k] = points[b,l,idx[b,j,k]] */
ORA_API void ora_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                              const int *idx, float *out) {
  for (int bi = 0; bi < b; ++bi) {
    const float *p = points + (size_t)bi * n * c;
    const int *id = idx + (size_t)bi * npoints * nsample;
    float *o = out + (size_t)bi * npoints * nsample * c;
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < npoints; ++j)
        for (int k = 0; k < nsample; ++k)
          o[((size_t)l * npoints + j) * nsample + k] = p[(size_t)l * n + id[j * nsample + k]];
  }
}

/* src/group_points_gpu.cu:43-64  scatter-add of the forward gather */
ORA_API void ora_group_points_grad(int b, int c, int n, int npoints, int nsample,
                                   const float *grad_out, const int *idx, float *grad_points) {
  for (int bi = 0; bi < b; ++bi) {
    const float *g = grad_out + (size_t)bi * npoints * nsample * c;
    const int *id = idx + (size_t)bi * npoints * nsample;
    float *gp = grad_points + (size_t)bi * n * c;
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < npoints; ++j)
        for (int k = 0; k < nsample; ++k)
          gp[(size_t)l * n + id[j * nsample + k]] += g[((size_t)l * npoints + j) * nsample + k];
  }
}

/*
 * src/interpolate_gpu.cu:9-59  three_nn_kernel.  bests are DOUBLE 1e40 (line 27), compared
 * against the float distance with strict '<' cascades; outputs are cast back to float
 * (1e40 -> +inf) when fewer than three known points exist.
 */
ORA_API void ora_three_nn(int b, int n, int m, const float *unknown, const float *known,
                          float *dist2, int *idx) {
  for (int bi = 0; bi < b; ++bi) {
    const float *u = unknown + (size_t)bi * n * 3;
    const float *kn = known + (size_t)bi * m * 3;
    float *od = dist2 + (size_t)bi * n * 3;
    int *oi = idx + (size_t)bi * n * 3;
    for (int j = 0; j < n; ++j) {
      const float ux = u[j * 3 + 0], uy = u[j * 3 + 1], uz = u[j * 3 + 2];
      double best1 = 1e40, best2 = 1e40, best3 = 1e40;
      int besti1 = 0, besti2 = 0, besti3 = 0;
      for (int k = 0; k < m; ++k) {
        const float d = sqdist3(ux, uy, uz, kn[k * 3 + 0], kn[k * 3 + 1], kn[k * 3 + 2]);
        if (d < best1) {
          best3 = best2; besti3 = besti2;
          best2 = best1; besti2 = besti1;
          best1 = d; besti1 = k;
        } else if (d < best2) {
          best3 = best2; besti3 = besti2;
          best2 = d; besti2 = k;
        } else if (d < best3) {
          best3 = d; besti3 = k;
        }
      }
      od[j * 3 + 0] = (float)best1; od[j * 3 + 1] = (float)best2; od[j * 3 + 2] = (float)best3;
      oi[j * 3 + 0] = besti1; oi[j * 3 + 1] = besti2; oi[j * 3 + 2] = besti3;
    }
  }
}

/*
 * src/interpolate_gpu.cu:72-101  three_interpolate_kernel:
 *   out = p[i1]*w1 + p[i2]*w2 + p[i3]*w3   (left to right).
 * Shared rounding recipe: fmaf(p3, w3, fmaf(p2, w2, p1 * w1)).
 */
ORA_API void ora_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                                   const float *weight, float *out) {
  for (int bi = 0; bi < b; ++bi) {
    const float *p = points + (size_t)bi * m * c;
    const int *id = idx + (size_t)bi * n * 3;
    const float *w = weight + (size_t)bi * n * 3;
    float *o = out + (size_t)bi * n * c;
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < n; ++j) {
        const float w1 = w[j * 3 + 0], w2 = w[j * 3 + 1], w3 = w[j * 3 + 2];
        const int i1 = id[j * 3 + 0], i2 = id[j * 3 + 1], i3 = id[j * 3 + 2];
        o[(size_t)l * n + j] =
            fmaf(p[(size_t)l * m + i3], w3, fmaf(p[(size_t)l * m + i2], w2, p[(size_t)l * m + i1] * w1));
      }
  }
}

/* src/interpolate_gpu.cu:116-143  scatter-add x3 */
ORA_API void ora_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                                        const int *idx, const float *weight, float *grad_points) {
  for (int bi = 0; bi < b; ++bi) {
    const float *g = grad_out + (size_t)bi * n * c;
    const int *id = idx + (size_t)bi * n * 3;
    const float *w = weight + (size_t)bi * n * 3;
    float *gp = grad_points + (size_t)bi * m * c;
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < n; ++j) {
        const float go = g[(size_t)l * n + j];
        gp[(size_t)l * m + id[j * 3 + 0]] += go * w[j * 3 + 0];
        gp[(size_t)l * m + id[j * 3 + 1]] += go * w[j * 3 + 1];
        gp[(size_t)l * m + id[j * 3 + 2]] += go * w[j * 3 + 2];
      }
  }
}

/*
 * pytorch3d 0.7.0 `knn_points` (third-party, NOT under /root/reference; pinned by
 * environment.yml:117).  Call sites: pointnet2_ops/pointnet2_utils.py:370,506.
 * Published semantics restated: for every p1[i], the K nearest points of p2 (only the first
 * lengths2[b] are valid; NULL = all), squared L2 in difference form, ascending; ties broken
 * by LOWER index (this repo's definition -- pytorch3d leaves it unspecified; parity unpinned).
 * Slots >= lengths2 keep dist 0 / idx 0 like pytorch3d's zero-initialised outputs.
 * idx is int64 (pytorch3d) -- callers cast to int32 for group_points.
 */
ORA_API void ora_knn_points(int b, int n1, int n2, int K, const float *p1, const float *p2,
                            const int64_t *lengths2, float *dists, int64_t *idx) {
  for (int bi = 0; bi < b; ++bi) {
    const float *a = p1 + (size_t)bi * n1 * 3;
    const float *q = p2 + (size_t)bi * n2 * 3;
    const int len2 = lengths2 ? (int)lengths2[bi] : n2;
    for (int i = 0; i < n1; ++i) {
      float *od = dists + ((size_t)bi * n1 + i) * K;
      int64_t *oi = idx + ((size_t)bi * n1 + i) * K;
      for (int k = 0; k < K; ++k) { od[k] = 0.0f; oi[k] = 0; }
      int cnt = 0;
      for (int j = 0; j < len2; ++j) {
        const float d = sqdist3(a[i * 3 + 0], a[i * 3 + 1], a[i * 3 + 2], q[j * 3 + 0], q[j * 3 + 1],
                                q[j * 3 + 2]);
        /* stable insertion: strict '<' so an equal distance never displaces an earlier index */
        int pos = cnt < K ? cnt : K;
        while (pos > 0 && d < od[pos - 1]) --pos;
        if (pos >= K) continue;
        const int last = cnt < K ? cnt : K - 1;
        for (int t = last; t > pos; --t) { od[t] = od[t - 1]; oi[t] = oi[t - 1]; }
        od[pos] = d; oi[pos] = j;
        if (cnt < K) ++cnt;
      }
    }
  }
}

/* pytorch3d `knn_gather` (pointnet2_utils.py:507): out[b,n,k,:] = x[b, idx[b,n,k], :] */
ORA_API void ora_knn_gather(int b, int n2, int u, int n1, int K, const float *x, const int64_t *idx,
                            float *out) {
  for (int bi = 0; bi < b; ++bi)
    for (int i = 0; i < n1; ++i)
      for (int k = 0; k < K; ++k) {
        const int64_t j = idx[((size_t)bi * n1 + i) * K + k];
        memcpy(out + (((size_t)bi * n1 + i) * K + k) * u, x + ((size_t)bi * n2 + j) * u,
               sizeof(float) * (size_t)u);
      }
}

/* pytorch3d 0.7.0 `sample_farthest_points` (third-party; call site pointnet2/models/point_upsample_decoder.py:178-180):
 * plain iterative farthest point sampling -- no near-origin skip, first index = start[b] (random in the reference,
 * hence parity there is distributional only), running min of squared distances, argmax with ties -> lowest index
 * (this repo's definition). */
ORA_API void ora_sample_farthest_points(int b, int n, int K, const float *points, const int *start, int *idx) {
  float *md = (float *)malloc(sizeof(float) * (size_t)n);
  for (int bi = 0; bi < b; ++bi) {
    const float *p = points + (size_t)bi * n * 3;
    int *o = idx + (size_t)bi * K;
    for (int k = 0; k < n; ++k) md[k] = 1e10f;
    int old = start ? start[bi] : 0;
    o[0] = old;
    for (int j = 1; j < K; ++j) {
      float best = -1.0f;
      int besti = 0;
      for (int k = 0; k < n; ++k) {
        const float d = sqdist3(p[k * 3], p[k * 3 + 1], p[k * 3 + 2], p[old * 3], p[old * 3 + 1], p[old * 3 + 2]);
        const float d2 = d < md[k] ? d : md[k];
        md[k] = d2;
        if (d2 > best) { best = d2; besti = k; }
      }
      old = besti;
      o[j] = old;
    }
  }
  free(md);
}
