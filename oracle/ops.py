"""oracle/ops.py -- TEST INFRASTRUCTURE ONLY.

numpy front-end of oracle/libslide_oracle.so (the plain-C CPU restatement of the reference's
`pointnet2_ops._ext` kernels and of pytorch3d's knn_points / knn_gather).  Output allocation
(zeros / 1e10 fill) follows the reference host wrappers:
  _ext-src/src/sampling.cpp:15-87, ball_query.cpp:10-38, group_points.cpp:12-62,
  interpolate.cpp:14-99.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libslide_oracle.so")


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(
            os.path.join(_HERE, "ops_cpu.c")):
        subprocess.check_call(["make", "-s", "-C", _HERE, "clean", "all"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))


def _l(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))


def opt_n_threads(w):
    return int(lib().ora_opt_n_threads(int(w)))


def gather_points(points, idx):
    points, pp = _f(points); idx, ip = _i(idx)
    b, c, n = points.shape; m = idx.shape[1]
    out = np.zeros((b, c, m), np.float32); _, op = _f(out)
    lib().ora_gather_points(b, c, n, m, pp, ip, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return out


def gather_points_grad(grad_out, idx, n):
    grad_out, gp = _f(grad_out); idx, ip = _i(idx)
    b, c, m = grad_out.shape
    out = np.zeros((b, c, n), np.float32)
    lib().ora_gather_points_grad(b, c, n, m, gp, ip, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return out


def furthest_point_sampling(points, nsamples, return_temp=False):
    points, pp = _f(points)
    b, n, _ = points.shape
    out = np.zeros((b, nsamples), np.int32)
    temp = np.full((b, n), 1e10, np.float32)
    lib().ora_furthest_point_sampling(b, n, int(nsamples), pp,
                                      temp.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                      out.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    return (out, temp) if return_temp else out


def ball_query(new_xyz, xyz, radius, nsample):
    new_xyz, qp = _f(new_xyz); xyz, xp = _f(xyz)
    b, m, _ = new_xyz.shape; n = xyz.shape[1]
    idx = np.zeros((b, m, nsample), np.int32); counts = np.zeros((b, m), np.int32)
    lib().ora_ball_query(b, n, m, ctypes.c_float(radius), int(nsample), qp, xp,
                         idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                         counts.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    return idx, counts


def group_points(points, idx):
    points, pp = _f(points); idx, ip = _i(idx)
    b, c, n = points.shape; _, npoints, nsample = idx.shape
    out = np.zeros((b, c, npoints, nsample), np.float32)
    lib().ora_group_points(b, c, n, npoints, nsample, pp, ip,
                           out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return out


def group_points_grad(grad_out, idx, n):
    grad_out, gp = _f(grad_out); idx, ip = _i(idx)
    b, c, npoints, nsample = grad_out.shape
    out = np.zeros((b, c, n), np.float32)
    lib().ora_group_points_grad(b, c, n, npoints, nsample, gp, ip,
                                out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return out


def three_nn(unknown, known):
    unknown, up = _f(unknown); known, kp = _f(known)
    b, n, _ = unknown.shape; m = known.shape[1]
    dist2 = np.zeros((b, n, 3), np.float32); idx = np.zeros((b, n, 3), np.int32)
    lib().ora_three_nn(b, n, m, up, kp, dist2.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                       idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    return dist2, idx


def three_interpolate(points, idx, weight):
    points, pp = _f(points); idx, ip = _i(idx); weight, wp = _f(weight)
    b, c, m = points.shape; n = idx.shape[1]
    out = np.zeros((b, c, n), np.float32)
    lib().ora_three_interpolate(b, c, m, n, pp, ip, wp,
                                out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    grad_out, gp = _f(grad_out); idx, ip = _i(idx); weight, wp = _f(weight)
    b, c, n = grad_out.shape
    out = np.zeros((b, c, m), np.float32)
    lib().ora_three_interpolate_grad(b, c, n, m, gp, ip, wp,
                                     out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return out


def knn_points(p1, p2, K, lengths2=None):
    """-> (dists f32 [B,N1,K] ascending squared L2, idx int64 [B,N1,K])."""
    p1, ap = _f(p1); p2, bp = _f(p2)
    b, n1, _ = p1.shape; n2 = p2.shape[1]
    dists = np.zeros((b, n1, K), np.float32); idx = np.zeros((b, n1, K), np.int64)
    lp = None
    if lengths2 is not None:
        lengths2, lp = _l(lengths2)
    lib().ora_knn_points(b, n1, n2, int(K), ap, bp, lp,
                         dists.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                         idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    return dists, idx


def knn_gather(x, idx):
    x, xp = _f(x); idx, ip = _l(idx)
    b, n2, u = x.shape; _, n1, K = idx.shape
    out = np.zeros((b, n1, K, u), np.float32)
    lib().ora_knn_gather(b, n2, u, n1, K, xp, ip, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return out


def gather_rows(points, idx):
    """row-layout counterpart of gather_points (the build's slide_gather_rows): points (B,N,C), idx (B,M) -> (B,M,C);
    by definition gather_points (sampling_gpu.cu:8-20) on the transposed table, transposed back"""
    p = np.ascontiguousarray(points, dtype=np.float32)
    return np.ascontiguousarray(gather_points(np.ascontiguousarray(p.transpose(0, 2, 1)), idx).transpose(0, 2, 1))


def sample_farthest_points(points, K, start_idx=None):
    """-> (selected points (B,K,C), idx int64 (B,K)); FPS runs on the first three channels"""
    pts3, pp = _f(points[:, :, 0:3])
    b, n, _ = pts3.shape
    idx = np.zeros((b, K), np.int32)
    sp = None
    if start_idx is not None:
        start_idx, sp = _i(start_idx)
    lib().ora_sample_farthest_points(b, n, int(K), pp, sp, idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    idx = idx.astype(np.int64)
    return np.take_along_axis(np.asarray(points), idx[..., None], axis=1), idx
