"""Drop-in counterpart of the reference's native module `pointnet2_ops._ext`
(_ext-src/src/bindings.cpp:6-19): the same nine functions, argument order, dtypes, output
allocation (zeros where an op leaves elements unwritten -- ball_query, the grads, three_nn, FPS -- and the FPS scratch
1e10; the three forward gathers write every element, so their outputs are allocated uninitialised) and error behaviour (RuntimeError on non-contiguous /
wrong dtype / CPU tensors: _ext-src/include/utils.h:5-25, sampling.cpp:21-34), implemented on
hand-written gfx950 HIP kernels through the C-ABI in include/slide_hip.h.

Asynchronous on torch's current stream, no host sync, inputs borrowed and never mutated.
"""
import ctypes

import torch

from ._lib import check, lib, ptr, stream_of


def _chk_contig(x, name):
    if not x.is_contiguous():
        raise RuntimeError("%s must be a contiguous tensor" % name)


def _chk_float(x, name):
    if x.dtype != torch.float32:
        raise RuntimeError("%s must be a float tensor" % name)


def _chk_int(x, name):
    if x.dtype != torch.int32:
        raise RuntimeError("%s must be an int tensor" % name)


def _chk_cuda(x, name):
    if not x.is_cuda:
        raise RuntimeError("%s must be a CUDA tensor" % name)


def _need_gpu(x):
    if not x.is_cuda:
        raise RuntimeError("CPU not supported")  # sampling.cpp:34 et al.


def gather_points(points, idx):
    """sampling.cpp:15-38  points (B,C,N) f32, idx (B,M) i32 -> (B,C,M)"""
    _chk_contig(points, "points"); _chk_contig(idx, "idx"); _chk_float(points, "points"); _chk_int(idx, "idx")
    if points.is_cuda:
        _chk_cuda(idx, "idx")
    # the kernel writes every element: the reference's zero fill would only add a second pass over the output
    out = torch.empty((points.size(0), points.size(1), idx.size(1)), device=points.device, dtype=torch.float32)
    _need_gpu(points)
    check(lib().gather_points_kernel_wrapper(points.size(0), points.size(1), points.size(2), idx.size(1),
                                             ptr(points), ptr(idx), ptr(out), stream_of()), "gather_points")
    return out


def gather_points_grad(grad_out, idx, n):
    """sampling.cpp:40-65"""
    _chk_contig(grad_out, "grad_out"); _chk_contig(idx, "idx"); _chk_float(grad_out, "grad_out"); _chk_int(idx, "idx")
    if grad_out.is_cuda:
        _chk_cuda(idx, "idx")
    out = torch.zeros((grad_out.size(0), grad_out.size(1), n), device=grad_out.device, dtype=torch.float32)
    _need_gpu(grad_out)
    check(lib().gather_points_grad_kernel_wrapper(grad_out.size(0), grad_out.size(1), int(n), idx.size(1),
                                                  ptr(grad_out), ptr(idx), ptr(out), stream_of()),
          "gather_points_grad")
    return out


def furthest_point_sampling(points, nsamples):
    """sampling.cpp:66-87  points (B,N,3) -> (B,nsamples) i32"""
    _chk_contig(points, "points"); _chk_float(points, "points")
    out = torch.zeros((points.size(0), nsamples), device=points.device, dtype=torch.int32)
    tmp = torch.full((points.size(0), points.size(1)), 1e10, device=points.device, dtype=torch.float32)
    _need_gpu(points)
    check(lib().furthest_point_sampling_kernel_wrapper(points.size(0), points.size(1), int(nsamples), ptr(points),
                                                       ptr(tmp), ptr(out), stream_of()), "furthest_point_sampling")
    return out


def three_nn(unknown, known):
    """interpolate.cpp:14-40 -> [dist2 (B,n,3) f32, idx (B,n,3) i32]"""
    _chk_contig(unknown, "unknowns"); _chk_contig(known, "knows"); _chk_float(unknown, "unknowns")
    _chk_float(known, "knows")
    if unknown.is_cuda:
        _chk_cuda(known, "knows")
    idx = torch.zeros((unknown.size(0), unknown.size(1), 3), device=unknown.device, dtype=torch.int32)
    dist2 = torch.zeros((unknown.size(0), unknown.size(1), 3), device=unknown.device, dtype=torch.float32)
    _need_gpu(unknown)
    check(lib().three_nn_kernel_wrapper(unknown.size(0), unknown.size(1), known.size(1), ptr(unknown), ptr(known),
                                        ptr(dist2), ptr(idx), stream_of()), "three_nn")
    return [dist2, idx]


def three_interpolate(points, idx, weight):
    """interpolate.cpp:42-70  points (B,c,m), idx (B,n,3), weight (B,n,3) -> (B,c,n)"""
    for t, n in ((points, "points"), (idx, "idx"), (weight, "weight")):
        _chk_contig(t, n)
    _chk_float(points, "points"); _chk_int(idx, "idx"); _chk_float(weight, "weight")
    if points.is_cuda:
        _chk_cuda(idx, "idx"); _chk_cuda(weight, "weight")
    out = torch.empty((points.size(0), points.size(1), idx.size(1)), device=points.device, dtype=torch.float32)
    _need_gpu(points)
    check(lib().three_interpolate_kernel_wrapper(points.size(0), points.size(1), points.size(2), idx.size(1),
                                                 ptr(points), ptr(idx), ptr(weight), ptr(out), stream_of()),
          "three_interpolate")
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    """interpolate.cpp:71-99"""
    for t, n in ((grad_out, "grad_out"), (idx, "idx"), (weight, "weight")):
        _chk_contig(t, n)
    _chk_float(grad_out, "grad_out"); _chk_int(idx, "idx"); _chk_float(weight, "weight")
    if grad_out.is_cuda:
        _chk_cuda(idx, "idx"); _chk_cuda(weight, "weight")
    out = torch.zeros((grad_out.size(0), grad_out.size(1), m), device=grad_out.device, dtype=torch.float32)
    _need_gpu(grad_out)
    check(lib().three_interpolate_grad_kernel_wrapper(grad_out.size(0), grad_out.size(1), grad_out.size(2), int(m),
                                                      ptr(grad_out), ptr(idx), ptr(weight), ptr(out), stream_of()),
          "three_interpolate_grad")
    return out


def ball_query(new_xyz, xyz, radius, nsample):
    """ball_query.cpp:10-38 -> (idx (B,m,nsample) i32, counts (B,m) i32)"""
    _chk_contig(new_xyz, "new_xyz"); _chk_contig(xyz, "xyz"); _chk_float(new_xyz, "new_xyz"); _chk_float(xyz, "xyz")
    if new_xyz.is_cuda:
        _chk_cuda(xyz, "xyz")
    idx = torch.zeros((new_xyz.size(0), new_xyz.size(1), nsample), device=new_xyz.device, dtype=torch.int32)
    counts = torch.zeros((new_xyz.size(0), new_xyz.size(1)), device=new_xyz.device, dtype=torch.int32)
    _need_gpu(new_xyz)
    check(lib().query_ball_point_kernel_wrapper(xyz.size(0), xyz.size(1), new_xyz.size(1), ctypes.c_float(radius),
                                                int(nsample), ptr(new_xyz), ptr(xyz), ptr(idx), ptr(counts),
                                                stream_of()), "ball_query")
    return idx, counts


def group_points(points, idx):
    """group_points.cpp:12-36  points (B,C,N), idx (B,np,ns) -> (B,C,np,ns)"""
    _chk_contig(points, "points"); _chk_contig(idx, "idx"); _chk_float(points, "points"); _chk_int(idx, "idx")
    if points.is_cuda:
        _chk_cuda(idx, "idx")
    out = torch.empty((points.size(0), points.size(1), idx.size(1), idx.size(2)), device=points.device,
                      dtype=torch.float32)
    _need_gpu(points)
    check(lib().group_points_kernel_wrapper(points.size(0), points.size(1), points.size(2), idx.size(1), idx.size(2),
                                            ptr(points), ptr(idx), ptr(out), stream_of()), "group_points")
    return out


def group_points_grad(grad_out, idx, n):
    """group_points.cpp:38-62"""
    _chk_contig(grad_out, "grad_out"); _chk_contig(idx, "idx"); _chk_float(grad_out, "grad_out"); _chk_int(idx, "idx")
    if grad_out.is_cuda:
        _chk_cuda(idx, "idx")
    out = torch.zeros((grad_out.size(0), grad_out.size(1), n), device=grad_out.device, dtype=torch.float32)
    _need_gpu(grad_out)
    check(lib().group_points_grad_kernel_wrapper(grad_out.size(0), grad_out.size(1), int(n), idx.size(1), idx.size(2),
                                                 ptr(grad_out), ptr(idx), ptr(out), stream_of()), "group_points_grad")
    return out


# ------------------------------------------------------------------ pytorch3d.ops.knn counterpart
def knn_points(p1, p2, K, lengths2=None):
    """pytorch3d knn_points (call sites pointnet2_ops/pointnet2_utils.py:370,506):
    -> (dists (B,N1,K) f32 ascending squared L2, idx (B,N1,K) int64)"""
    p1 = p1.contiguous(); p2 = p2.contiguous()
    _chk_float(p1, "p1"); _chk_float(p2, "p2"); _need_gpu(p1); _chk_cuda(p2, "p2")
    B, N1, _ = p1.shape
    dists = torch.zeros((B, N1, K), device=p1.device, dtype=torch.float32)
    idx = torch.zeros((B, N1, K), device=p1.device, dtype=torch.int64)
    lp = None
    if lengths2 is not None:
        lengths2 = lengths2.to(device=p1.device, dtype=torch.int64).contiguous()
        lp = ptr(lengths2)
    check(lib().slide_knn_points(B, N1, p2.size(1), int(K), ptr(p1), ptr(p2), lp, ptr(dists), ptr(idx), stream_of()),
          "knn_points")
    return dists, idx


def knn_gather(x, idx):
    """pytorch3d knn_gather: x (B,N2,U), idx (B,N1,K) int64 -> (B,N1,K,U)"""
    x = x.contiguous(); idx = idx.contiguous()
    _chk_float(x, "x"); _need_gpu(x)
    if idx.dtype != torch.int64:
        raise RuntimeError("idx must be an int64 tensor")
    B, N2, U = x.shape
    _, N1, K = idx.shape
    out = torch.empty((B, N1, K, U), device=x.device, dtype=torch.float32)
    check(lib().slide_knn_gather(B, N2, U, N1, K, ptr(x), ptr(idx), ptr(out), stream_of()), "knn_gather")
    return out


def gather_rows(points, idx):
    """row-layout counterpart of gather_points: points (B,N,C) float32, idx (B,M) int32 -> (B,M,C).  Moves only the gathered rows
    (gather_points on the reference's (B,C,N) layout pays a 64-byte sector per gathered element)."""
    points = points.contiguous(); idx = idx.contiguous()
    _chk_float(points, "points"); _need_gpu(points)
    if idx.dtype != torch.int32:
        raise RuntimeError("idx must be an int tensor")
    B, N, C = points.shape
    M = idx.shape[1]
    out = torch.empty((B, M, C), device=points.device, dtype=torch.float32)
    check(lib().slide_gather_rows(B, N, M, C, ptr(points), ptr(idx), ptr(out), stream_of()), "gather_rows")
    return out


def sample_farthest_points(points, lengths=None, K=50, random_start_point=False, start_idx=None):
    """pytorch3d.ops.sample_farthest_points counterpart (call site point_upsample_decoder.py:178-180):
    -> (selected points (B,K,C), idx int64 (B,K)).  FPS on points[..., :3]; `random_start_point` draws the first index
    from torch's device generator (the reference's results are therefore only distributionally reproducible)."""
    assert lengths is None
    _need_gpu(points)
    xyz = points[:, :, 0:3].contiguous().float()
    B, N, _ = xyz.shape
    if start_idx is None and random_start_point:
        start_idx = torch.randint(0, N, (B,), device=points.device)
    sp = None
    if start_idx is not None:
        # any non-negative integer is a valid start: it is taken modulo N (callers hand one per-shape draw to every level of a
        # decoder, whose candidate counts differ -- ADVICE r4); no device sync, never out of bounds
        start_idx = torch.remainder(start_idx.to(device=points.device, dtype=torch.int64), N).to(torch.int32).contiguous()
        sp = ptr(start_idx)
    idx = torch.zeros((B, K), device=points.device, dtype=torch.int32)
    tmp = torch.full((B, N), 1e10, device=points.device, dtype=torch.float32)
    check(lib().slide_sample_farthest_points(B, N, int(K), ptr(xyz), sp, ptr(tmp), ptr(idx), stream_of()),
          "sample_farthest_points")
    idx = idx.long()
    return torch.gather(points, 1, idx.unsqueeze(-1).expand(-1, -1, points.shape[2])), idx
