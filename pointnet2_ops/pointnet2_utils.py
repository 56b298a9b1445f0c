"""`pointnet2_ops.pointnet2_utils` on HIP kernels: the same functions / classes as the reference module
(pointnet2_ops_lib/pointnet2_ops/pointnet2_utils.py): six autograd wrappers (:62-304), QueryAndGroup (:307-448),
GroupAll (:451-494), group_knn (:497-524), count_to_mask / average_feature (:36-60).  `knn` stands in for
`pytorch3d.ops.knn` (knn_points / knn_gather), which the reference imports from un-vendored pytorch3d 0.7.0."""
import collections
import types

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

import pointnet2_ops._ext as _ext
from slide_amd import _ext as _hip
from slide_amd import rows as _rows

_KNN = collections.namedtuple("KNN", "dists idx knn")


def _knn_points(p1, p2, lengths1=None, lengths2=None, norm=2, K=1, version=-1, return_nn=False, return_sorted=True):
    d, i = _hip.knn_points(p1, p2, K, lengths2)
    nn_ = _hip.knn_gather(p2.contiguous(), i) if return_nn else None
    return _KNN(d, i, nn_)


knn = types.SimpleNamespace(knn_points=_knn_points, knn_gather=lambda x, idx, lengths=None: _hip.knn_gather(x, idx))


def count_to_mask(count, K):
    """(B, np) neighbour counts -> (B, np, K) bool: slot k holds a real neighbour (reference :36-44)"""
    return torch.arange(K, device=count.device, dtype=count.dtype).expand(count.shape + (K,)) < count.unsqueeze(-1)


def average_feature(feature, count, K):
    """mean of (B, C, np, K) over the first `count` neighbour slots (all K for count == 'all'; reference :47-60)"""
    counts = None if isinstance(count, str) else count
    return _rows.to_ncx(_rows.pool(_rows.from_ncx(feature, half=False), K, _rows.POOL_AVG, counts))


class FurthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz, npoint):
        out = _ext.furthest_point_sampling(xyz, npoint)
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        return ()


furthest_point_sample = FurthestPointSampling.apply


class GatherOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        ctx.save_for_backward(idx, features)
        return _ext.gather_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        idx, features = ctx.saved_tensors
        return _ext.gather_points_grad(grad_out.contiguous(), idx, features.size(2)), None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown, known):
        dist2, idx = _ext.three_nn(unknown, known)
        dist = torch.sqrt(dist2)
        ctx.mark_non_differentiable(dist, idx)
        return dist, idx

    @staticmethod
    def backward(ctx, grad_dist, grad_idx):
        return ()


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, features, idx, weight):
        ctx.save_for_backward(idx, weight, features)
        return _ext.three_interpolate(features, idx, weight)

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight, features = ctx.saved_tensors
        g = _ext.three_interpolate_grad(grad_out.contiguous(), idx, weight, features.size(2))
        return g, torch.zeros_like(idx), torch.zeros_like(weight)


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        ctx.save_for_backward(idx, features)
        return _ext.group_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        idx, features = ctx.saved_tensors
        return _ext.group_points_grad(grad_out.contiguous(), idx, features.size(2)), torch.zeros_like(idx)


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        output, counts = _ext.ball_query(new_xyz, xyz, radius, nsample)
        ctx.mark_non_differentiable(output)
        return output, counts

    @staticmethod
    def backward(ctx, grad_out, grad_counts=None):
        return ()


ball_query = BallQuery.apply


class QueryAndGroup(nn.Module):
    """Neighbour search (ball query or kNN) + grouped-feature assembly (reference :307-448).

    The search is a HIP kernel; the assembly -- gather the neighbours' features, append relative / absolute / centre
    coordinates, substitute the centre for empty balls -- is ONE row-major kernel (slide_amd.rows.group) whose
    [B * npoint * K][C + 3..9] matrix is what the row-major module path consumes directly (`rows`); `forward` lays it
    out as the reference's (B, C + 3..9, npoint, K) tensor."""

    def __init__(self, radius, nsample, use_xyz=True, include_abs_coordinate=False, include_center_coordinate=False,
                 neighbor_def="radius"):
        super().__init__()
        assert neighbor_def in ("radius", "nn")
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz
        self.include_abs_coordinate = include_abs_coordinate
        self.include_center_coordinate = include_center_coordinate
        self.neighbor_def = neighbor_def
        self.neighbor_stats = None
        self.neighbor_num_quantile = None
        self.quantile = torch.tensor([0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1])

    def search(self, xyz, new_xyz, length=None):
        """-> (idx (B, npoint, K) int32 [ball] / int64 [kNN], counts (B, npoint))"""
        if self.neighbor_def == "nn":
            K = min(self.nsample, xyz.shape[1])
            idx = knn.knn_points(new_xyz, xyz, K=K, lengths2=length).idx
            counts = torch.full(idx.shape[:2], float(K), device=xyz.device)
            if length is not None:
                counts = torch.minimum(counts, length.unsqueeze(1))
            return idx, counts
        if length is not None:
            raise Exception("radius neighbor definition has not supported point clouds with different lengths")
        return ball_query(self.radius, self.nsample, xyz, new_xyz)

    def layout_flags(self):
        if not self.use_xyz:
            return _rows.GROUP_NO_XYZ
        return (_rows.GROUP_ABS if self.include_abs_coordinate else 0) | (_rows.GROUP_CENTER if self.include_center_coordinate else 0)

    def rows(self, xyz, new_xyz, feat=None, subset=True, record_neighbor_stats=False, length=None, half=None):
        """feat: slide_amd.rows.Rows [B * N] or None -> (grouped Rows [B * npoint * K], counts, K)"""
        if feat is None and not self.use_xyz:
            raise AssertionError("Cannot have not features and not use xyz as a feature!")
        idx, counts = self.search(xyz, new_xyz, length)
        if record_neighbor_stats:
            c = counts.float()
            self.neighbor_stats = torch.stack([c.min(), c.mean(), c.max()])
            self.neighbor_num_quantile = torch.quantile(c, self.quantile.to(c.device)).long()
        stand_in = counts if (self.neighbor_def == "radius" and not subset) else None  # empty balls: the centre itself
        grouped = _rows.group(xyz[..., :3], new_xyz[..., :3], feat, idx, self.layout_flags(), empty_counts=stand_in, half=half)
        return grouped, counts, idx.shape[2]

    def forward(self, xyz, new_xyz, features=None, subset=True, record_neighbor_stats=False, return_counts=False,
                length=None):
        feat = _rows.from_ncx(features, half=False) if features is not None else None
        grouped, counts, K = self.rows(xyz, new_xyz, feat, subset, record_neighbor_stats, length, half=False)
        out = _rows.to_ncx(grouped, (new_xyz.shape[1], K))
        return (out, counts) if return_counts else out


class GroupAll(nn.Module):
    """one group holding every point (reference :451-494): (B, C [+ 3], 1, N)"""

    def __init__(self, use_xyz=True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz, new_xyz, features=None):
        coords = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            return coords
        feats = features.unsqueeze(2)
        return torch.cat([feats, coords], dim=1) if self.use_xyz else feats


def group_knn_rows(x, y, feat_at_y, K):
    """kNN feature propagation input on Rows: feat_at_y Rows [B * Ny] -> Rows [B * Nx * K] with the channel layout
    [feat | d2 | w | abs | rel | centre] (reference :497-524; w = normalised inverse squared distance)"""
    d2, idx, _ = knn.knn_points(x, y, K=K)
    return _rows.group(y, x, feat_at_y, idx, _rows.GROUP_FP, d2=d2)


def group_knn(x, y, features_at_y, K, transpose=False):
    """K nearest neighbours of every x in y: (B, Nx, K, C + 11), or (B, C + 11, Nx, K) with channel-first features
    (transpose=True), as the reference returns them"""
    feat = _rows.from_ncx(features_at_y, half=False) if transpose else _rows.from_points(features_at_y, half=False)
    g = group_knn_rows(x, y, feat, K)
    if transpose:
        return _rows.to_ncx(g, (x.shape[1], K))
    return _rows.to_points(g).reshape(x.shape[0], x.shape[1], K, g.C)
