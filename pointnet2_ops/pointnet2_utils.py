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

_KNN = collections.namedtuple("KNN", "dists idx knn")


def _knn_points(p1, p2, lengths1=None, lengths2=None, norm=2, K=1, version=-1, return_nn=False, return_sorted=True):
    d, i = _hip.knn_points(p1, p2, K, lengths2)
    nn_ = _hip.knn_gather(p2.contiguous(), i) if return_nn else None
    return _KNN(d, i, nn_)


knn = types.SimpleNamespace(knn_points=_knn_points, knn_gather=lambda x, idx, lengths=None: _hip.knn_gather(x, idx))


def count_to_mask(count, K):
    mask = torch.arange(K, device=count.device, dtype=count.dtype)
    B, npoint = count.size()
    mask = mask.repeat(B, npoint).view(B, npoint, -1)
    return mask < count.unsqueeze(-1)


def average_feature(feature, count, K):
    if isinstance(count, str) and count == "all":
        return F.avg_pool2d(feature, kernel_size=[1, feature.size(3)]).squeeze(-1)
    count = torch.clamp(count, min=1)
    mask = count_to_mask(count, K).unsqueeze(1)
    return (feature * mask).sum(dim=-1) / count.unsqueeze(1)


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
    """radius (ball query) or nn (kNN) grouping + coordinate feature assembly (reference :307-448)"""

    def __init__(self, radius, nsample, use_xyz=True, include_abs_coordinate=False, include_center_coordinate=False,
                 neighbor_def="radius"):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz
        self.include_abs_coordinate = include_abs_coordinate
        self.include_center_coordinate = include_center_coordinate
        self.neighbor_stats = None
        self.quantile = torch.tensor([0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1])
        self.neighbor_num_quantile = None
        self.neighbor_def = neighbor_def
        assert neighbor_def in ("radius", "nn")

    def forward(self, xyz, new_xyz, features=None, subset=True, record_neighbor_stats=False, return_counts=False,
                length=None):
        if self.neighbor_def == "radius":
            if length is not None:
                raise Exception("radius neighbor definition has not supported point clouds with different lengths")
            idx, counts = ball_query(self.radius, self.nsample, xyz, new_xyz)
        else:
            num_neighbors = min(self.nsample, xyz.shape[1])
            _, idx, _ = knn.knn_points(new_xyz, xyz, K=num_neighbors, lengths2=length)
            idx = idx.int()
            B, npoint, K = idx.size()
            counts = torch.ones(B, npoint, device=new_xyz.device) * K
            if length is not None:
                counts = torch.minimum(counts, length.unsqueeze(1))
        xyz_trans = xyz.transpose(1, 2).contiguous()
        abs_xyz = grouping_operation(xyz_trans, idx)
        new_xyz_trans = new_xyz.transpose(1, 2).unsqueeze(-1)
        patch = (not subset) and self.neighbor_def == "radius"
        if patch:  # empty balls: the centre itself stands in as the only neighbour, with zero features
            have_neigh = (counts > 0).float().unsqueeze(1).unsqueeze(-1).detach()
            no_neigh = 1 - have_neigh
            abs_xyz = have_neigh * abs_xyz + no_neigh * new_xyz_trans
        relative_xyz = abs_xyz - new_xyz_trans
        grouped_xyz = torch.cat([relative_xyz, abs_xyz], dim=1) if self.include_abs_coordinate else relative_xyz
        if self.include_center_coordinate:
            grouped_xyz = torch.cat([grouped_xyz, new_xyz_trans.expand(-1, -1, -1, grouped_xyz.shape[3])], dim=1)
        if features is not None:
            grouped_features = grouping_operation(features, idx)
            if patch:
                grouped_features = have_neigh * grouped_features
            new_features = torch.cat([grouped_features, grouped_xyz], dim=1) if self.use_xyz else grouped_features
        else:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            new_features = grouped_xyz
        if record_neighbor_stats:
            with torch.no_grad():
                c = counts.float()
                self.neighbor_stats = torch.stack([c.min(), c.mean(), c.max()])
                self.neighbor_num_quantile = torch.quantile(c, self.quantile.to(c.device)).long()
        return (new_features, counts) if return_counts else new_features


class GroupAll(nn.Module):
    def __init__(self, use_xyz=True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            return grouped_xyz
        grouped_features = features.unsqueeze(2)
        return torch.cat([grouped_features, grouped_xyz], dim=1) if self.use_xyz else grouped_features


def group_knn(x, y, features_at_y, K, transpose=False):
    """K nearest neighbours of every x in y with [feats, d2, w, abs, rel, centre] per neighbour (reference :497-524)"""
    feats = features_at_y.transpose(1, 2).contiguous() if transpose else features_at_y
    dist, idx, nn_abs = knn.knn_points(x, y, K=K, return_nn=True)
    nbr = knn.knn_gather(feats, idx)
    x_repeat = x.unsqueeze(2).repeat(1, 1, K, 1)
    rel = nn_abs - x_repeat
    dist = dist.unsqueeze(3)
    recip = 1.0 / (dist + 1e-8)
    weight = recip / torch.sum(recip, dim=2, keepdim=True)
    new_features = torch.cat([nbr, dist, weight, nn_abs, rel, x_repeat], dim=3)
    if transpose:
        new_features = new_features.transpose(2, 3).transpose(1, 2)
    return new_features
