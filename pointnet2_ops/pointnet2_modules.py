"""`pointnet2_ops.pointnet2_modules` on HIP kernels: same classes, constructor signatures and state-dict names as the
reference (pointnet2_ops_lib/pointnet2_ops/pointnet2_modules.py): Swish (:17-22), MyGroupNorm (:24-42),
build_shared_mlp (:44-69), Mlp_plus_t_emb (:71-176), pooling_features (:179-211), PointnetSAModule[MSG] (:213-462),
PointnetFPModule (:465-588), FeatureMapModule (:591-663), PointnetKnnFPModule (:666-873).

This is the GENERAL (any N / K / radius-or-nn) inference path: neighbour search, FPS and gathers are the HIP `_ext`
kernels, 1x1 convolutions / linears the HIP MFMA GEMM, GroupNorm a HIP kernel; concatenation, ReLU and the softmax
glue are torch tensor ops.  The latent-DDPM configurations run on the fused engine instead (slide_amd.engine)."""
import copy
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from pointnet2_ops import pointnet2_utils
from pointnet2_ops.attention import AttentionModule, GlobalAttentionModule
from slide_amd.nn_ops import HipConv1x1, HipGroupNorm, HipLinear


def swish(x):
    return x * torch.sigmoid(x)


class Swish(nn.Module):
    def forward(self, x):
        return swish(x)


class MyGroupNorm(nn.Module):
    def __init__(self, num_groups, num_channels, fused_relu=False):
        super().__init__()
        assert num_channels >= num_groups
        self.num_channels = num_channels - num_channels % num_groups
        self.num_groups = num_groups
        self.fused_relu = fused_relu  # the ReLU that follows in the reference's Sequential, applied by the same kernel
        self.group_norm = HipGroupNorm(self.num_groups, self.num_channels)

    def forward(self, x):
        return self.group_norm(x, relu=self.fused_relu)


def _act(activation):
    assert activation in ["relu", "swish"]
    return nn.ReLU(True) if activation == "relu" else Swish()


def build_shared_mlp(mlp_spec: List[int], bn: bool = True, bn_first: bool = False, bias: bool = False,
                     activation: str = "relu"):
    layers = []
    for i in range(1, len(mlp_spec)):
        if bn_first:
            if bn:
                layers.append(MyGroupNorm(min(32, mlp_spec[i - 1]), mlp_spec[i - 1]))
            layers.append(_act(activation))
        layers.append(HipConv1x1(mlp_spec[i - 1], mlp_spec[i], bias=bias))
        if not bn_first:
            fuse = bn and activation == "relu"
            if bn:
                layers.append(MyGroupNorm(min(32, mlp_spec[i]), mlp_spec[i], fused_relu=fuse))
            # (the Sequential keeps the reference's indices: the activation slot holds no parameters)
            layers.append(nn.Identity() if fuse else _act(activation))
    return nn.Sequential(*layers)


class Mlp_plus_t_emb(nn.Module):
    def __init__(self, mlp_spec, bn, t_dim=128, include_t=True, bn_first=False, bias=False, first_conv=False,
                 first_conv_in_channel=0, res_connect=False, include_condition=False, condition_dim=128,
                 include_second_condition=False, second_condition_dim=128, activation="relu"):
        super().__init__()
        self.include_t = include_t
        if include_t:
            self.fc = HipLinear(t_dim, mlp_spec[1])
        self.include_condition = include_condition
        if include_condition:
            self.fc_condition = HipLinear(condition_dim, mlp_spec[2])
        self.include_second_condition = include_second_condition
        if include_second_condition:
            self.fc_second_condition = HipLinear(second_condition_dim, mlp_spec[-1])
        self.first_conv_bool = first_conv
        if first_conv:
            self.first_conv = HipConv1x1(first_conv_in_channel, mlp_spec[0], bias=bias)
        self.res_connect_bool = res_connect
        if res_connect:
            self.res_connect = None if mlp_spec[0] == mlp_spec[-1] else HipConv1x1(mlp_spec[0], mlp_spec[-1], bias=bias)
        assert len(mlp_spec) >= 3
        if include_second_condition:
            assert len(mlp_spec) >= 4
        self.first_mlp = build_shared_mlp(mlp_spec[0:2], bn, bn_first=bn_first, bias=bias, activation=activation)
        self.second_mlp = build_shared_mlp(mlp_spec[1:3], bn, bn_first=bn_first, bias=bias, activation=activation)
        self.rest_mlp = (build_shared_mlp(mlp_spec[2:], bn, bn_first=bn_first, bias=bias, activation=activation)
                         if len(mlp_spec) > 3 else None)

    def forward(self, feature, t_emb=None, condition_emb=None, second_condition_emb=None):
        if self.first_conv_bool:
            feature = self.first_conv(feature)
        h = self.first_mlp(feature)
        if self.include_t:
            if t_emb is None:
                raise Exception("Should pass t_emb to the forward function")
            h = h + self.fc(t_emb).unsqueeze(2).unsqueeze(3)
        elif t_emb is not None:
            raise Exception("This module does not include t but t_emb is given")
        h = self.second_mlp(h)
        if self.include_condition:
            if condition_emb is None:
                raise Exception("Should pass condition_emb to the forward function")
            h = h + self.fc_condition(condition_emb).unsqueeze(2).unsqueeze(3)
        elif condition_emb is not None:
            raise Exception("This module does not include condition but condition_emb is given")
        if self.rest_mlp is not None:
            h = self.rest_mlp(h)
        if self.include_second_condition:
            if second_condition_emb is None:
                raise Exception("Should pass second_condition_emb to the forward function")
            h = h + self.fc_second_condition(second_condition_emb).unsqueeze(2).unsqueeze(3)
        elif second_condition_emb is not None:
            raise Exception("This module does not include condition but condition_emb is given")
        if self.res_connect_bool:
            h = h + (self.res_connect(feature) if self.res_connect is not None else feature)
        return h


def pooling_features(feature, count=None, pooling="max"):
    assert pooling in ["max", "avg", "avg_max", "max_avg"]
    K = feature.size(3)
    if pooling == "max":
        return F.max_pool2d(feature, kernel_size=[1, K]).squeeze(-1)
    if pooling == "avg":
        return pointnet2_utils.average_feature(feature, count, K)
    half_C = int(feature.shape[1] / 2)
    mx = F.max_pool2d(feature[:, 0:half_C], kernel_size=[1, K]).squeeze(-1)
    return torch.cat([mx, pointnet2_utils.average_feature(feature[:, half_C:], count, K)], dim=1)


def _xyz_extra(use_xyz, include_abs_coordinate, include_center_coordinate):
    return (3 + (3 if include_abs_coordinate else 0) + (3 if include_center_coordinate else 0)) if use_xyz else 0


class _PointnetSAModuleBase(nn.Module):
    def __init__(self):
        super().__init__()
        self.npoint = None
        self.groupers = None
        self.mlps = None

    def forward(self, xyz, features, t_emb=None, condition_emb=None, second_condition_emb=None, subset=True,
                record_neighbor_stats=False, pooling="max", length=None):
        new_features_list = []
        xyz_flipped = xyz.transpose(1, 2).contiguous()
        assert self.npoint is not None
        if xyz.shape[1] <= self.npoint:
            new_xyz = xyz
            if self.use_attention_module:
                new_xyz_feat = features
        else:
            fidx = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
            new_xyz = pointnet2_utils.gather_operation(xyz_flipped, fidx).transpose(1, 2).contiguous()
            if self.use_attention_module:
                new_xyz_feat = pointnet2_utils.gather_operation(features, fidx)
        for i in range(len(self.groupers)):
            grouped_features, count = self.groupers[i](xyz, new_xyz, features, subset=subset,
                                                       record_neighbor_stats=record_neighbor_stats, return_counts=True,
                                                       length=length)
            out_features = self.mlps[i](grouped_features, t_emb=t_emb if self.include_t else None,
                                        condition_emb=condition_emb if self.include_condition else None,
                                        second_condition_emb=second_condition_emb if self.include_second_condition else None)
            if self.use_attention_module:
                new_features = self.attention_modules[i](new_xyz_feat, grouped_features, out_features, count)
            else:
                new_features = pooling_features(out_features, count=count, pooling=pooling)
            if self.use_global_attention_module:
                new_features = torch.cat([new_features, new_xyz.transpose(1, 2)], dim=1)
                new_features = self.global_attention_modules[i](new_features)
            new_features_list.append(new_features)
        return new_xyz, torch.cat(new_features_list, dim=1)


class PointnetSAModuleMSG(_PointnetSAModuleBase):
    def __init__(self, npoint, radii, nsamples, mlps, bn=True, use_xyz=True, t_dim=128, include_t=False,
                 include_abs_coordinate=False, include_center_coordinate=False, bn_first=False, bias=False, first_conv=False,
                 first_conv_in_channel=0, res_connect=False, include_condition=False, condition_dim=128,
                 include_second_condition=False, second_condition_dim=128, neighbor_def="radius", activation="relu",
                 attention_setting=None, global_attention_setting=None):
        super().__init__()
        self.include_t, self.t_dim = include_t, t_dim
        self.include_condition, self.condition_dim = include_condition, condition_dim
        self.include_second_condition, self.second_condition_dim = include_second_condition, second_condition_dim
        assert len(radii) == len(nsamples) == len(mlps)
        self.npoint = npoint
        self.groupers, self.mlps = nn.ModuleList(), nn.ModuleList()
        self.use_attention_module = bool(attention_setting and attention_setting["use_attention_module"])
        self.attention_modules = nn.ModuleList() if self.use_attention_module else None
        self.use_global_attention_module = bool(global_attention_setting and
                                                global_attention_setting["use_global_attention_module"])
        self.global_attention_modules = nn.ModuleList() if self.use_global_attention_module else None
        extra = _xyz_extra(use_xyz, include_abs_coordinate, include_center_coordinate)
        for i in range(len(radii)):
            self.groupers.append(
                pointnet2_utils.QueryAndGroup(radii[i], nsamples[i], use_xyz=use_xyz,
                                              include_abs_coordinate=include_abs_coordinate,
                                              include_center_coordinate=include_center_coordinate, neighbor_def=neighbor_def)
                if npoint is not None else pointnet2_utils.GroupAll(use_xyz))
            mlp_spec = mlps[i]
            ori_first_conv_in_channel = copy.deepcopy(first_conv_in_channel)
            ori_mlp_spec0 = copy.deepcopy(mlp_spec[0])
            if first_conv:
                first_conv_in_channel += extra
            else:
                mlp_spec[0] += extra
            self.mlps.append(Mlp_plus_t_emb(mlp_spec, bn, t_dim=self.t_dim, include_t=include_t, bn_first=bn_first, bias=bias,
                                            first_conv=first_conv, first_conv_in_channel=first_conv_in_channel,
                                            res_connect=res_connect, include_condition=include_condition,
                                            condition_dim=condition_dim, include_second_condition=include_second_condition,
                                            second_condition_dim=second_condition_dim, activation=activation))
            if self.use_attention_module:
                C_in1 = ori_first_conv_in_channel if first_conv else ori_mlp_spec0
                C_in2 = first_conv_in_channel if first_conv else mlp_spec[0]
                self.attention_modules.append(AttentionModule(
                    C_in1, C_in2, C_in1, C_in2, mlp_spec[-1], attention_bn=attention_setting["attention_bn"],
                    transform_grouped_feat_out=attention_setting["transform_grouped_feat_out"],
                    last_activation=attention_setting["last_activation"]))
            if self.use_global_attention_module:
                self.global_attention_modules.append(GlobalAttentionModule(
                    mlp_spec[-1], additional_dim=3, attention_bn=global_attention_setting["attention_bn"],
                    last_activation=global_attention_setting["last_activation"]))


class PointnetSAModule(PointnetSAModuleMSG):
    def __init__(self, mlp, npoint=None, radius=None, nsample=None, bn=True, use_xyz=True, t_dim=128, include_t=False,
                 include_abs_coordinate=False, include_center_coordinate=False, bn_first=False, bias=False, first_conv=False,
                 first_conv_in_channel=0, res_connect=False, include_condition=False, condition_dim=128,
                 include_second_condition=False, second_condition_dim=128, neighbor_def="radius", activation="relu",
                 attention_setting=None, global_attention_setting=None):
        super().__init__(mlps=[mlp], npoint=npoint, radii=[radius], nsamples=[nsample], bn=bn, use_xyz=use_xyz, t_dim=t_dim,
                         include_t=include_t, include_abs_coordinate=include_abs_coordinate,
                         include_center_coordinate=include_center_coordinate, bn_first=bn_first, bias=bias,
                         first_conv=first_conv, first_conv_in_channel=first_conv_in_channel, res_connect=res_connect,
                         include_condition=include_condition, condition_dim=condition_dim,
                         include_second_condition=include_second_condition, second_condition_dim=second_condition_dim,
                         neighbor_def=neighbor_def, activation=activation, attention_setting=attention_setting,
                         global_attention_setting=global_attention_setting)


class PointnetFPModule(nn.Module):
    """three_nn / three_interpolate feature propagation"""

    def __init__(self, mlp, bn=True, t_dim=128, include_t=False, bn_first=False, bias=False, first_conv=False,
                 first_conv_in_channel=0, res_connect=False, include_condition=False, condition_dim=128,
                 include_second_condition=False, second_condition_dim=128, include_grouper=False, radius=0, nsample=32,
                 use_xyz=True, include_abs_coordinate=True, include_center_coordinate=False, neighbor_def="radius",
                 activation="relu"):
        super().__init__()
        self.include_t, self.t_dim = include_t, t_dim
        self.include_condition, self.include_second_condition = include_condition, include_second_condition
        self.include_grouper = include_grouper
        if include_grouper:
            extra = _xyz_extra(use_xyz, include_abs_coordinate, include_center_coordinate)
            if first_conv:
                first_conv_in_channel += extra
            else:
                mlp[0] += extra
            self.grouper = pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz,
                                                         include_abs_coordinate=include_abs_coordinate,
                                                         include_center_coordinate=include_center_coordinate,
                                                         neighbor_def=neighbor_def)
        self.mlp = Mlp_plus_t_emb(mlp, bn, t_dim=t_dim, include_t=include_t, bn_first=bn_first, bias=bias,
                                  first_conv=first_conv, first_conv_in_channel=first_conv_in_channel, res_connect=res_connect,
                                  include_condition=include_condition, condition_dim=condition_dim,
                                  include_second_condition=include_second_condition,
                                  second_condition_dim=second_condition_dim, activation=activation)

    def forward(self, unknown, known, unknow_feats, known_feats, t_emb=None, condition_emb=None, second_condition_emb=None,
                record_neighbor_stats=False, pooling="max"):
        if known is not None:
            dist, idx = pointnet2_utils.three_nn(unknown, known)
            dist_recip = 1.0 / (dist + 1e-8)
            weight = dist_recip / torch.sum(dist_recip, dim=2, keepdim=True)
            interpolated = pointnet2_utils.three_interpolate(known_feats, idx, weight)
        else:
            interpolated = known_feats.expand(*(list(known_feats.size()[0:2]) + [unknown.size(1)]))
        new_features = torch.cat([interpolated, unknow_feats], dim=1) if unknow_feats is not None else interpolated
        if self.include_grouper:
            new_features, count = self.grouper(unknown, unknown, new_features, subset=True,
                                               record_neighbor_stats=record_neighbor_stats, return_counts=True)
        else:
            new_features = new_features.unsqueeze(-1)
        new_features = self.mlp(new_features, t_emb=t_emb if self.include_t else None,
                                condition_emb=condition_emb if self.include_condition else None,
                                second_condition_emb=second_condition_emb if self.include_second_condition else None)
        if self.include_grouper:
            return pooling_features(new_features, count=count, pooling=pooling)
        return new_features.squeeze(-1)


class FeatureMapModule(nn.Module):
    def __init__(self, mlp, radius, K, use_xyz=True, include_abs_coordinate=True, include_center_coordinate=False, bn=True,
                 bn_first=True, bias=True, res_connect=True, first_conv=False, first_conv_in_channel=0, neighbor_def="radius",
                 activation="relu", attention_setting=None, query_feature_dim=None):
        super().__init__()
        self.use_attention_module = bool(attention_setting and attention_setting["use_attention_module"])
        extra = _xyz_extra(use_xyz, include_abs_coordinate, include_center_coordinate)
        if first_conv:
            first_conv_in_channel += extra
        else:
            mlp[0] += extra
        self.mlp = Mlp_plus_t_emb(mlp, bn, include_t=False, bn_first=bn_first, bias=bias, first_conv=first_conv,
                                  first_conv_in_channel=first_conv_in_channel, res_connect=res_connect,
                                  include_condition=False, activation=activation)
        self.mapper = pointnet2_utils.QueryAndGroup(radius, K, use_xyz=use_xyz, include_abs_coordinate=include_abs_coordinate,
                                                    include_center_coordinate=include_center_coordinate,
                                                    neighbor_def=neighbor_def)
        if self.use_attention_module:
            C_in2 = first_conv_in_channel if first_conv else mlp[0]
            self.attention_module = AttentionModule(query_feature_dim, C_in2, query_feature_dim, C_in2, mlp[-1],
                                                    attention_bn=attention_setting["attention_bn"],
                                                    transform_grouped_feat_out=attention_setting["transform_grouped_feat_out"],
                                                    last_activation=attention_setting["last_activation"])

    def forward(self, xyz, features, new_xyz, subset=False, record_neighbor_stats=True, pooling="max",
                features_at_new_xyz=None):
        new_features, count = self.mapper(xyz, new_xyz, features, subset=subset, record_neighbor_stats=record_neighbor_stats,
                                          return_counts=True)
        out_features = self.mlp(new_features)
        if self.use_attention_module:
            return self.attention_module(features_at_new_xyz, new_features, out_features, count)
        return pooling_features(out_features, count=count, pooling=pooling)


class PointnetKnnFPModule(nn.Module):
    """kNN-attention feature propagation + skip concat + MLP"""

    def __init__(self, mlp1, mlp2, K, bn=True, t_dim=128, include_t=False, bn_first=False, bias=False, first_conv=False,
                 first_conv_in_channel1=0, first_conv_in_channel2=0, res_connect=False, include_condition=False,
                 condition_dim=128, include_second_condition=False, second_condition_dim=128, include_grouper=False, radius=0,
                 nsample=32, use_xyz=True, include_abs_coordinate=True, include_center_coordinate=False,
                 neighbor_def="radius", activation="relu", attention_setting=None, global_attention_setting=None):
        super().__init__()
        self.include_t, self.t_dim = include_t, t_dim
        self.include_condition, self.include_second_condition = include_condition, include_second_condition
        self.K = K
        if first_conv:
            first_conv_in_channel1 += 11
        else:
            mlp1[0] = mlp1[0] + 11
        self.mlp1 = Mlp_plus_t_emb(mlp1, bn, t_dim=t_dim, include_t=False, bn_first=bn_first, bias=bias, first_conv=first_conv,
                                   first_conv_in_channel=first_conv_in_channel1, res_connect=res_connect,
                                   include_condition=include_second_condition, condition_dim=second_condition_dim,
                                   activation=activation)
        self.use_attention_module = bool(attention_setting and attention_setting["use_attention_module"])
        if self.use_attention_module:
            C_in1 = first_conv_in_channel2 - mlp1[-1] if first_conv else mlp2[0] - mlp1[-1]
            C_in2 = first_conv_in_channel1 if first_conv else mlp1[0]
            self.attention_module = AttentionModule(C_in1, C_in2, C_in1, C_in2, mlp1[-1],
                                                    attention_bn=attention_setting["attention_bn"],
                                                    transform_grouped_feat_out=attention_setting["transform_grouped_feat_out"],
                                                    last_activation=attention_setting["last_activation"])
        self.include_grouper = include_grouper
        if include_grouper:
            extra = _xyz_extra(use_xyz, include_abs_coordinate, include_center_coordinate)
            if first_conv:
                first_conv_in_channel2 += extra
            else:
                mlp2[0] += extra
            self.grouper = pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz,
                                                         include_abs_coordinate=include_abs_coordinate,
                                                         include_center_coordinate=include_center_coordinate,
                                                         neighbor_def=neighbor_def)
        elif first_conv:
            first_conv_in_channel2 += 3
        else:
            mlp2[0] = mlp2[0] + 3
        self.mlp2 = Mlp_plus_t_emb(mlp2, bn, t_dim=t_dim, include_t=include_t, bn_first=bn_first, bias=bias,
                                   first_conv=first_conv, first_conv_in_channel=first_conv_in_channel2, res_connect=res_connect,
                                   include_condition=include_condition, condition_dim=condition_dim, activation=activation)
        self.use_global_attention_module = bool(global_attention_setting and
                                                global_attention_setting["use_global_attention_module"])
        if self.use_global_attention_module:
            self.global_attention_module = GlobalAttentionModule(mlp2[-1], additional_dim=3,
                                                                 attention_bn=global_attention_setting["attention_bn"],
                                                                 last_activation=global_attention_setting["last_activation"])

    def forward(self, unknown, known, unknow_feats, known_feats, t_emb=None, condition_emb=None, second_condition_emb=None,
                record_neighbor_stats=False, pooling="max"):
        if self.use_attention_module or self.use_global_attention_module:
            assert known is not None and unknown is not None
        if known is not None:
            grouped = pointnet2_utils.group_knn(unknown, known, known_feats, self.K, transpose=True).contiguous()
            out = self.mlp1(grouped, t_emb=None,
                            condition_emb=second_condition_emb if self.include_second_condition else None)
            if self.use_attention_module:
                interpolated = self.attention_module(unknow_feats, grouped, out, count="all")
            else:
                interpolated = pooling_features(out, count="all", pooling=pooling)
        else:
            interpolated = known_feats.expand(*(list(known_feats.size()[0:2]) + [unknown.size(1)]))
        new_features = torch.cat([interpolated, unknow_feats], dim=1) if unknow_feats is not None else interpolated
        if self.include_grouper:
            new_features, count = self.grouper(unknown, unknown, new_features, subset=True,
                                               record_neighbor_stats=record_neighbor_stats, return_counts=True)
        else:
            new_features = torch.cat([new_features, unknown.transpose(1, 2)], dim=1).unsqueeze(-1)
        new_features = self.mlp2(new_features, t_emb=t_emb if self.include_t else None,
                                 condition_emb=condition_emb if self.include_condition else None)
        if self.include_grouper:
            return pooling_features(new_features, count=count, pooling=pooling)
        new_features = new_features.squeeze(-1)
        if self.use_global_attention_module:
            new_features = self.global_attention_module(torch.cat([new_features, unknown.transpose(1, 2)], dim=1))
        return new_features
