"""`pointnet2_ops.pointnet2_modules` on HIP kernels: same classes, constructor signatures and state-dict names as the
reference (pointnet2_ops_lib/pointnet2_ops/pointnet2_modules.py): Swish (:17-22), MyGroupNorm (:24-42),
build_shared_mlp (:44-69), Mlp_plus_t_emb (:71-176), pooling_features (:179-211), PointnetSAModule[MSG] (:213-462),
PointnetFPModule (:465-588), FeatureMapModule (:591-663), PointnetKnnFPModule (:666-873).

Two execution paths per module:
  * ROW-MAJOR (slide_amd.rows; taken for the configurations every shipped model uses -- kNN grouping, GroupNorm after the
    convolution, ReLU, vector attention): a grouped activation is one [B * npoint * K][channels] matrix from the grouping
    kernel to the attention reduction; reference-layout (B, C, N) tensors exist only at the module boundary.  `forward_rows`
    methods take / return `slide_amd.rows.Rows`.
  * GENERAL (any radius-or-nn grouping / bn_first / swish / pooling): the reference's tensor program on NCHW tensors with HIP
    kernels for search, FPS, gathers, 1x1 convolutions and GroupNorm; concatenation, ReLU and softmax glue are torch ops.
The latent-DDPM configurations run on the fused engine instead (slide_amd.engine)."""
import copy
import os
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from pointnet2_ops import pointnet2_utils
from pointnet2_ops.attention import AttentionModule, GlobalAttentionModule
from slide_amd import rows as R
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


def _seq_rows_ok(seq):
    """conv -> [GroupNorm] -> [ReLU] stages only (bn_first / swish Sequentials run on the general path)"""
    layers = [l for l in seq if not isinstance(l, nn.Identity)]
    if not layers or not isinstance(layers[0], HipConv1x1):
        return False
    for a, b in zip(layers, layers[1:] + [None]):
        if isinstance(a, HipConv1x1):
            continue
        if isinstance(a, MyGroupNorm) and (b is None or isinstance(b, (HipConv1x1, nn.ReLU))):
            continue
        if isinstance(a, nn.ReLU) and (b is None or isinstance(b, HipConv1x1)):
            continue
        return False
    return True


def _seq_rows(seq, x, addvec=None, residual=None):
    """a shared-MLP Sequential on Rows: every stage = one GEMM + one in-place normalise / ReLU pass; the per-sample
    embedding vector and the residual the reference adds AFTER the Sequential ride on the last stage's pass"""
    layers = [l for l in seq if not isinstance(l, nn.Identity)]
    i, n = 0, len(layers)
    while i < n:
        x = R.conv(x, layers[i])
        i += 1
        gn, relu = None, False
        if i < n and isinstance(layers[i], MyGroupNorm):
            gn, relu = layers[i].group_norm, layers[i].fused_relu
            i += 1
        if i < n and isinstance(layers[i], nn.ReLU):
            relu = True
            i += 1
        last = i >= n
        R.norm_act(x, gn, relu=relu, addvec=addvec if last else None, residual=residual if last else None)
    return x


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

    def rows_ok(self):
        return all(_seq_rows_ok(m) for m in (self.first_mlp, self.second_mlp, self.rest_mlp) if m is not None)

    def _check_embeddings(self, t_emb, condition_emb, second_condition_emb):
        if self.include_t and t_emb is None:
            raise Exception("Should pass t_emb to the forward function")
        if not self.include_t and t_emb is not None:
            raise Exception("This module does not include t but t_emb is given")
        if self.include_condition and condition_emb is None:
            raise Exception("Should pass condition_emb to the forward function")
        if not self.include_condition and condition_emb is not None:
            raise Exception("This module does not include condition but condition_emb is given")
        if self.include_second_condition and second_condition_emb is None:
            raise Exception("Should pass second_condition_emb to the forward function")
        if not self.include_second_condition and second_condition_emb is not None:
            raise Exception("This module does not include condition but condition_emb is given")

    def forward_rows(self, x, t_emb=None, condition_emb=None, second_condition_emb=None):
        """x: Rows [B * S][C_in] -> Rows [B * S][mlp_spec[-1]]: 3-5 GEMMs, each followed by ONE in-place pass that
        normalises, applies the ReLU and adds the embedding vector / the residual (reference :119-176)"""
        self._check_embeddings(t_emb, condition_emb, second_condition_emb)
        feat = R.conv(x, self.first_conv) if self.first_conv_bool else x
        h = _seq_rows(self.first_mlp, feat, addvec=self.fc(t_emb) if self.include_t else None)
        h = _seq_rows(self.second_mlp, h, addvec=self.fc_condition(condition_emb) if self.include_condition else None)
        res = None
        if self.res_connect_bool:
            res = R.conv(feat, self.res_connect) if self.res_connect is not None else feat
        vec2 = self.fc_second_condition(second_condition_emb) if self.include_second_condition else None
        if self.rest_mlp is not None:
            return _seq_rows(self.rest_mlp, h, addvec=vec2, residual=res)
        return R.norm_act(h, addvec=vec2, residual=res)

    def forward(self, feature, t_emb=None, condition_emb=None, second_condition_emb=None):
        if _rows_enabled(feature) and feature.dim() == 4 and self.rows_ok():
            out = self.forward_rows(R.from_ncx(feature), t_emb, condition_emb, second_condition_emb)
            return R.to_ncx(out, feature.shape[2:])
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


def _rows_enabled(t):
    """row-major fast path switch: CUDA tensors, unless SLIDE_MODULE_ROWS=0 forces the general NCHW program (A/B tests)"""
    return t.is_cuda and os.environ.get("SLIDE_MODULE_ROWS", "1") != "0"


def _group_flags(grouper):
    if not grouper.use_xyz:
        return R.GROUP_NO_XYZ
    return (R.GROUP_ABS if grouper.include_abs_coordinate else 0) | (R.GROUP_CENTER if grouper.include_center_coordinate else 0)


def _nn_grouper(g):
    return isinstance(g, pointnet2_utils.QueryAndGroup) and g.neighbor_def == "nn"


def _group_rows(grouper, xyz, new_xyz, feat, record_neighbor_stats=False):
    """QueryAndGroup('nn') on Rows: (grouped Rows [B * npoint * K], K)"""
    K = min(grouper.nsample, xyz.shape[1])
    _, idx, _ = pointnet2_utils.knn.knn_points(new_xyz, xyz, K=K)
    if record_neighbor_stats:  # every centre has exactly K neighbours under the kNN definition
        grouper.neighbor_stats = torch.full((3,), float(K), device=xyz.device)
        grouper.neighbor_num_quantile = torch.full((grouper.quantile.numel(),), K, dtype=torch.long, device=xyz.device)
    return R.group(xyz, new_xyz, feat, idx, _group_flags(grouper)), K


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
        assert self.npoint is not None
        if (_rows_enabled(xyz) and features is not None and length is None and self.use_attention_module
                and not self.use_global_attention_module and all(_nn_grouper(g) for g in self.groupers)
                and all(m.rows_ok() for m in self.mlps) and xyz.shape[2] == 3):
            return self._forward_rows(xyz, features, t_emb, condition_emb, second_condition_emb, record_neighbor_stats)
        new_features_list = []
        xyz_flipped = xyz.transpose(1, 2).contiguous()
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


    def _forward_rows(self, xyz, features, t_emb, condition_emb, second_condition_emb, record_neighbor_stats):
        """kNN grouping + Mlp + vector attention with every K-expanded tensor row-major (see slide_amd.rows)"""
        feat = R.from_ncx(features)
        if xyz.shape[1] <= self.npoint:
            new_xyz, query = xyz, feat
        else:
            fidx = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
            new_xyz = pointnet2_utils.gather_operation(xyz.transpose(1, 2).contiguous(), fidx).transpose(1, 2).contiguous()
            query = R.gather_rows(feat, fidx)
        outs = []
        for grouper, mlp, att in zip(self.groupers, self.mlps, self.attention_modules):
            grouped, K = _group_rows(grouper, xyz, new_xyz, feat, record_neighbor_stats)
            h = mlp.forward_rows(grouped, t_emb=t_emb if self.include_t else None,
                                 condition_emb=condition_emb if self.include_condition else None,
                                 second_condition_emb=second_condition_emb if self.include_second_condition else None)
            outs.append(att.forward_rows(query, grouped, h, K))
        return new_xyz, R.to_ncx(outs[0] if len(outs) == 1 else R.concat_cols(outs))


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
        if (_rows_enabled(xyz) and self.use_attention_module and features_at_new_xyz is not None and _nn_grouper(self.mapper)
                and self.mlp.rows_ok() and xyz.shape[2] == 3):
            grouped, K = _group_rows(self.mapper, xyz, new_xyz, R.from_ncx(features), record_neighbor_stats)
            out = self.attention_module.forward_rows(R.from_ncx(features_at_new_xyz), grouped, self.mlp.forward_rows(grouped), K)
            return R.to_ncx(out)
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
        if (known is not None and _rows_enabled(unknown) and self.use_attention_module and not self.use_global_attention_module
                and not self.include_grouper and unknow_feats is not None and self.mlp1.rows_ok() and self.mlp2.rows_ok()
                and unknown.shape[2] == 3):
            # group_knn rows [feat | d2 | w | abs | rel | centre] -> mlp1 -> attention over the K known neighbours;
            # [interpolated | skip features | xyz] -> mlp2, all row-major
            d2, idx, _ = pointnet2_utils.knn.knn_points(unknown, known, K=self.K)
            skip = R.from_ncx(unknow_feats)
            grouped = R.group(known, unknown, R.from_ncx(known_feats), idx, R.GROUP_FP, d2=d2)
            h = self.mlp1.forward_rows(grouped, condition_emb=second_condition_emb if self.include_second_condition else None)
            interpolated = self.attention_module.forward_rows(skip, grouped, h, self.K)
            out = self.mlp2.forward_rows(R.concat_cols([interpolated, skip, unknown]),
                                         t_emb=t_emb if self.include_t else None,
                                         condition_emb=condition_emb if self.include_condition else None)
            return R.to_ncx(out)
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
