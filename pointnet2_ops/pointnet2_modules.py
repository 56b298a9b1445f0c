"""`pointnet2_ops.pointnet2_modules` on HIP kernels: same classes, constructor signatures and state-dict names as the
reference (pointnet2_ops_lib/pointnet2_ops/pointnet2_modules.py): Swish (:17-22), MyGroupNorm (:24-42),
build_shared_mlp (:44-69), Mlp_plus_t_emb (:71-176), pooling_features (:179-211), PointnetSAModule[MSG] (:213-462),
PointnetFPModule (:465-588), FeatureMapModule (:591-663), PointnetKnnFPModule (:666-873).

Execution model (not the reference's): inside a module every K-expanded activation is a ROW-MAJOR matrix
`slide_amd.rows.Rows` [B * npoint * K][channels] -- written by the grouping kernel, read and written by the MFMA GEMMs,
normalised in place, reduced over K by the attention / pooling kernel.  A module's `forward` takes and returns the
reference's (B, C, N) tensors and converts once at entry and exit; `forward_rows` / `rows_or_ncx` are the row-major entry
points modules use among themselves.  Only Sequentials the row-major stage interpreter does not cover (GroupNorm BEFORE the
convolution, swish) and GlobalAttentionModule run as tensor programs on NCHW tensors (HipConv1x1 / HipGroupNorm kernels,
torch glue).  The latent-DDPM configurations run on the fused engine instead (slide_amd.engine)."""
from typing import List

import torch
import torch.nn as nn

from pointnet2_ops import pointnet2_utils
from pointnet2_ops.attention import AttentionModule, GlobalAttentionModule
from slide_amd import _ext
from slide_amd import rows as R
from slide_amd.nn_ops import HipConv1x1, HipGroupNorm, HipLinear

_POOL_MODES = {"max": R.POOL_MAX, "avg": R.POOL_AVG, "avg_max": R.POOL_MAX_AVG, "max_avg": R.POOL_MAX_AVG}


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


def _stages(seq):
    """a shared-MLP Sequential as [(conv, group_norm or None, relu)] stages, or None when it is not of that form
    (GroupNorm before the convolution, swish)"""
    layers = [l for l in seq if not isinstance(l, nn.Identity)]
    out, i = [], 0
    while i < len(layers):
        if not isinstance(layers[i], HipConv1x1):
            return None
        conv, gn, relu = layers[i], None, False
        i += 1
        if i < len(layers) and isinstance(layers[i], MyGroupNorm):
            gn, relu = layers[i].group_norm, layers[i].fused_relu
            i += 1
        if i < len(layers) and isinstance(layers[i], nn.ReLU):
            relu = True
            i += 1
        out.append((conv, gn, relu))
    return out or None


def _run_stages(stages, x, addvec=None, residual=None, defer_last=False):
    """every stage = one GEMM + one normalise / ReLU step; the per-sample embedding vector and the residual that the
    reference adds AFTER the Sequential ride on the last stage's step.  A step whose only consumer is the next GEMM is
    DEFERRED (slide_amd.rows.norm_act(defer=True)): statistics and per-sample scale / shift only, the consumer normalises
    the raw tensor while loading it -- no pass over the K-expanded tensor at all.  defer_last: the caller feeds the result
    straight into another convolution."""
    for k, (conv, gn, relu) in enumerate(stages):
        last = k == len(stages) - 1
        x = R.norm_act(R.conv(x, conv, stats="raw" if gn is not None else None), gn, relu=relu,
                       addvec=addvec if last else None, residual=residual if last else None,
                       defer=(not last) or (defer_last and residual is None))
    return x


class Mlp_plus_t_emb(nn.Module):
    def __init__(self, mlp_spec, bn, t_dim=128, include_t=True, bn_first=False, bias=False, first_conv=False,
                 first_conv_in_channel=0, res_connect=False, include_condition=False, condition_dim=128,
                 include_second_condition=False, second_condition_dim=128, activation="relu"):
        super().__init__()
        assert len(mlp_spec) >= 3 and (len(mlp_spec) >= 4 or not include_second_condition)
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
        if res_connect:  # identity when the widths agree
            self.res_connect = None if mlp_spec[0] == mlp_spec[-1] else HipConv1x1(mlp_spec[0], mlp_spec[-1], bias=bias)
        kw = dict(bn_first=bn_first, bias=bias, activation=activation)
        self.first_mlp = build_shared_mlp(mlp_spec[0:2], bn, **kw)
        self.second_mlp = build_shared_mlp(mlp_spec[1:3], bn, **kw)
        self.rest_mlp = build_shared_mlp(mlp_spec[2:], bn, **kw) if len(mlp_spec) > 3 else None

    def rows_ok(self):
        return all(_stages(m) is not None for m in (self.first_mlp, self.second_mlp, self.rest_mlp) if m is not None)

    def _embedding_vectors(self, t_emb, condition_emb, second_condition_emb):
        """the three per-sample vectors added after first_mlp / second_mlp / rest_mlp (None where the module has none);
        a missing or superfluous embedding is an error, as in the reference (:142-170)"""
        vecs = []
        for have, given, name, fc in ((self.include_t, t_emb, "t_emb", "fc"),
                                      (self.include_condition, condition_emb, "condition_emb", "fc_condition"),
                                      (self.include_second_condition, second_condition_emb, "second_condition_emb",
                                       "fc_second_condition")):
            if have and given is None:
                raise Exception("Should pass %s to the forward function" % name)
            if not have and given is not None:
                raise Exception("This module does not include %s but it is given" % name)
            vecs.append(getattr(self, fc)(given) if have else None)
        return vecs

    def forward_rows(self, x, t_emb=None, condition_emb=None, second_condition_emb=None):
        """x: Rows [B * S][C_in] -> Rows [B * S][mlp_spec[-1]]: 3-5 GEMMs, each followed by ONE in-place pass that
        normalises, applies the ReLU and adds the embedding vector / the residual (reference :119-176)"""
        v_t, v_c, v_c2 = self._embedding_vectors(t_emb, condition_emb, second_condition_emb)
        feat = R.conv(x, self.first_conv) if self.first_conv_bool else x
        h = _run_stages(_stages(self.first_mlp), feat, addvec=v_t, defer_last=True)
        h = _run_stages(_stages(self.second_mlp), h, addvec=v_c, defer_last=self.rest_mlp is not None)
        skip = None
        if self.res_connect_bool:
            skip = R.conv(feat, self.res_connect) if self.res_connect is not None else feat
        if self.rest_mlp is not None:
            return _run_stages(_stages(self.rest_mlp), h, addvec=v_c2, residual=skip)
        return R.norm_act(h, addvec=v_c2, residual=skip)

    def _forward_ncx(self, feature, t_emb, condition_emb, second_condition_emb):
        """tensor program for Sequentials outside the stage form (bn_first, swish)"""
        v_t, v_c, v_c2 = self._embedding_vectors(t_emb, condition_emb, second_condition_emb)
        if self.first_conv_bool:
            feature = self.first_conv(feature)
        h = self.first_mlp(feature)
        for seq, vec in ((None, v_t), (self.second_mlp, v_c), (self.rest_mlp, v_c2)):
            if seq is not None:
                h = seq(h)
            if vec is not None:
                h = h + vec[:, :, None, None]
        if self.res_connect_bool:
            h = h + (self.res_connect(feature) if self.res_connect is not None else feature)
        return h

    def rows_or_ncx(self, x, spatial, **emb):
        """Rows in, Rows out, whichever program the Sequentials allow; spatial = (npoint, K) of the rows"""
        if self.rows_ok():
            return self.forward_rows(x, **emb)
        y = self._forward_ncx(R.to_ncx(x, spatial), emb.get("t_emb"), emb.get("condition_emb"), emb.get("second_condition_emb"))
        return R.from_ncx(y)

    def forward(self, feature, t_emb=None, condition_emb=None, second_condition_emb=None):
        """feature (B, C, npoint, K) -> (B, mlp_spec[-1], npoint, K)"""
        if not feature.is_cuda:
            raise RuntimeError("CPU not supported")
        if feature.dim() == 4 and self.rows_ok():
            out = self.forward_rows(R.from_ncx(feature), t_emb, condition_emb, second_condition_emb)
            return R.to_ncx(out, feature.shape[2:])
        return self._forward_ncx(feature, t_emb, condition_emb, second_condition_emb)


def pool_rows(x, K, pooling, counts=None):
    return R.pool(x, K, _POOL_MODES[pooling], counts)


def pooling_features(feature, count=None, pooling="max"):
    """(B, C, npoint, K) -> (B, C, npoint): max over all K slots, mean over the first `count` slots, or max for the first
    half of the channels and mean for the second ('avg_max')"""
    assert pooling in _POOL_MODES
    K = feature.shape[3]
    return R.to_ncx(pool_rows(R.from_ncx(feature, half=False), K, pooling, None if isinstance(count, str) else count))


def _xyz_extra(use_xyz, include_abs_coordinate, include_center_coordinate):
    return (3 + (3 if include_abs_coordinate else 0) + (3 if include_center_coordinate else 0)) if use_xyz else 0


def _effective_counts(grouper, counts, length):
    """neighbour counts a reduction over K has to honour; None when every slot is a real neighbour (kNN grouping of
    full-length clouds), which lets the kernels skip the count reads"""
    return None if (grouper.neighbor_def == "nn" and length is None) else counts


def _emb_kwargs(module, t_emb, condition_emb, second_condition_emb=None):
    return dict(t_emb=t_emb if module.include_t else None,
                condition_emb=condition_emb if module.include_condition else None,
                second_condition_emb=second_condition_emb if getattr(module, "include_second_condition", False) else None)


class PointnetSAModuleMSG(nn.Module):
    """set abstraction: FPS centres, one (grouper, Mlp, [attention | pooling]) branch per scale, outputs concatenated"""

    def __init__(self, npoint, radii, nsamples, mlps, bn=True, use_xyz=True, t_dim=128, include_t=False,
                 include_abs_coordinate=False, include_center_coordinate=False, bn_first=False, bias=False, first_conv=False,
                 first_conv_in_channel=0, res_connect=False, include_condition=False, condition_dim=128,
                 include_second_condition=False, second_condition_dim=128, neighbor_def="radius", activation="relu",
                 attention_setting=None, global_attention_setting=None):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.npoint = npoint
        self.include_t, self.t_dim = include_t, t_dim
        self.include_condition, self.condition_dim = include_condition, condition_dim
        self.include_second_condition, self.second_condition_dim = include_second_condition, second_condition_dim
        self.use_attention_module = bool(attention_setting and attention_setting["use_attention_module"])
        self.use_global_attention_module = bool(global_attention_setting and
                                                global_attention_setting["use_global_attention_module"])
        self.groupers, self.mlps = nn.ModuleList(), nn.ModuleList()
        self.attention_modules = nn.ModuleList() if self.use_attention_module else None
        self.global_attention_modules = nn.ModuleList() if self.use_global_attention_module else None
        coord_ch = _xyz_extra(use_xyz, include_abs_coordinate, include_center_coordinate)
        for radius, nsample, spec in zip(radii, nsamples, mlps):
            self.groupers.append(
                pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz, include_abs_coordinate=include_abs_coordinate,
                                              include_center_coordinate=include_center_coordinate, neighbor_def=neighbor_def)
                if npoint is not None else pointnet2_utils.GroupAll(use_xyz))
            # the coordinate channels enter through first_conv when there is one, else through the first MLP layer
            # (the caller's spec list is widened in place, like the reference's)
            point_ch = first_conv_in_channel if first_conv else spec[0]
            grouped_ch = point_ch + coord_ch
            if not first_conv:
                spec[0] = grouped_ch
            self.mlps.append(Mlp_plus_t_emb(spec, bn, t_dim=t_dim, include_t=include_t, bn_first=bn_first, bias=bias,
                                            first_conv=first_conv, first_conv_in_channel=grouped_ch if first_conv else 0,
                                            res_connect=res_connect, include_condition=include_condition,
                                            condition_dim=condition_dim, include_second_condition=include_second_condition,
                                            second_condition_dim=second_condition_dim, activation=activation))
            if self.use_attention_module:
                self.attention_modules.append(AttentionModule(
                    point_ch, grouped_ch, point_ch, grouped_ch, spec[-1], attention_bn=attention_setting["attention_bn"],
                    transform_grouped_feat_out=attention_setting["transform_grouped_feat_out"],
                    last_activation=attention_setting["last_activation"]))
            if self.use_global_attention_module:
                self.global_attention_modules.append(GlobalAttentionModule(
                    spec[-1], additional_dim=3, attention_bn=global_attention_setting["attention_bn"],
                    last_activation=global_attention_setting["last_activation"]))
            if first_conv:
                first_conv_in_channel = grouped_ch  # (the reference accumulates the widening across scales)

    def forward(self, xyz, features, t_emb=None, condition_emb=None, second_condition_emb=None, subset=True,
                record_neighbor_stats=False, pooling="max", length=None):
        """xyz (B, N, 3), features (B, C, N) -> (centres (B, npoint, 3), (B, sum C_out, npoint))"""
        assert self.npoint is not None
        if not xyz.is_cuda:
            raise RuntimeError("CPU not supported")
        feat = R.from_ncx(features) if features is not None else None
        if xyz.shape[1] <= self.npoint:  # nothing to sub-sample: every point is a centre
            centres, query = xyz, feat
        else:
            picked = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
            # (row-layout gather of the (B, N, 3) coordinates: one launch that moves exactly the picked rows, instead of transpose ->
            #  gather_points on (B, 3, N) -> transpose; the reference's gather_operation call site: pointnet2_modules.py:368-378)
            if xyz.requires_grad:  # gather_rows has no autograd Function: keep the coordinates' gradient path
                centres = pointnet2_utils.gather_operation(xyz.transpose(1, 2).contiguous(), picked).transpose(1, 2).contiguous()
            else:
                centres = _ext.gather_rows(xyz.contiguous().float(), picked)
            query = R.gather_rows(feat, picked) if (self.use_attention_module and feat is not None) else None
        emb = _emb_kwargs(self, t_emb, condition_emb, second_condition_emb)
        outs = []
        for i, (grouper, mlp) in enumerate(zip(self.groupers, self.mlps)):
            grouped, counts, K = grouper.rows(xyz, centres, feat, subset, record_neighbor_stats, length)
            h = mlp.rows_or_ncx(grouped, (centres.shape[1], K), **emb)
            counts = _effective_counts(grouper, counts, length)
            if self.use_attention_module:
                o = self.attention_modules[i].forward_rows(query, grouped, h, K, counts)
            else:
                o = pool_rows(h, K, pooling, counts)
            if self.use_global_attention_module:
                o = R.from_ncx(self.global_attention_modules[i](torch.cat([R.to_ncx(o), centres.transpose(1, 2)], dim=1)))
            outs.append(o)
        return centres, R.to_ncx(outs[0] if len(outs) == 1 else R.concat_cols(outs))


class PointnetSAModule(PointnetSAModuleMSG):
    """single-scale set abstraction"""

    def __init__(self, mlp, npoint=None, radius=None, nsample=None, bn=True, use_xyz=True, t_dim=128, include_t=False,
                 include_abs_coordinate=False, include_center_coordinate=False, bn_first=False, bias=False, first_conv=False,
                 first_conv_in_channel=0, res_connect=False, include_condition=False, condition_dim=128,
                 include_second_condition=False, second_condition_dim=128, neighbor_def="radius", activation="relu",
                 attention_setting=None, global_attention_setting=None):
        super().__init__(npoint, [radius], [nsample], [mlp], bn=bn, use_xyz=use_xyz, t_dim=t_dim, include_t=include_t,
                         include_abs_coordinate=include_abs_coordinate, include_center_coordinate=include_center_coordinate,
                         bn_first=bn_first, bias=bias, first_conv=first_conv, first_conv_in_channel=first_conv_in_channel,
                         res_connect=res_connect, include_condition=include_condition, condition_dim=condition_dim,
                         include_second_condition=include_second_condition, second_condition_dim=second_condition_dim,
                         neighbor_def=neighbor_def, activation=activation, attention_setting=attention_setting,
                         global_attention_setting=global_attention_setting)


def _make_local_grouper(owner, enabled, first_conv, first_conv_in_channel, spec, radius, nsample, use_xyz,
                        include_abs_coordinate, include_center_coordinate, neighbor_def):
    """optional QueryAndGroup of a feature-propagation module over its own output points; returns the widened
    first_conv_in_channel (spec[0] is widened in place when there is no first_conv)"""
    owner.include_grouper = enabled
    if not enabled:
        return first_conv_in_channel
    coord_ch = _xyz_extra(use_xyz, include_abs_coordinate, include_center_coordinate)
    owner.grouper = pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz, include_abs_coordinate=include_abs_coordinate,
                                                  include_center_coordinate=include_center_coordinate, neighbor_def=neighbor_def)
    if first_conv:
        return first_conv_in_channel + coord_ch
    spec[0] += coord_ch
    return first_conv_in_channel


class PointnetFPModule(nn.Module):
    """feature propagation by inverse-distance interpolation over the three nearest known points, then an Mlp"""

    def __init__(self, mlp, bn=True, t_dim=128, include_t=False, bn_first=False, bias=False, first_conv=False,
                 first_conv_in_channel=0, res_connect=False, include_condition=False, condition_dim=128,
                 include_second_condition=False, second_condition_dim=128, include_grouper=False, radius=0, nsample=32,
                 use_xyz=True, include_abs_coordinate=True, include_center_coordinate=False, neighbor_def="radius",
                 activation="relu"):
        super().__init__()
        self.include_t, self.t_dim = include_t, t_dim
        self.include_condition, self.include_second_condition = include_condition, include_second_condition
        first_conv_in_channel = _make_local_grouper(self, include_grouper, first_conv, first_conv_in_channel, mlp, radius, nsample,
                                                    use_xyz, include_abs_coordinate, include_center_coordinate, neighbor_def)
        self.mlp = Mlp_plus_t_emb(mlp, bn, t_dim=t_dim, include_t=include_t, bn_first=bn_first, bias=bias,
                                  first_conv=first_conv, first_conv_in_channel=first_conv_in_channel, res_connect=res_connect,
                                  include_condition=include_condition, condition_dim=condition_dim,
                                  include_second_condition=include_second_condition,
                                  second_condition_dim=second_condition_dim, activation=activation)

    def forward(self, unknown, known, unknow_feats, known_feats, t_emb=None, condition_emb=None, second_condition_emb=None,
                record_neighbor_stats=False, pooling="max"):
        """unknown (B, n, 3), known (B, m, 3) or None, unknow_feats (B, C1, n) or None, known_feats (B, C2, m) -> (B, C, n)"""
        n = unknown.shape[1]
        if known is None:  # one global feature vector, broadcast to every point
            spread = known_feats.expand(-1, -1, n)
        else:
            dist, nearest = pointnet2_utils.three_nn(unknown, known)
            w = 1.0 / (dist + 1e-8)
            spread = pointnet2_utils.three_interpolate(known_feats, nearest, w / w.sum(dim=2, keepdim=True))
        parts = [R.from_ncx(spread)] + ([R.from_ncx(unknow_feats)] if unknow_feats is not None else [])
        x = parts[0] if len(parts) == 1 else R.concat_cols(parts)
        emb = _emb_kwargs(self, t_emb, condition_emb, second_condition_emb)
        if not self.include_grouper:
            return R.to_ncx(self.mlp.rows_or_ncx(x, (n, 1), **emb))
        grouped, counts, K = self.grouper.rows(unknown, unknown, x, True, record_neighbor_stats)
        h = self.mlp.rows_or_ncx(grouped, (n, K), **emb)
        return R.to_ncx(pool_rows(h, K, pooling, _effective_counts(self.grouper, counts, None)))


class FeatureMapModule(nn.Module):
    """features of a cloud mapped onto another set of points: grouping around the new points, Mlp, attention / pooling"""

    def __init__(self, mlp, radius, K, use_xyz=True, include_abs_coordinate=True, include_center_coordinate=False, bn=True,
                 bn_first=True, bias=True, res_connect=True, first_conv=False, first_conv_in_channel=0, neighbor_def="radius",
                 activation="relu", attention_setting=None, query_feature_dim=None):
        super().__init__()
        self.use_attention_module = bool(attention_setting and attention_setting["use_attention_module"])
        coord_ch = _xyz_extra(use_xyz, include_abs_coordinate, include_center_coordinate)
        if first_conv:
            first_conv_in_channel += coord_ch
        else:
            mlp[0] += coord_ch
        grouped_ch = first_conv_in_channel if first_conv else mlp[0]
        self.mlp = Mlp_plus_t_emb(mlp, bn, include_t=False, bn_first=bn_first, bias=bias, first_conv=first_conv,
                                  first_conv_in_channel=first_conv_in_channel, res_connect=res_connect,
                                  include_condition=False, activation=activation)
        self.mapper = pointnet2_utils.QueryAndGroup(radius, K, use_xyz=use_xyz, include_abs_coordinate=include_abs_coordinate,
                                                    include_center_coordinate=include_center_coordinate,
                                                    neighbor_def=neighbor_def)
        if self.use_attention_module:
            self.attention_module = AttentionModule(query_feature_dim, grouped_ch, query_feature_dim, grouped_ch, mlp[-1],
                                                    attention_bn=attention_setting["attention_bn"],
                                                    transform_grouped_feat_out=attention_setting["transform_grouped_feat_out"],
                                                    last_activation=attention_setting["last_activation"])

    def forward(self, xyz, features, new_xyz, subset=False, record_neighbor_stats=True, pooling="max",
                features_at_new_xyz=None):
        """xyz (B, N, 3) with features (B, C, N) -> features at new_xyz (B, m, 3): (B, mlp[-1], m)"""
        grouped, counts, K = self.mapper.rows(xyz, new_xyz, R.from_ncx(features), subset, record_neighbor_stats)
        h = self.mlp.rows_or_ncx(grouped, (new_xyz.shape[1], K))
        counts = _effective_counts(self.mapper, counts, None)
        if self.use_attention_module:
            return R.to_ncx(self.attention_module.forward_rows(R.from_ncx(features_at_new_xyz), grouped, h, K, counts))
        return R.to_ncx(pool_rows(h, K, pooling, counts))


class PointnetKnnFPModule(nn.Module):
    """feature propagation by attention (or pooling) over the K nearest known points, skip concatenation, second Mlp"""

    def __init__(self, mlp1, mlp2, K, bn=True, t_dim=128, include_t=False, bn_first=False, bias=False, first_conv=False,
                 first_conv_in_channel1=0, first_conv_in_channel2=0, res_connect=False, include_condition=False,
                 condition_dim=128, include_second_condition=False, second_condition_dim=128, include_grouper=False, radius=0,
                 nsample=32, use_xyz=True, include_abs_coordinate=True, include_center_coordinate=False,
                 neighbor_def="radius", activation="relu", attention_setting=None, global_attention_setting=None):
        super().__init__()
        self.K = K
        self.include_t, self.t_dim = include_t, t_dim
        self.include_condition, self.include_second_condition = include_condition, include_second_condition
        self.use_attention_module = bool(attention_setting and attention_setting["use_attention_module"])
        self.use_global_attention_module = bool(global_attention_setting and
                                                global_attention_setting["use_global_attention_module"])
        # stage 1 sees a known point's features + the 11 geometry channels of group_knn (d2, weight, abs, rel, centre)
        if first_conv:
            first_conv_in_channel1 += 11
        else:
            mlp1[0] += 11
        grouped_ch = first_conv_in_channel1 if first_conv else mlp1[0]
        self.mlp1 = Mlp_plus_t_emb(mlp1, bn, t_dim=t_dim, include_t=False, bn_first=bn_first, bias=bias, first_conv=first_conv,
                                   first_conv_in_channel=first_conv_in_channel1, res_connect=res_connect,
                                   include_condition=include_second_condition, condition_dim=second_condition_dim,
                                   activation=activation)
        if self.use_attention_module:  # the query is the skip feature: what stage 2 receives minus stage 1's output
            skip_ch = (first_conv_in_channel2 if first_conv else mlp2[0]) - mlp1[-1]
            self.attention_module = AttentionModule(skip_ch, grouped_ch, skip_ch, grouped_ch, mlp1[-1],
                                                    attention_bn=attention_setting["attention_bn"],
                                                    transform_grouped_feat_out=attention_setting["transform_grouped_feat_out"],
                                                    last_activation=attention_setting["last_activation"])
        first_conv_in_channel2 = _make_local_grouper(self, include_grouper, first_conv, first_conv_in_channel2, mlp2, radius,
                                                     nsample, use_xyz, include_abs_coordinate, include_center_coordinate,
                                                     neighbor_def)
        if not include_grouper:  # stage 2 sees the point's own coordinates instead
            if first_conv:
                first_conv_in_channel2 += 3
            else:
                mlp2[0] += 3
        self.mlp2 = Mlp_plus_t_emb(mlp2, bn, t_dim=t_dim, include_t=include_t, bn_first=bn_first, bias=bias,
                                   first_conv=first_conv, first_conv_in_channel=first_conv_in_channel2, res_connect=res_connect,
                                   include_condition=include_condition, condition_dim=condition_dim, activation=activation)
        if self.use_global_attention_module:
            self.global_attention_module = GlobalAttentionModule(mlp2[-1], additional_dim=3,
                                                                 attention_bn=global_attention_setting["attention_bn"],
                                                                 last_activation=global_attention_setting["last_activation"])

    def forward(self, unknown, known, unknow_feats, known_feats, t_emb=None, condition_emb=None, second_condition_emb=None,
                record_neighbor_stats=False, pooling="max"):
        """unknown (B, n, 3), known (B, m, 3) or None, unknow_feats (B, C1, n) or None, known_feats (B, C2, m) -> (B, C, n)"""
        if not unknown.is_cuda:
            raise RuntimeError("CPU not supported")
        if self.use_attention_module or self.use_global_attention_module:
            assert known is not None and unknown is not None
        n = unknown.shape[1]
        skip = R.from_ncx(unknow_feats) if unknow_feats is not None else None
        if known is None:
            carried = R.from_ncx(known_feats.expand(-1, -1, n))
        else:
            grouped = pointnet2_utils.group_knn_rows(unknown, known, R.from_ncx(known_feats), self.K)
            h = self.mlp1.rows_or_ncx(grouped, (n, self.K),
                                      condition_emb=second_condition_emb if self.include_second_condition else None)
            if self.use_attention_module:
                carried = self.attention_module.forward_rows(skip, grouped, h, self.K)
            else:
                carried = pool_rows(h, self.K, pooling)
        parts = [carried] + ([skip] if skip is not None else [])
        emb = dict(t_emb=t_emb if self.include_t else None, condition_emb=condition_emb if self.include_condition else None)
        if self.include_grouper:
            x = parts[0] if len(parts) == 1 else R.concat_cols(parts)
            regrouped, counts, K2 = self.grouper.rows(unknown, unknown, x, True, record_neighbor_stats)
            out = pool_rows(self.mlp2.rows_or_ncx(regrouped, (n, K2), **emb), K2, pooling,
                            _effective_counts(self.grouper, counts, None))
        else:
            out = self.mlp2.rows_or_ncx(R.concat_cols(parts + [unknown]), (n, 1), **emb)
        out = R.to_ncx(out)
        if self.use_global_attention_module:
            out = self.global_attention_module(torch.cat([out, unknown.transpose(1, 2)], dim=1))
        return out
