"""`PointNet2CloudCondition` -- the per-timestep denoiser (reference: pointnet2/models/pointnet2_with_pcld_condition.py:
27-489 on top of pointnet2/models/pointnet2_ssg_sem.py:34-177) assembled from the HIP-backed `pointnet2_ops` modules, with
the reference's attribute / state-dict names, for the configuration family every shipped
latent-DDPM config uses (no condition cloud: include_local_feature / include_global_feature False).

Two execution paths:
  forward(...)             general module path (any N, FPS when N > npoint), one HIP launch per primitive
  forward(..., fused=True) the fused plan of slide_amd.engine.DenoiserEngine (16 latent points), built lazily per batch size
Configuration branches beside the shipped family (round 6; module path only -- the fused plan raises for them): the three-nearest-
neighbour FP module (`use_knn_FP` False: reference pointnet2_ssg_sem.py:160-176), `bn_first` (GroupNorm -> activation -> conv, and the
activation + conv output head: pointnet2_with_pcld_condition.py:259-264) and `bn` False (:272-277); pinned by
tests/golden/golden_denoiser_variants.npz (generated from the imported reference, tools/gen_golden.py).
"""
import numpy as np
import torch
import torch.nn as nn

from pointnet2_ops.pointnet2_modules import PointnetFPModule, PointnetKnnFPModule, PointnetSAModule
from slide_amd.nn_ops import HipConv1x1, HipGroupNorm, HipLinear


def swish(x):
    return x * torch.sigmoid(x)


def calc_t_emb(ts, t_emb_dim):
    """sinusoidal timestep embedding (pointnet2/models/pointnet2_ssg_sem.py:14-31)"""
    assert t_emb_dim % 2 == 0
    half = t_emb_dim // 2
    c = np.log(10000) / (half - 1)
    freq = torch.exp(torch.arange(half) * -c).to(ts.device)
    arg = ts.unsqueeze(1) * freq
    return torch.cat((torch.sin(arg), torch.cos(arg)), 1)


class PointNet2CloudCondition(nn.Module):
    def __init__(self, hparams):
        super().__init__()
        self.hparams = hp = hparams
        if hp.get("include_local_feature", True) or hp.get("include_global_feature", False):
            raise NotImplementedError("condition-cloud branches are outside the latent-DDPM sampling path (SURVEY.md section 8)")
        arch = hp["architecture"]
        assert hp.get("activation", "relu") == "relu" and hp.get("point_upsample_factor", 1) == 1
        assert not hp.get("use_position_encoding", False) and not hp.get("concate_partial_with_noisy_input", False)
        self.bn, self.bn_first, self.use_knn_FP = hp.get("bn", True), hp["bn_first"], arch.get("use_knn_FP", False)
        att = hp.get("attention_setting", None)
        t_dim = hp["t_dim"]
        self.class_emb = nn.Embedding(hp["num_class"], hp["class_condition_dim"]) if hp["include_class_condition"] else None
        in_fea = hp["in_fea_dim"] + (3 if hp["attach_position_to_input_feature"] else 0)
        self.fc_t1, self.fc_t2 = HipLinear(t_dim, 4 * t_dim), HipLinear(4 * t_dim, 4 * t_dim)
        common = dict(t_dim=4 * t_dim, include_t=hp["include_t"], bn_first=self.bn_first, res_connect=hp["res_connect"], bias=hp["bias"],
                      include_condition=hp["include_class_condition"], condition_dim=hp["class_condition_dim"],
                      use_xyz=hp["model.use_xyz"], include_abs_coordinate=hp["include_abs_coordinate"],
                      include_center_coordinate=hp.get("include_center_coordinate", False),
                      neighbor_def=arch["neighbor_definition"], bn=self.bn)
        f, depth = arch["feature_dim"], arch["mlp_depth"]
        self.SA_modules = nn.ModuleList()
        for i in range(len(arch["npoint"])):
            first_conv = self.bn_first and i == 0  # (bn_first: a convolution ahead of the first GroupNorm, pointnet2_ssg_sem.py:66-73)
            spec = [in_fea if (i == 0 and not first_conv) else f[i]] + [f[i]] * (depth - 1) + [f[i + 1]]
            self.SA_modules.append(PointnetSAModule(npoint=arch["npoint"][i], radius=arch["radius"][i], nsample=arch["nsample"][i],
                                                    mlp=spec, first_conv=first_conv, first_conv_in_channel=in_fea,
                                                    attention_setting=att, **common))
        d, ddepth = arch["decoder_feature_dim"], arch["decoder_mlp_depth"]
        assert d[-1] == f[-1]
        self.FP_modules = nn.ModuleList()
        for i in range(len(d) - 1):
            skip = in_fea if i == 0 else f[i]
            if self.use_knn_FP:
                self.FP_modules.append(PointnetKnnFPModule(mlp1=[d[i + 1]] + [d[i]] * ddepth, mlp2=[d[i] + skip] + [d[i]] * ddepth,
                                                           K=arch.get("K", 3), first_conv=False,
                                                           include_grouper=arch.get("include_grouper", False),
                                                           radius=arch["radius"][i], nsample=arch["nsample"][i],
                                                           attention_setting=att, **common))
            else:  # three nearest known points, inverse-distance weights, ONE Mlp (pointnet2_ssg_sem.py:160-176)
                self.FP_modules.append(PointnetFPModule(mlp=[d[i + 1] + skip] + [d[i]] * ddepth, first_conv=False,
                                                        include_grouper=arch.get("include_grouper", False),
                                                        radius=arch["radius"][i], nsample=arch["nsample"][i], **common))
        self.transform_output = hp.get("transform_output", True)
        if self.transform_output:  # (layer positions as in the reference's Sequentials: state-dict keys fc_lyaer.<index>.*)
            if self.bn_first:
                self.fc_lyaer = nn.Sequential(nn.ReLU(True), HipConv1x1(d[0] + 3, hp["out_dim"], ndim=1))
            elif self.bn:
                self.fc_lyaer = nn.Sequential(HipConv1x1(d[0] + 3, 128, bias=hp["bias"], ndim=1), HipGroupNorm(32, 128),
                                              nn.ReLU(True), HipConv1x1(128, hp["out_dim"], ndim=1))
            else:
                self.fc_lyaer = nn.Sequential(HipConv1x1(d[0] + 3, 128, bias=hp["bias"], ndim=1), nn.ReLU(True),
                                              HipConv1x1(128, hp["out_dim"], ndim=1))
        self._engines = {}

    def _break_up_pc(self, pc):
        return pc[..., 0:3].contiguous(), (pc[..., 3:].transpose(1, 2).contiguous() if pc.size(-1) > 3 else None)

    @torch.no_grad()
    def forward(self, pointcloud, condition=None, ts=None, label=None, use_retained_condition_feature=False, fused=False):
        assert condition is None
        if fused:
            return self._fused(pointcloud, ts, label)
        hp = self.hparams
        pc = torch.cat([pointcloud, pointcloud[:, :, 0:3]], dim=2) if hp["attach_position_to_input_feature"] else pointcloud
        xyz, features = self._break_up_pc(pc)
        t_emb = None
        if ts is not None and hp["include_t"]:
            t_emb = swish(self.fc_t2(swish(self.fc_t1(calc_t_emb(ts, hp["t_dim"])))))
        cond = self.class_emb(label) if (label is not None and self.class_emb is not None) else None
        l_xyz, l_features = [xyz], [features]
        for i, m in enumerate(self.SA_modules):
            nx, nf = m(l_xyz[i], l_features[i], t_emb=t_emb, condition_emb=cond, subset=True, pooling=hp.get("pooling", "max"))
            l_xyz.append(nx); l_features.append(nf)
        for i in range(-1, -(len(self.FP_modules) + 1), -1):
            l_features[i - 1] = self.FP_modules[i](l_xyz[i - 1], l_xyz[i], l_features[i - 1], l_features[i], t_emb=t_emb,
                                                   condition_emb=cond)
        if not self.transform_output:  # feature-extractor use (autoencoder decoder levels): per-point features
            return l_features[0].transpose(1, 2).contiguous()
        out = torch.cat([l_features[0], xyz.transpose(1, 2)], dim=1)
        if self.bn_first:
            return self.fc_lyaer[1](torch.relu(out)).transpose(1, 2).contiguous()
        h = self.fc_lyaer[0](out)
        if not self.bn:
            return self.fc_lyaer[2](torch.relu(h)).transpose(1, 2).contiguous()
        h = self.fc_lyaer[1](h, relu=True)
        return self.fc_lyaer[3](h).transpose(1, 2).contiguous()

    def _fused(self, pointcloud, ts, label, prec="fp32"):
        if not (self.use_knn_FP and self.bn and not self.bn_first):
            raise NotImplementedError("the fused plan covers the shipped latent-DDPM configuration family (use_knn_FP, bn, not bn_first); "
                                      "this configuration runs on the module path: forward(..., fused=False)")
        from slide_amd.engine import DenoiserEngine
        B = pointcloud.shape[0]
        key = (B, prec)
        if key not in self._engines:
            self._engines = {key: DenoiserEngine(self.hparams, self.state_dict(), B, pointcloud.device, prec=prec)}
        return self._engines[key].forward(pointcloud, ts, label)
