"""`PointNet2Encoder` (reference: pointnet2/models/pointnet2_feature_extractor.py:25-218) on the HIP-backed `pointnet2_ops`
modules: a stack of set-abstraction modules (farthest-point down-sampling, kNN grouping, shared Mlp, vector attention)
conditioned on the class embedding and, when configured, on a `Pnet2Stage` global feature.  Used twice by the autoencoder's
ENCODE path: on the 2048-point input cloud (2048 -> 1024 -> 256 -> 64 -> 32 points, K = 32)
and on the 16 key points inside `PointUpsampleDecoder.propagate_feature`.  Parameter names equal the reference's
(`class_emb`, `global_pnet.*`, `fc_t1/2`, `SA_modules.{i}.*`, `fc_lyaer.*`)."""
import numpy as np
import torch
import torch.nn as nn

from pointnet2_ops.pointnet2_modules import PointnetSAModule
from models.pnet import Pnet2Stage
from models.pointnet2_with_pcld_condition import calc_t_emb, swish
from slide_amd.nn_ops import HipConv1x1, HipLinear


class PointNet2Encoder(nn.Module):
    def __init__(self, hparams):
        super().__init__()
        self.hparams = hp = hparams
        # the parent project's switches (round 6; no shipped configuration sets them; reference :33-58, :96-106): activation, NeRF-style
        # position encoding of the coordinates, bn_first (a convolution ahead of the first GroupNorm), global attention per level
        self.activation = hp.get("activation", "relu")
        assert self.activation in ("relu", "swish")
        self.bn_first = hp["bn_first"]
        self.pe_freqs = None
        if hp.get("use_position_encoding", False):
            m = hp["position_encoding_multires"]
            self.pe_freqs = [float(2.0 ** k) for k in np.linspace(0.0, m - 1, m)]
        pe = 0 if self.pe_freqs is None else 6 * len(self.pe_freqs)
        gatt = hp.get("global_attention_setting", None)
        arch = hp["architecture"]
        if hp["include_class_condition"]:
            self.class_emb = nn.Embedding(hp["num_class"], hp["class_condition_dim"])
        self.in_fea_dim = hp["in_fea_dim"] + (3 if hp["attach_position_to_input_feature"] else 0) + pe
        self.include_global_feature = hp.get("include_global_feature", False)
        if self.include_global_feature and pe:
            raise NotImplementedError("position encoding with the global feature: the reference widens the global PointNet's input twice "
                                      "(pointnet2_feature_extractor.py:73-78) and its forward cannot run")
        gdim = None
        if self.include_global_feature:
            pa = hp["pnet_global_feature_architecture"]
            if pa[0][0] != self.in_fea_dim:  # the reference corrects the configured input width in place (:73-75)
                pa[0][0] = self.in_fea_dim
            gdim = pa[1][-1]
            self.global_pnet = Pnet2Stage(pa[0], pa[1], bn=hp.get("bn", True),
                                          remove_last_activation=hp.get("global_feature_remove_last_activation", True))
        t_dim = hp["t_dim"]
        self.fc_t1, self.fc_t2 = HipLinear(t_dim, 4 * t_dim), HipLinear(4 * t_dim, 4 * t_dim)
        if self.include_global_feature:
            cond = dict(include_condition=True, condition_dim=gdim, include_second_condition=hp["include_class_condition"],
                        second_condition_dim=hp["class_condition_dim"])
        else:
            cond = dict(include_condition=hp["include_class_condition"], condition_dim=hp["class_condition_dim"],
                        include_second_condition=False, second_condition_dim=None)
        f, depth = arch["feature_dim"], arch["mlp_depth"]
        nd = arch["neighbor_definition"]
        self.SA_modules = nn.ModuleList()
        for i in range(len(arch["npoint"])):
            first_conv = self.bn_first and i == 0
            spec = [self.in_fea_dim if (i == 0 and not first_conv) else f[i]] + [f[i]] * (depth - 1) + [f[i + 1]]
            ga = gatt if (gatt is not None and gatt["use_global_attention_module"] and i in gatt["global_attention_layer_index"]) else None
            self.SA_modules.append(PointnetSAModule(
                npoint=arch["npoint"][i], radius=arch["radius"][i], nsample=arch["nsample"][i], mlp=spec,
                use_xyz=hp["model.use_xyz"], t_dim=4 * t_dim, include_t=hp["include_t"],
                include_abs_coordinate=hp["include_abs_coordinate"],
                include_center_coordinate=hp.get("include_center_coordinate", False), bn_first=self.bn_first, first_conv=first_conv,
                first_conv_in_channel=self.in_fea_dim, res_connect=hp["res_connect"], bias=hp["bias"],
                neighbor_def=nd[i] if isinstance(nd, list) else nd, bn=hp.get("bn", True), activation=self.activation,
                attention_setting=hp.get("attention_setting", None), global_attention_setting=ga, **cond))
        self.transform_output = hp.get("transform_output", False)
        if self.transform_output:
            self.fc_lyaer = nn.Sequential(HipConv1x1(f[-1], hp["out_dim"], ndim=1))

    def _break_up_pc(self, pc):
        return pc[..., 0:3].contiguous(), (pc[..., 3:].transpose(1, 2).contiguous() if pc.size(-1) > 3 else None)

    @torch.no_grad()
    def forward(self, pointcloud, ts=None, label=None):
        """pointcloud (B,N,3+C) -> (last-level features (B,n,C'), l_xyz, l_features)"""
        hp = self.hparams
        pc = pointcloud
        if self.pe_freqs is not None:  # [sin(f x), cos(f x)] over the octave frequencies (reference models/model_utils.py:3-51)
            x3 = pointcloud[:, :, 0:3]
            pc = torch.cat([pc] + [fn(x3 * fq) for fq in self.pe_freqs for fn in (torch.sin, torch.cos)], dim=2)
        pc = torch.cat([pc, pointcloud[:, :, 0:3]], dim=2) if hp["attach_position_to_input_feature"] else pc
        xyz, features = self._break_up_pc(pc)
        t_emb = None
        if ts is not None and hp["include_t"]:
            t_emb = swish(self.fc_t2(swish(self.fc_t1(calc_t_emb(ts, hp["t_dim"])))))
        class_emb = self.class_emb(label) if (label is not None and hp["include_class_condition"]) else None
        cond, second = class_emb, None
        if self.include_global_feature:
            gin = xyz if hp["in_fea_dim"] == 0 else torch.cat([xyz, pointcloud[:, :, 3:3 + hp["in_fea_dim"]]], dim=2)
            cond, second = self.global_pnet(gin.transpose(1, 2).contiguous()), class_emb
        l_xyz, l_features = [xyz], [features]
        for i, m in enumerate(self.SA_modules):
            nx, nf = m(l_xyz[i], l_features[i], t_emb=t_emb, condition_emb=cond, second_condition_emb=second, subset=True,
                       record_neighbor_stats=hp.get("record_neighbor_stats", False), pooling=hp.get("pooling", "max"))
            l_xyz.append(nx); l_features.append(nf)
        out = l_features[-1].transpose(1, 2).contiguous()  # (the reference's transform_output branch discards its result)
        return out, l_xyz, l_features
