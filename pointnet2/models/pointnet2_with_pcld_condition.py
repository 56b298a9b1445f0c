"""`PointNet2CloudCondition` -- the per-timestep denoiser (reference: pointnet2/models/pointnet2_with_pcld_condition.py:
27-489 on top of pointnet2/models/pointnet2_ssg_sem.py:34-177) assembled from the HIP-backed `pointnet2_ops` modules, with
the reference's attribute / state-dict names, for the configuration family every shipped
latent-DDPM config uses (no condition cloud: include_local_feature / include_global_feature False), and -- round 6, module path --
its CONDITION-CLOUD form (reference :94-260, :301-447): a second PointNet++ over the condition cloud whose encoder / decoder levels
feed the noisy cloud's levels through feature-transfer modules (`include_local_feature`), and / or a `Pnet2Stage` global feature of
the condition cloud in the Mlps' first condition slot (`include_global_feature`), with the retained-feature path a sampler uses
(`use_retained_condition_feature`, `reset_cond_features`); pinned by tests/golden/golden_denoiser_condition.npz.  The parent
project's other switches of the class (swish, position encoding, global attention per level, up-sampling head, condition cloud
concatenated behind an indicator channel): tests/golden/golden_denoiser_switches.npz.

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

from pointnet2_ops.pointnet2_modules import FeatureMapModule, PointnetFPModule, PointnetKnnFPModule, PointnetSAModule, Swish
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
        arch = hp["architecture"]
        self.bn, self.bn_first, self.use_knn_FP = hp.get("bn", True), hp["bn_first"], arch.get("use_knn_FP", False)
        self.include_local_feature = hp.get("include_local_feature", True)
        self.include_global_feature = hp.get("include_global_feature", False)
        # the parent project's switches (reference :47-93, :119-126, :243-257): activation, NeRF-style position encoding of the
        # coordinates as extra input features, the condition cloud concatenated to the noisy one (an indicator channel tells them
        # apart), an up-sampling output head, global attention behind chosen levels
        self.activation = hp.get("activation", "relu")
        assert self.activation in ("relu", "swish")
        self.concat_partial = hp.get("concate_partial_with_noisy_input", False)
        assert not (self.concat_partial and (self.include_local_feature or self.include_global_feature))
        self.pe_freqs = None
        if hp.get("use_position_encoding", False):
            self.pe_freqs = [float(2.0 ** k) for k in np.linspace(0.0, hp["position_encoding_multires"] - 1, hp["position_encoding_multires"])]
        pe = 0 if self.pe_freqs is None else 6 * len(self.pe_freqs)
        self.gatt = hp.get("global_attention_setting", None)
        up = hp.get("point_upsample_factor", 1)
        if up > 1:
            if hp["first_refine_coarse_points"]:
                up = up + (0 if hp["include_displacement_center_to_final_output"] else 1)
            else:
                assert not hp["include_displacement_center_to_final_output"]
        self.out_dim = int(hp["out_dim"] * up)
        self.pooling = hp.get("pooling", "max")
        self.att = att = hp.get("attention_setting", None)
        t_dim = hp["t_dim"]
        self.class_emb = nn.Embedding(hp["num_class"], hp["class_condition_dim"]) if hp["include_class_condition"] else None
        attach = 3 if hp["attach_position_to_input_feature"] else 0
        in_fea = hp["in_fea_dim"] + attach + pe
        self.partial_in_fea_dim = cin = hp.get("partial_in_fea_dim", hp["in_fea_dim"]) + attach + pe  # condition cloud's feature channels
        self.fc_t1, self.fc_t2 = HipLinear(t_dim, 4 * t_dim), HipLinear(4 * t_dim, 4 * t_dim)
        # ---- global feature of the condition cloud: it takes the first condition slot of every Mlp, the class embedding the second
        gdim = None
        if self.include_global_feature:
            from models.pnet import Pnet2Stage
            pa = hp["pnet_global_feature_architecture"]
            gdim = pa[1][-1]
            self.global_pnet = Pnet2Stage([pa[0][0] + pe] + list(pa[0][1:]), pa[1], bn=self.bn,
                                          remove_last_activation=hp.get("global_feature_remove_last_activation", True))
        cls = hp["include_class_condition"]
        self._geo = geo = dict(use_xyz=hp["model.use_xyz"], include_abs_coordinate=hp["include_abs_coordinate"],
                               include_center_coordinate=hp.get("include_center_coordinate", False))
        self._mlp = mlp_kw = dict(bn=self.bn, bn_first=self.bn_first, res_connect=hp["res_connect"], bias=hp["bias"],
                                  activation=self.activation)
        if gdim is not None:
            noisy_cond = dict(include_condition=True, condition_dim=gdim, include_second_condition=cls,
                              second_condition_dim=hp["class_condition_dim"])
        else:
            noisy_cond = dict(include_condition=cls, condition_dim=hp["class_condition_dim"])
        noisy = dict(t_dim=4 * t_dim, include_t=hp["include_t"], **noisy_cond)
        plain = dict(t_dim=4 * t_dim, include_t=False, include_condition=False)  # the condition cloud's own network: no t, no class
        f, depth = arch["feature_dim"], arch["mlp_depth"]
        d, ddepth = arch["decoder_feature_dim"], arch["decoder_mlp_depth"]
        assert d[-1] == f[-1]
        enc_map = dec_map = None
        if self.include_local_feature:
            ca, ma = hp["condition_net_architecture"], hp["feature_mapper_architecture"]
            cf, cd = ca["feature_dim"], ca["decoder_feature_dim"]
            assert cd[-1] == cf[-1] and len(ca["npoint"]) == len(arch["npoint"])
            enc_map, dec_map = ma["encoder_feature_map_dim"], ma["decoder_feature_map_dim"]
            self.SA_modules_condition = self._sa_stack(ca, cf, ca["mlp_depth"], cin, None, att, plain)
            # feature transfer: condition cloud level i -> the noisy cloud's level i, queried with the noisy points' own features
            fm_att = None if att is None else dict(att, use_attention_module=att["add_attention_to_FeatureMapper_module"])
            self.encoder_feature_map = nn.ModuleList()
            for i, od in enumerate(enc_map):
                fc0 = self.bn_first and i == 0
                ind = cin if (i == 0 and not fc0) else cf[i]
                self.encoder_feature_map.append(FeatureMapModule(
                    [ind] + [od] * ma["encoder_mlp_depth"], ma["encoder_radius"][i], ma["encoder_nsample"][i], first_conv=fc0,
                    first_conv_in_channel=cin, neighbor_def=ma["neighbor_definition"], attention_setting=fm_att,
                    query_feature_dim=in_fea if i == 0 else f[i], **geo, **mlp_kw))
        self.SA_modules = self._sa_stack(arch, f, depth, in_fea, enc_map, att, noisy)
        if self.include_local_feature:
            self.FP_modules_condition = self._fp_stack(ca, cd, ca["decoder_mlp_depth"], cf, cin, None, att, plain)
            self.decoder_feature_map = nn.ModuleList()
            for i, od in enumerate(dec_map):
                self.decoder_feature_map.append(FeatureMapModule(
                    [cd[i]] + [od] * ma["decoder_mlp_depth"], ma["decoder_radius"][i], ma["decoder_nsample"][i], first_conv=False,
                    first_conv_in_channel=0, neighbor_def=ma["neighbor_definition"], attention_setting=fm_att,
                    query_feature_dim=d[i], **geo, **mlp_kw))
        self.FP_modules = self._fp_stack(arch, d, ddepth, f, in_fea, None if dec_map is None else dec_map[1:], att, noisy)
        self.transform_output = hp.get("transform_output", True)
        if self.transform_output:  # (layer positions as in the reference's Sequentials: state-dict keys fc_lyaer.<index>.*)
            hin = d[0] + 3 + (dec_map[0] if dec_map is not None else 0)
            act = (lambda: nn.ReLU(True)) if self.activation == "relu" else Swish
            if self.bn_first:
                self.fc_lyaer = nn.Sequential(act(), HipConv1x1(hin, self.out_dim, ndim=1))
            elif self.bn:
                self.fc_lyaer = nn.Sequential(HipConv1x1(hin, 128, bias=hp["bias"], ndim=1), HipGroupNorm(32, 128),
                                              act(), HipConv1x1(128, self.out_dim, ndim=1))
            else:
                self.fc_lyaer = nn.Sequential(HipConv1x1(hin, 128, bias=hp["bias"], ndim=1), act(),
                                              HipConv1x1(128, self.out_dim, ndim=1))
        self._engines = {}
        self.reset_cond_features()

    # ---- module stacks (reference: pointnet2_ssg_sem.py:44-177)
    def _sa_stack(self, a, f, depth, in_dim, extra, att, emb):
        """set-abstraction levels; extra[i] = channels the level's input gains from the condition cloud (feature transfer)"""
        out = nn.ModuleList()
        nb = a["neighbor_definition"]
        for i in range(len(a["npoint"])):
            first_conv = self.bn_first and i == 0  # (bn_first: a convolution ahead of the first GroupNorm)
            e = 0 if extra is None else extra[i]
            c0 = in_dim + e if i == 0 else f[i] + e
            spec = [f[i] + e if (i == 0 and first_conv) else c0] + [f[i]] * (depth - 1) + [f[i + 1]]
            out.append(PointnetSAModule(npoint=a["npoint"][i], radius=a["radius"][i], nsample=a["nsample"][i], mlp=spec,
                                        first_conv=first_conv, first_conv_in_channel=in_dim + e, attention_setting=att,
                                        global_attention_setting=self._gatt_at(i) if a is self.hparams["architecture"] else None,
                                        neighbor_def=nb[i] if isinstance(nb, list) else nb, **self._geo, **self._mlp, **emb))
        return out

    def _fp_stack(self, a, d, ddepth, f, in_dim, extra, att, emb):
        out = nn.ModuleList()
        nb = a["neighbor_definition"]
        for i in range(len(d) - 1):
            skip = in_dim if i == 0 else f[i]
            e = 0 if extra is None else extra[i]
            kw = dict(first_conv=False, include_grouper=a.get("include_grouper", False), radius=a["radius"][i], nsample=a["nsample"][i],
                      neighbor_def=nb[i] if isinstance(nb, list) else nb, **self._geo, **self._mlp, **emb)
            if a.get("use_knn_FP", False):
                out.append(PointnetKnnFPModule(mlp1=[d[i + 1] + e] + [d[i]] * ddepth, mlp2=[d[i] + skip] + [d[i]] * ddepth,
                                               K=a.get("K", 3), attention_setting=att,
                                               global_attention_setting=self._gatt_at(i) if a is self.hparams["architecture"] else None, **kw))
            else:  # three nearest known points, inverse-distance weights, ONE Mlp
                out.append(PointnetFPModule(mlp=[d[i + 1] + skip + e] + [d[i]] * ddepth, **kw))
        return out

    def _gatt_at(self, i):
        g = self.gatt
        return g if (g is not None and g["use_global_attention_module"] and i in g["global_attention_layer_index"]) else None

    def _position_code(self, xyz):
        """[sin(f x), cos(f x)] over the octave frequencies (reference models/model_utils.py:3-51; input not included)"""
        return torch.cat([fn(xyz * f) for f in self.pe_freqs for fn in (torch.sin, torch.cos)], dim=-1)

    def reset_cond_features(self):
        """forget the retained condition-cloud features (use_retained_condition_feature: a sampler calls the network 1000 times with
        ONE condition cloud, whose own network does not see t)"""
        self.l_uvw = self.encoder_cond_features = self.decoder_cond_features = self.global_feature = None

    def _break_up_pc(self, pc):
        return pc[..., 0:3].contiguous(), (pc[..., 3:].transpose(1, 2).contiguous() if pc.size(-1) > 3 else None)

    @torch.no_grad()
    def forward(self, pointcloud, condition=None, ts=None, label=None, use_retained_condition_feature=False, fused=False):
        hp = self.hparams
        local, glob = self.include_local_feature, self.include_global_feature
        assert (condition is not None) == (local or glob or self.concat_partial), "a condition cloud is given exactly when the configuration uses one"
        if fused:
            return self._fused(pointcloud, ts, label)
        keep = use_retained_condition_feature
        attach = hp["attach_position_to_input_feature"]
        n_own = pointcloud.shape[1]
        if self.concat_partial:  # ONE cloud: the noisy points (indicator 0) followed by the condition points (indicator 1)
            assert pointcloud.shape[2] == 3 and condition.shape[2] in (3, 4) and condition.shape[0] == pointcloud.shape[0]
            tag = lambda c, v: torch.cat([c, torch.full_like(c[:, :, :1], v)], dim=2)
            pointcloud = torch.cat([tag(pointcloud, 0.0), condition if condition.shape[2] == 4 else tag(condition, 1.0)], dim=1)
            condition = None
        code = (lambda c: torch.cat([c, self._position_code(c[:, :, 0:3])], dim=2)) if self.pe_freqs is not None else (lambda c: c)
        pc = code(pointcloud)
        pc = torch.cat([pc, pointcloud[:, :, 0:3]], dim=2) if attach else pc
        xyz, features = self._break_up_pc(pc)
        if condition is not None:
            cpc = code(condition)
            cpc = torch.cat([cpc, condition[:, :, 0:3]], dim=2) if attach else cpc
            uvw, cond_features = self._break_up_pc(cpc)
        t_emb = None
        if ts is not None and hp["include_t"]:
            t_emb = swish(self.fc_t2(swish(self.fc_t1(calc_t_emb(ts, hp["t_dim"])))))
        cls = self.class_emb(label) if (label is not None and self.class_emb is not None) else None
        if glob:
            if keep and self.global_feature is not None:
                gfeat = self.global_feature
            else:
                own = self.partial_in_fea_dim - (3 if attach else 0)  # the condition cloud's own feature channels (+ its position code)
                gin = torch.cat([uvw, cpc[:, :, 3:3 + own]], dim=2) if own > 0 else uvw
                gfeat = self.global_pnet(gin.transpose(1, 2).contiguous())
                if keep:
                    self.global_feature = gfeat.detach().clone()
            emb = dict(t_emb=t_emb, condition_emb=gfeat, second_condition_emb=cls if hp["include_class_condition"] else None)
        else:
            emb = dict(t_emb=t_emb, condition_emb=cls)
        pool = dict(pooling=self.pooling)
        # ---- encoder: the condition cloud's levels run beside the noisy cloud's; level i of the noisy cloud reads level i of the
        # condition cloud through a feature-transfer module queried with its own features
        retained_enc = keep and self.encoder_cond_features is not None
        retained_dec = keep and self.decoder_cond_features is not None
        if local:
            l_uvw, l_cf = ([uvw], [cond_features]) if not retained_enc else (self.l_uvw, self.encoder_cond_features)
        l_xyz, l_features = [xyz], [features]
        for i, m in enumerate(self.SA_modules):
            x_in = l_features[i]
            if local:
                if not retained_enc:
                    nu, nf = self.SA_modules_condition[i](l_uvw[i], l_cf[i], subset=True, **pool)
                    l_uvw.append(nu); l_cf.append(nf)
                mapped = self.encoder_feature_map[i](l_uvw[i], l_cf[i], l_xyz[i], subset=False, features_at_new_xyz=l_features[i], **pool)
                x_in = torch.cat([mapped, l_features[i]], dim=1)
            nx, nf = m(l_xyz[i], x_in, subset=True, **emb, **pool)
            l_xyz.append(nx); l_features.append(nf)
        if local and keep and not retained_enc:
            self.l_uvw, self.encoder_cond_features = l_uvw, [None if v is None else v.clone() for v in l_cf]
        # ---- decoder
        if local:
            l_cd = list(l_cf) if not retained_dec else self.decoder_cond_features
        for i in range(-1, -(len(self.FP_modules) + 1), -1):
            k_in = l_features[i]
            if local:
                if not retained_dec:
                    l_cd[i - 1] = self.FP_modules_condition[i](l_uvw[i - 1], l_uvw[i], l_cd[i - 1], l_cd[i], **pool)
                mapped = self.decoder_feature_map[i](l_uvw[i], l_cd[i], l_xyz[i], subset=False, features_at_new_xyz=l_features[i], **pool)
                k_in = torch.cat([mapped, l_features[i]], dim=1)
            l_features[i - 1] = self.FP_modules[i](l_xyz[i - 1], l_xyz[i], l_features[i - 1], k_in, **emb, **pool)
        out = l_features[0]
        if local:
            if keep and not retained_dec:
                self.decoder_cond_features = [None if v is None else v.clone() for v in l_cd]
            mapped = self.decoder_feature_map[0](l_uvw[0], l_cd[0], l_xyz[0], subset=False, features_at_new_xyz=l_features[0], **pool)
            out = torch.cat([mapped, l_features[0]], dim=1)
        if not self.transform_output:  # feature-extractor use (autoencoder decoder levels): per-point features
            y = out.transpose(1, 2).contiguous()
            return y[:, :n_own].contiguous() if self.concat_partial else y
        out = torch.cat([out, xyz.transpose(1, 2)], dim=1)
        relu = self.activation == "relu"
        act = torch.relu if relu else swish
        if self.bn_first:
            y = self.fc_lyaer[1](act(out))
        else:
            h = self.fc_lyaer[0](out)
            if not self.bn:
                y = self.fc_lyaer[2](act(h))
            else:
                h = self.fc_lyaer[1](h, relu=relu)
                y = self.fc_lyaer[3](h if relu else swish(h))
        y = y.transpose(1, 2).contiguous()
        return y[:, :n_own].contiguous() if self.concat_partial else y

    def _fused(self, pointcloud, ts, label, prec="fp32"):
        if (not (self.use_knn_FP and self.bn and not self.bn_first) or self.include_local_feature or self.include_global_feature
                or self.activation != "relu" or self.pe_freqs is not None or self.concat_partial or self.gatt or self.out_dim != self.hparams["out_dim"]):
            raise NotImplementedError("the fused plan covers the shipped latent-DDPM configuration family (use_knn_FP, bn, not bn_first, no condition cloud); "
                                      "this configuration runs on the module path: forward(..., fused=False)")
        from slide_amd.engine import DenoiserEngine
        B = pointcloud.shape[0]
        key = (B, prec)
        if key not in self._engines:
            self._engines = {key: DenoiserEngine(self.hparams, self.state_dict(), B, pointcloud.device, prec=prec)}
        return self._engines[key].forward(pointcloud, ts, label)
