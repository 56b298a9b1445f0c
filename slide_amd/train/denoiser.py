"""Trainable PointNet2CloudCondition (pointnet2/models/pointnet2_with_pcld_condition.py:286-489) on the differentiable row-major
layers of functions.py -- the configuration family every shipped DDPM config uses (16 latent points <= npoint: no FPS; 'nn' grouping
over all points; kNN feature propagation with K = 8; attention aggregation; MyGroupNorm; bias; res_connect; no condition cloud).

Parameters carry the reference's state-dict names, so reference checkpoints load with load_state_dict and
trained weights drop into the fused sampling engine (slide_amd.engine.DenoiserEngine) unchanged.  Activations are fp32 rows
[B * S, ld]: S = 256 (SA blocks: 16 points x 16 neighbours), 128 (kNN-FP blocks: 16 x 8) or 16 (per-point layers)."""
import numpy as np
import torch
import torch.nn as nn

from .. import _ext, model_spec
from ..rows import GROUP_ABS, GROUP_CENTER, GROUP_FP
from . import functions as F


def _swish(x):
    return x * torch.sigmoid(x)  # pointnet2/models/pointnet2_ssg_sem.py:9-10


class TrainableDenoiser(nn.Module):
    NP = 16

    def __init__(self, hp, state_dict=None):
        super().__init__()
        arch = hp["architecture"]
        assert not hp.get("include_local_feature", True) and not hp.get("include_global_feature", False)
        assert arch["neighbor_definition"] == "nn" and arch.get("use_knn_FP", False) and not arch.get("include_grouper", False)
        assert hp["attach_position_to_input_feature"] and hp["include_abs_coordinate"] and hp.get("include_center_coordinate", False)
        assert hp["bias"] and hp["res_connect"] and not hp["bn_first"] and hp.get("bn", True) and hp["include_t"]
        assert all(n >= self.NP for n in arch["npoint"]) and all(ns >= self.NP for ns in arch["nsample"]) and arch["K"] == 8
        self.hp = hp
        self._names = []
        for name, shape in model_spec.denoiser_param_spec(hp):
            self._register(name, nn.Parameter(torch.zeros(*shape)))
        if state_dict is not None:
            self.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in state_dict.items()})

    # parameters live on a tree of plain containers so that state_dict() yields the reference's dotted names
    def _register(self, name, param):
        mod = self
        parts = name.split(".")
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, nn.Module())
            mod = mod._modules[p]
        mod.register_parameter(parts[-1], param)
        self._names.append(name)

    def _p(self, name):
        mod = self
        parts = name.split(".")
        for p in parts[:-1]:
            mod = mod._modules[p]
        return mod._parameters[parts[-1]]

    def _has(self, name):
        return name in self._names

    def reset_parameters(self, seed=0):
        """PyTorch's default initialisation of the reference's layers (the reference trains from it, pointnet2/train.py:100-112):
        Conv / Linear weights and biases U(-1/sqrt(fan_in), 1/sqrt(fan_in)), GroupNorm weight 1 / bias 0, Embedding N(0, 1)"""
        g = torch.Generator().manual_seed(int(seed))
        fan = {}
        with torch.no_grad():
            for name in self._names:
                p = self._p(name)
                if "group_norm" in name or name.startswith("fc_lyaer.1."):
                    p.fill_(1.0 if name.endswith(".weight") else 0.0)
                elif name == "class_emb.weight":
                    p.copy_(torch.randn(p.shape, generator=g))
                elif name.endswith(".weight"):
                    fan[name[:-7]] = int(np.prod(p.shape[1:]))
                    b = 1.0 / np.sqrt(fan[name[:-7]])
                    p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * b)
            for name in self._names:
                if name.endswith(".bias") and name[:-5] in fan:
                    p = self._p(name)
                    p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) / np.sqrt(fan[name[:-5]]))
        return self

    # ------------------------------------------------------------------ layers
    def _shared(self, x, prefix, B, S):
        """build_shared_mlp stage (pointnet2_modules.py:44-69): Conv(1x1, bias) -> MyGroupNorm -> ReLU"""
        y = F.conv_rows(x, self._p(prefix + ".0.weight"), self._p(prefix + ".0.bias"))
        C = self._p(prefix + ".0.weight").shape[0]
        gam = self._p(prefix + ".1.group_norm.weight")
        return F.gn_rows(y, gam, self._p(prefix + ".1.group_norm.bias"), B, S, min(32, C), False, True)

    @staticmethod
    def _add_vec(x, vec, B, S):
        """x[(b, s)][c] += vec[b][c] (the t- / class-embedding terms: one vector per sample)"""
        v = F.pad_cols(vec, x.shape[1])
        return (x.view(B, S, -1) + v[:, None, :]).reshape(B * S, -1)

    def _mlp(self, x, prefix, B, S, t_emb, cond_emb):
        """Mlp_plus_t_emb.forward (pointnet2_modules.py:119-176)"""
        h = self._shared(x, prefix + ".first_mlp", B, S)
        if self._has(prefix + ".fc.weight"):
            h = self._add_vec(h, torch.nn.functional.linear(t_emb, self._p(prefix + ".fc.weight"), self._p(prefix + ".fc.bias")), B, S)
        h = self._shared(h, prefix + ".second_mlp", B, S)
        if self._has(prefix + ".fc_condition.weight"):
            h = self._add_vec(h, torch.nn.functional.linear(cond_emb, self._p(prefix + ".fc_condition.weight"),
                                                            self._p(prefix + ".fc_condition.bias")), B, S)
        if self._has(prefix + ".rest_mlp.0.weight"):
            h = self._shared(h, prefix + ".rest_mlp", B, S)
        assert self._has(prefix + ".res_connect.weight"), "identity res_connect is not used by the shipped DDPM configs"
        return h + F.conv_rows(x, self._p(prefix + ".res_connect.weight"), self._p(prefix + ".res_connect.bias"))

    def _attention(self, feat, grouped, out, prefix, B, K):
        """AttentionModule.forward (attention.py:70-96), 'nn' grouping: every neighbour counts"""
        S = self.NP * K
        q = F.conv_rows(feat, self._p(prefix + ".feat_conv.weight"), self._p(prefix + ".feat_conv.bias"))
        k = F.conv_rows(grouped, self._p(prefix + ".grouped_feat_conv.weight"), self._p(prefix + ".grouped_feat_conv.bias"))
        C1, C2 = self._p(prefix + ".feat_conv.weight").shape[0], self._p(prefix + ".grouped_feat_conv.weight").shape[0]
        s = F.concat_qk(q, k, K, C1, C2)                                             # relu(cat([q.expand, k]))
        s = F.gn_rows(s, self._p(prefix + ".weight_conv.1.group_norm.weight"), self._p(prefix + ".weight_conv.1.group_norm.bias"),
                      B, S, min(32, C1 + C2), False, False)
        s = F.conv_rows(s, self._p(prefix + ".weight_conv.2.weight"), self._p(prefix + ".weight_conv.2.bias"))
        inter = self._p(prefix + ".weight_conv.2.weight").shape[0]
        s = F.gn_rows(s, self._p(prefix + ".weight_conv.4.group_norm.weight"), self._p(prefix + ".weight_conv.4.group_norm.bias"),
                      B, S, min(32, inter), True, False)                             # ReLU, then MyGroupNorm
        scores = F.conv_rows(s, self._p(prefix + ".weight_conv.5.weight"), self._p(prefix + ".weight_conv.5.bias"))
        cout = self._p(prefix + ".weight_conv.5.weight").shape[0]
        v = F.conv_rows(out, self._p(prefix + ".feat_out_conv.0.weight"), self._p(prefix + ".feat_out_conv.0.bias"))
        v = F.gn_rows(v, self._p(prefix + ".feat_out_conv.1.group_norm.weight"), self._p(prefix + ".feat_out_conv.1.group_norm.bias"),
                      B, S, min(32, cout), False, True)
        return F.attend_rows(scores, v, K, cout), cout

    # ------------------------------------------------------------------ forward
    def forward(self, pointcloud, ts, label):
        """pointcloud (B, 16, 3 + in_fea_dim), ts (B,), label (B,) -> (B, 16, out_dim)"""
        hp, arch = self.hp, self.hp["architecture"]
        B, N = pointcloud.shape[:2]
        assert N == self.NP
        pc = pointcloud.float()
        pc = torch.cat([pc, pc[:, :, 0:3]], dim=2)                                   # :332-334 (scale factor 1)
        xyz = pc[:, :, 0:3].contiguous()
        C0 = pc.shape[2] - 3
        feat0 = F.pad_cols(pc[:, :, 3:].reshape(B * N, C0))
        # t-embedding (pointnet2_ssg_sem.py:14-31) -> fc_t1 -> swish -> fc_t2 -> swish; class embedding
        half = hp["t_dim"] // 2
        freq = torch.exp(torch.arange(half, device=pc.device, dtype=torch.float32) * -(np.log(10000) / (half - 1)))
        arg = ts.float()[:, None] * freq[None]
        t_emb = torch.cat([torch.sin(arg), torch.cos(arg)], dim=1)
        t_emb = _swish(torch.nn.functional.linear(t_emb, self._p("fc_t1.weight"), self._p("fc_t1.bias")))
        t_emb = _swish(torch.nn.functional.linear(t_emb, self._p("fc_t2.weight"), self._p("fc_t2.bias")))
        cond = self._p("class_emb.weight")[label.long()]
        # one sorted 16 x 16 neighbour table serves every block (all levels hold the same 16 points); K = 8 is its prefix
        with torch.no_grad():
            d2, idx = _ext.knn_points(xyz, xyz, self.NP, None)
            idx8, d28 = idx[:, :, :8].contiguous(), d2[:, :, :8].contiguous()
        feats, chans = [feat0], [C0]
        nsa, nfp = len(arch["npoint"]), len(arch["decoder_feature_dim"]) - 1
        for i in range(nsa):                                                           # PointnetSAModule (pointnet2_modules.py:222-292)
            pfx = "SA_modules.%d" % i
            g = F.group_rows(feats[i], xyz, xyz, idx, None, GROUP_ABS | GROUP_CENTER, chans[i])
            out = self._mlp(g, pfx + ".mlps.0", B, 256, t_emb, cond)
            o, c = self._attention(feats[i], g, out, pfx + ".attention_modules.0", B, 16)
            feats.append(o); chans.append(c)
        for i in range(-1, -(nfp + 1), -1):                                            # PointnetKnnFPModule (:771-873)
            pfx = "FP_modules.%d" % (nfp + i)
            U, CU, Kf, C2 = feats[i - 1], chans[i - 1], feats[i], chans[i]
            g = F.group_rows(Kf, xyz, xyz, idx8, d28, GROUP_FP, C2)
            out = self._mlp(g, pfx + ".mlp1", B, 128, None, None)
            interp, c = self._attention(U, g, out, pfx + ".attention_module", B, 8)
            z = F.pad_cols(torch.cat([interp[:, :c], U[:, :CU], xyz.reshape(B * N, 3)], dim=1))
            o = self._mlp(z, pfx + ".mlp2", B, 16, t_emb, cond)
            feats[i - 1], chans[i - 1] = o, self._p(pfx + ".mlp2.res_connect.weight").shape[0]
        # fc_lyaer: Conv1d -> GroupNorm(32, 128) -> ReLU -> Conv1d (:480-483)
        h = F.pad_cols(torch.cat([feats[0][:, :chans[0]], xyz.reshape(B * N, 3)], dim=1))
        h = F.conv_rows(h, self._p("fc_lyaer.0.weight"), self._p("fc_lyaer.0.bias"))
        h = F.gn_rows(h, self._p("fc_lyaer.1.weight"), self._p("fc_lyaer.1.bias"), B, 16, 32, False, True)
        h = F.conv_rows(h, self._p("fc_lyaer.3.weight"), self._p("fc_lyaer.3.bias"))
        return h[:, :hp["out_dim"]].reshape(B, N, hp["out_dim"])
