"""One level of the latent-point decoder (reference: pointnet2/models/point_upsample_decoder.py:24-190) on the HIP-backed
`pointnet2_ops` modules: a PointNet++ feature extractor on the current level's points, a cross-set attention feature
mapper that pulls the previous level's features onto them (FeatureMapModule, subset=False), and a 1x1 convolution whose
output splits every point into `point_upsample_factor` children, thinned to `num_output_points` by farthest point
sampling (pytorch3d `sample_farthest_points` in the reference; slide_amd._ext.sample_farthest_points here).

`decode_only=True` builds just the splitting head (`fc_layer`) -- all `PointAutoencoder.decode` needs from the key-point
encoder level.  Built in full, a level whose architecture has no `decoder_feature_dim` extracts its features with a
`PointNet2Encoder` (the key-point encoder of the ENCODE path), and with
`apply_kl_regularization` the extractor and the mapper emit (mean | logvar) and the level keeps the posterior mode or a
sample (reference :93-104, pointnet2/data_utils/distributions.py:4-43)."""
import copy

import torch
import torch.nn as nn

from pointnet2_ops.pointnet2_modules import FeatureMapModule
from models.point_upsample_module import point_upsample
from models.pointnet2_feature_extractor import PointNet2Encoder
from models.pointnet2_with_pcld_condition import PointNet2CloudCondition
from slide_amd import _ext as _hip
from slide_amd.nn_ops import HipConv1x1


class PointUpsampleDecoder(nn.Module):
    def __init__(self, config, in_dim, apply_kl_regularization=False, decode_only=False):
        super().__init__()
        self.hparams = hp = config
        self.apply_kl_regularization = apply_kl_regularization
        arch = hp["architecture"]
        has_fp = "decoder_feature_dim" in arch
        query_dim = arch["decoder_feature_dim"][0] if has_fp else arch["feature_dim"][-1]
        fm = hp["feature_mapper_setting"]
        up = hp["upsampling_setting"]
        self.decode_only = decode_only
        if not decode_only:
            cfg = copy.deepcopy(hp)
            kl = 2 if apply_kl_regularization else 1  # (mean | logvar) channels, reference :37-44,56-60
            if has_fp:
                cfg["architecture"]["decoder_feature_dim"][0] *= kl
                self.feature_extractor = PointNet2CloudCondition(cfg)
            else:
                cfg["architecture"]["feature_dim"][-1] *= kl
                self.feature_extractor = PointNet2Encoder(cfg)
            self.feature_mapper = FeatureMapModule(
                [in_dim] + [fm["out_dim"] * kl] * fm["mlp_depth"], fm["radius"], fm["nsample"], use_xyz=hp["model.use_xyz"],
                include_abs_coordinate=hp["include_abs_coordinate"],
                include_center_coordinate=hp.get("include_center_coordinate", False), bn=hp["bn"], bn_first=hp["bn_first"],
                bias=hp["bias"], res_connect=hp["res_connect"], first_conv=False, first_conv_in_channel=0,
                neighbor_def=fm["neighbor_definition"], activation=hp.get("activation", "relu"),
                attention_setting=hp["attention_setting"], query_feature_dim=query_dim)
        factor = up["point_upsample_factor"]
        if up["first_refine_coarse_points"]:
            factor += 0 if up["include_displacement_center_to_final_output"] else 1
        else:
            assert not up["include_displacement_center_to_final_output"]
        self.point_upsample_factor = factor
        self.upsampling_setting = up
        self.fc_layer = HipConv1x1(query_dim + fm["out_dim"] + hp["in_fea_dim"] + 3, int(hp["out_dim"] * factor), ndim=1)

    def propagate_feature(self, xyz, features, new_xyz, ts=None, label=None, sample_posterior=True):
        """xyz (B,N1,3) with features (B,N1,C1) -> features at new_xyz (B,N2,3+in_fea): [extracted | mapped]"""
        if self.decode_only:
            raise NotImplementedError("this level was built decode-only")
        out = self.feature_extractor(new_xyz, ts=ts, label=label)
        if isinstance(out, tuple):  # PointNet2Encoder returns (features, l_xyz, l_features)
            out = out[0]
        if self.apply_kl_regularization:
            out = self._posterior(out, sample_posterior)
        mapped = self.feature_mapper(xyz, features.transpose(1, 2).contiguous(), new_xyz[:, :, 0:3].contiguous(), subset=False,
                                     record_neighbor_stats=False, pooling=None,
                                     features_at_new_xyz=out.transpose(1, 2).contiguous()).transpose(1, 2)
        if self.apply_kl_regularization:
            mapped = self._posterior(mapped, sample_posterior)
        return torch.cat([out, mapped], dim=2), None

    @staticmethod
    def _posterior(parameters, sample):
        """DiagonalGaussianDistribution over the channel halves of (B,N,2C): mode, or mean + std * N(0,1)"""
        mean, logvar = torch.chunk(parameters, 2, dim=2)
        if not sample:
            return mean.contiguous()
        return mean + torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0)) * torch.randn_like(mean)

    def upsample_points(self, final_feature, new_xyz, fps_start_idx=None):
        hp, up = self.hparams, self.upsampling_setting
        split = self.fc_layer(torch.cat([final_feature, new_xyz], dim=2).transpose(1, 2).contiguous()).transpose(1, 2)
        in_dim = hp.get("in_position_and_normal_dim", hp["out_dim"])
        coarse = new_xyz[:, :, 0:in_dim]
        if in_dim < hp["out_dim"]:  # key points carry no normals: they are generated from scratch
            coarse = torch.cat([coarse, coarse.new_zeros(coarse.shape[0], coarse.shape[1], hp["out_dim"] - in_dim)], dim=2)
        pts = point_upsample(coarse, split, self.point_upsample_factor,
                             include_displacement_center_to_final_output=up["include_displacement_center_to_final_output"],
                             output_scale_factor_value=up["output_scale_factor"],
                             first_refine_coarse_points=up["first_refine_coarse_points"])
        n_out = up["num_output_points"]
        assert pts.shape[1] >= n_out
        if pts.shape[1] > n_out:
            pts, _ = _hip.sample_farthest_points(pts.contiguous(), K=n_out, random_start_point=fps_start_idx is None,
                                                 start_idx=fps_start_idx)
        return pts

    @torch.no_grad()
    def forward(self, xyz, features, new_xyz, ts=None, label=None, sample_posterior=True, fps_start_idx=None):
        feat, _ = self.propagate_feature(xyz, features, new_xyz, ts=ts, label=label, sample_posterior=sample_posterior)
        return feat, self.upsample_points(feat, new_xyz, fps_start_idx)
