"""`PointAutoencoder` (reference: pointnet2/models/autoencoder.py:11-45).  DECODE: 16 key points + 48-dim latent features ->
256 -> 1024 -> 2048 points x 6 (xyz + normal).  ENCODE (built when an encoder config is given; SURVEY.md section 8(f) item
1): 2048 x 6 input cloud -> `PointNet2Encoder` (2048 -> 1024 -> 256 -> 64 -> 32 points, K = 32) -> the key-point encoder's
`propagate_feature` -> 48-dim latent features at the 16 key points.  Parameter names follow the reference (`encoder.*`,
`keypoint_encoder.*`, `decoder.decoders.{i}.*`) so `load_state_dict(ckpt['model_state_dict'])` fills it from a released
checkpoint.  Training (`forward`, losses) is out of scope."""
import torch
import torch.nn as nn

from models.keypoint_decoder import KeypointDecoder, level_feature_dim
from models.point_upsample_decoder import PointUpsampleDecoder
from models.pointnet2_feature_extractor import PointNet2Encoder


class PointAutoencoder(nn.Module):
    def __init__(self, encoder_config, decoder_config_list, apply_kl_regularization=False, kl_weight=0, feature_weight=None):
        super().__init__()
        self.apply_kl_regularization, self.kl_weight, self.feature_weight = apply_kl_regularization, kl_weight, feature_weight
        enc_dim = encoder_config["architecture"]["feature_dim"][-1] if encoder_config is not None else 0
        self.has_encoder = encoder_config is not None
        if self.has_encoder:
            self.encoder = PointNet2Encoder(encoder_config)
        self.keypoint_encoder = PointUpsampleDecoder(decoder_config_list[0], in_dim=enc_dim,
                                                     apply_kl_regularization=apply_kl_regularization,
                                                     decode_only=not self.has_encoder)
        self.decoder = KeypointDecoder(decoder_config_list[1:], level_feature_dim(decoder_config_list[0]))

    @torch.no_grad()
    def encode(self, pointcloud, keypoint, ts=None, label=None, sample_posterior=True):
        """pointcloud (B,N,6), keypoint (B,16,3) -> latent features at the key points (B,16,48)"""
        if not self.has_encoder:
            raise NotImplementedError("this autoencoder was built decode-only (no encoder config)")
        out, l_xyz, _ = self.encoder(pointcloud, ts=ts, label=label)
        feat, _ = self.keypoint_encoder.propagate_feature(l_xyz[-1], out, keypoint, ts=ts, label=label,
                                                          sample_posterior=sample_posterior)
        return feat

    def forward(self, *a, **k):
        raise NotImplementedError("training (reconstruction / KL losses) is out of scope (SURVEY.md section 8(f) item 4)")

    @torch.no_grad()
    def decode(self, keypoint, feature_at_keypoint, ts=None, label=None, fps_start_idx=None):
        new_xyz = self.keypoint_encoder.upsample_points(feature_at_keypoint, keypoint, fps_start_idx)
        return self.decoder(keypoint[:, :, 0:3], feature_at_keypoint, new_xyz, ts=ts, label=label,
                            fps_start_idx=fps_start_idx)[-1]
