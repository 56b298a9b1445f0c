"""`PointAutoencoder` -- DECODE side (reference: pointnet2/models/autoencoder.py:11-45): 16 key points + 48-dim latent
features -> 256 -> 1024 -> 2048 points x 6 (xyz + normal).  Parameter names follow the reference
(`keypoint_encoder.fc_layer.*`, `decoder.decoders.{i}.*`) so `load_state_dict(ckpt['model_state_dict'], strict=False)`
fills the decode path from a released checkpoint.  The encoder (`PointNet2Encoder`) is
not built: `encode` / `forward` raise."""
import torch
import torch.nn as nn

from models.keypoint_decoder import KeypointDecoder, level_feature_dim
from models.point_upsample_decoder import PointUpsampleDecoder


class PointAutoencoder(nn.Module):
    def __init__(self, encoder_config, decoder_config_list, apply_kl_regularization=False, kl_weight=0, feature_weight=None):
        super().__init__()
        self.apply_kl_regularization, self.kl_weight, self.feature_weight = apply_kl_regularization, kl_weight, feature_weight
        enc_dim = encoder_config["architecture"]["feature_dim"][-1] if encoder_config is not None else 0
        self.keypoint_encoder = PointUpsampleDecoder(decoder_config_list[0], in_dim=enc_dim, decode_only=True)
        self.decoder = KeypointDecoder(decoder_config_list[1:], level_feature_dim(decoder_config_list[0]))

    def encode(self, *a, **k):
        raise NotImplementedError("the encode path is not part of the sampling hot path (SURVEY.md section 8(f))")

    forward = encode

    @torch.no_grad()
    def decode(self, keypoint, feature_at_keypoint, ts=None, label=None, fps_start_idx=None):
        new_xyz = self.keypoint_encoder.upsample_points(feature_at_keypoint, keypoint, fps_start_idx)
        return self.decoder(keypoint[:, :, 0:3], feature_at_keypoint, new_xyz, ts=ts, label=label,
                            fps_start_idx=fps_start_idx)[-1]
