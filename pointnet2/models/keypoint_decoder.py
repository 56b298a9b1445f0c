"""Stack of PointUpsampleDecoder levels (reference: pointnet2/models/keypoint_decoder.py:7-36): level i propagates the
features of level i-1 onto the points produced so far and splits them again."""
import torch.nn as nn

from models.point_upsample_decoder import PointUpsampleDecoder


def level_feature_dim(cfg):
    arch = cfg["architecture"]
    base = arch["decoder_feature_dim"][0] if "decoder_feature_dim" in arch else arch["feature_dim"][-1]
    return base + cfg["feature_mapper_setting"]["out_dim"]


class KeypointDecoder(nn.Module):
    def __init__(self, config_list, feature_dim):
        super().__init__()
        self.decoders = nn.ModuleList()
        for cfg in config_list:
            self.decoders.append(PointUpsampleDecoder(cfg, in_dim=feature_dim))
            feature_dim = level_feature_dim(cfg)

    def forward(self, xyz0, features0, xyz1, ts=None, label=None, fps_start_idx=None):
        l_xyzs, feats = [xyz0, xyz1], features0
        for i, dec in enumerate(self.decoders):
            feats, new_xyz = dec(l_xyzs[i][:, :, 0:3], feats, l_xyzs[i + 1], ts=ts, label=label, fps_start_idx=fps_start_idx)
            l_xyzs.append(new_xyz)
        return l_xyzs
