"""Checkpoint loading with the reference's pickle layout (pointnet2/train.py:243-255): a dict with
`model_state_dict` and `ema_state_list` (one {param_name: tensor} per EMA rate, only trainable params,
pointnet2/data_utils/ema.py:13-18).  Inference = model_state_dict overlaid by ema_state_list[ema_idx]
(pointnet2/sampling_and_inference/point_cloud_generation.py:24-27)."""
import numpy as np
import torch

from .model_spec import denoiser_param_spec
from .synth import synth_state_dict


def load_denoiser_state(hp, ckpt=None, ema_idx=1, seed=0):
    spec = dict(denoiser_param_spec(hp))
    if ckpt is None:
        return synth_state_dict(spec, seed)  # random-init weights of the right architecture (no network for real ones)
    ck = torch.load(ckpt, map_location="cpu")
    sd = dict(ck["model_state_dict"])
    if ema_idx >= 0:
        sd.update(ck["ema_state_list"][ema_idx])
    out = {}
    for name, shape in spec.items():
        if name not in sd:
            raise KeyError("checkpoint lacks parameter %s" % name)
        t = sd[name].detach().cpu().numpy().astype(np.float32)
        if tuple(t.shape) != tuple(shape):
            raise ValueError("parameter %s has shape %s, expected %s" % (name, t.shape, shape))
        out[name] = t
    return out
