"""Deterministic synthetic weights / inputs (there is no network for checkpoints or ShapeNet).

`synth_state_dict(spec)` fills a {state_dict_name: shape} spec with reproducible values keyed on
the parameter NAME (crc32) so that fixtures only have to store inputs and outputs, never weights
(SURVEY.md section 8(c)).  Names follow the reference checkpoints.
"""
import zlib

import numpy as np


def synth_tensor(name, shape, seed=0):
    rs = np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    shape = tuple(int(s) for s in shape)
    if name == "class_emb.weight":
        return rs.standard_normal(shape).astype(np.float32)
    if name.endswith(".weight") and len(shape) == 1:          # GroupNorm gamma
        return (1.0 + 0.2 * rs.standard_normal(shape)).astype(np.float32)
    if name.endswith(".bias"):
        return (0.1 * rs.standard_normal(shape)).astype(np.float32)
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
    return (rs.standard_normal(shape) / np.sqrt(max(fan_in, 1))).astype(np.float32)


def synth_state_dict(spec, seed=0):
    """spec: iterable of (name, shape) or dict name->shape."""
    items = spec.items() if isinstance(spec, dict) else spec
    return {n: synth_tensor(n, s, seed) for n, s in items}


def synth_keypoints(batch, npoint=16, seed=0):
    """'GT-like' synthetic key points: points on an ellipsoid
    surface in roughly [-1,1]^3, first point the centroid (the reference prepends the centroid
    before FPS, P2/data_utils/points_sampling.py:156-187)."""
    rs = np.random.RandomState(1234 + seed)
    v = rs.standard_normal((batch, npoint, 3))
    v /= np.linalg.norm(v, axis=2, keepdims=True)
    v *= np.array([0.9, 0.35, 0.6])[None, None]
    v[:, 0] = v[:, 1:].mean(axis=1)
    return v.astype(np.float32)
