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


def synth_keypoints(batch, npoint=16, seed=0, cloud_points=2048):
    """'GT-like' synthetic key points built the way the reference builds ground-truth key points (SURVEY.md section 8(d)
    config 3): a `cloud_points`-point cloud in roughly [-1,1]^3 (here: seeded samples on an ellipsoid surface, the
    stand-in for a ShapeNet surface x2), the CENTROID PREPENDED, then farthest-point sampling to `npoint` starting at the
    centroid -- sample_keypoints(add_centroid=True) -> pytorch3d sample_farthest_points(random_start_point=False)
    (pointnet2/data_utils/points_sampling.py:156-187; call site train_latent_ddpm.py:187-193).  Plain numpy FPS
    (difference-form squared distances, first maximum wins); key point 0 is therefore the centroid."""
    rs = np.random.RandomState(1234 + seed)
    v = rs.standard_normal((batch, cloud_points, 3))
    v /= np.linalg.norm(v, axis=2, keepdims=True)
    v *= np.array([0.9, 0.35, 0.6])[None, None]
    v += 0.02 * rs.standard_normal(v.shape)  # not a perfect quadric: breaks exact ties
    v = v.astype(np.float32)
    x = np.concatenate([v.mean(axis=1, keepdims=True, dtype=np.float32), v], axis=1)  # (B, P+1, 3)
    sel = np.zeros((batch, npoint), np.int64)
    d = np.full((batch, x.shape[1]), np.inf, np.float32)
    ar = np.arange(batch)
    for j in range(1, npoint):
        last = x[ar, sel[:, j - 1]][:, None, :]
        d = np.minimum(d, ((x - last) ** 2).sum(axis=2, dtype=np.float32))
        sel[:, j] = d.argmax(axis=1)
    return np.ascontiguousarray(x[ar[:, None], sel]).astype(np.float32)
