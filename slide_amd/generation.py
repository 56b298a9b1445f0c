"""Generation harness: this repo's counterpart of the reference's `evaluate_per_rank` / `gather_generated_results`
(pointnet2/mesh_evaluation.py:15-186) for the latent-DDPM path.

Multi-GPU (one process per GPU, `torch.distributed`): samples are independent (GroupNorm only, no cross-sample op), so
the requested samples are split into contiguous per-rank slices exactly like the reference's per-rank datasets
(ceil division, last rank short: pointnet2/shapenet_psr_dataloader/npz_dataset.py:90-96) with NO data-path
collective; a single all-gather of the generated latents replaces the reference's per-rank npz files + rank-0 file
concatenation (mesh_evaluation.py:156-186).  Over xGMI this is RCCL (backend "nccl"); the CPU tests run it on gloo.
"""
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from .configs import CATEGORY_IDS, CATEGORY_NAMES


def shard_range(num_samples, rank, world_size):
    """contiguous slice [start, end) of rank `rank` (ceil division; trailing ranks may be short or empty)"""
    if world_size <= 1:
        return 0, num_samples
    per = int(np.ceil(num_samples / world_size))
    return min(rank * per, num_samples), min((rank + 1) * per, num_samples)


def batches(start, end, batch_size):
    i = start
    while i < end:
        yield i, min(i + batch_size, end)
        i += batch_size


def all_gather_rows(local, num_samples, world_size, device=None):
    """one collective: every rank contributes its (n_local, ...) rows; returns the (num_samples, ...) concatenation in
    rank order on every rank.  Ranks hold different row counts, so rows are padded to the per-rank capacity."""
    if world_size <= 1:
        return local
    per = int(np.ceil(num_samples / world_size))
    t = torch.as_tensor(local)
    dev = device if device is not None else t.device
    pad = torch.zeros((per,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
    pad[: t.shape[0]] = t.to(dev)
    out = [torch.empty_like(pad) for _ in range(world_size)]
    dist.all_gather(out, pad)
    rows = []
    for r in range(world_size):
        s, e = shard_range(num_samples, r, world_size)
        rows.append(out[r][: e - s])
    return torch.cat(rows, dim=0)


def generate_latents(num_samples, batch_size, labels, run_batch, rank=0, world_size=1, gather_device=None,
                     row_shape=None, row_dtype=torch.float32):
    """Drives `run_batch(labels_np [b], lo, hi) -> (b, ...) tensor/ndarray` over this rank's slice and gathers.
    labels: int array [num_samples].  Returns (latents [num_samples, ...] on every rank, per-sample seconds [n_local]).
    row_shape / row_dtype: shape of one output row, e.g. (16, 51); lets a rank whose slice is EMPTY (more ranks than
    samples) join the all-gather without running a chain just to learn the shape."""
    start, end = shard_range(num_samples, rank, world_size)
    outs, timing = [], []
    for lo, hi in batches(start, end, batch_size):
        t0 = time.time()
        o = torch.as_tensor(run_batch(np.asarray(labels[lo:hi]), lo, hi))
        if o.is_cuda:
            torch.cuda.synchronize(o.device)  # the reference's timer stops without a device sync (mesh_evaluation.py:126)
        timing.extend([(time.time() - t0) / (hi - lo)] * (hi - lo))
        outs.append(o)
    local = torch.cat(outs, dim=0) if outs else None
    if row_shape is None and world_size > 1:
        # one tiny object collective tells a rank with an EMPTY slice (more ranks than samples) the row shape; it does
        # not run a throw-away 1000-step chain to learn it
        shapes = [None] * world_size
        dist.all_gather_object(shapes, None if local is None else (tuple(local.shape[1:]), local.dtype))
        row_shape, row_dtype = next(s_ for s_ in shapes if s_ is not None)
    if local is None:  # a rank with an empty slice still takes part in the all-gather
        local = torch.empty((0,) + tuple(row_shape or ()), dtype=row_dtype, device=gather_device)
    full = all_gather_rows(local, num_samples, world_size, device=gather_device)
    return full, np.asarray(timing)


def save_generated(save_dir, points, labels, timing, num_points, keypoint=None, keypoint_feature=None, ckpt_info=""):
    """npz schema of mesh_evaluation.py:135-150: points,label,category,category_name,timing[,keypoint,keypoint_feature]"""
    os.makedirs(save_dir, exist_ok=True)
    f = os.path.join(save_dir, "shapenet_psr_generated_data_%d_pts%s.npz" % (num_points, ckpt_info))
    labels = np.asarray(labels)
    d = {"points": np.asarray(points), "label": labels, "category": [CATEGORY_IDS[int(i)] for i in labels],
         "category_name": [CATEGORY_NAMES[int(i)] for i in labels], "timing": np.asarray(timing)}
    if keypoint is not None:
        d["keypoint"] = np.asarray(keypoint)
    if keypoint_feature is not None:
        d["keypoint_feature"] = np.asarray(keypoint_feature)
    np.savez(f, **d)
    return f
