"""The reference's training LOOP around the graphed step: key-point sampling, EMA weight sets, checkpoints in the reference's pickle
layout (so the generation CLIs -- slide_amd/checkpoint.py -- and the reference itself can load them), resume from the newest one.
Counterpart of pointnet2/train.py:33-300 (position DDPM, task 'keypoint_generation') and pointnet2/train_latent_ddpm.py:36-290
(feature DDPM on autoencoder latents) without their dataset (ShapeNet loading is out of scope: batches come from an iterable of
numpy dicts) and without the evaluation passes at the checkpoints."""
import os
import re
import time

import numpy as np
import torch
import torch.distributed as dist

from .. import _ext
from .dp import broadcast_parameters
from .graph import GraphedTrainingStep


class EmaSet:
    """one shadow copy of the trainable parameters per rate (pointnet2/data_utils/ema.py:4-30: shadow = (1 - mu) p + mu shadow after
    every optimizer step); update() is two multi-tensor launches per rate and capturable"""

    def __init__(self, net, rates):
        self.rates = [float(r) for r in (rates or [])]
        self.names = [k for k, p in net.named_parameters() if p.requires_grad]
        self.params = [p for _, p in net.named_parameters() if p.requires_grad]
        self.shadows = [[p.detach().clone() for p in self.params] for _ in self.rates]

    def update(self):
        with torch.no_grad():
            for mu, sh in zip(self.rates, self.shadows):
                torch._foreach_lerp_(sh, [p.detach() for p in self.params], 1.0 - mu)

    def state_list(self):
        return [{k: t.detach().clone() for k, t in zip(self.names, sh)} for sh in self.shadows]

    def load_state_list(self, lst):
        for sh, sd in zip(self.shadows, lst):
            for k, t in zip(self.names, sh):
                t.copy_(sd[k].to(t.device))


def sample_keypoints(x, K, add_centroid=True, random_subsample=False):
    """pointnet2/data_utils/points_sampling.py:156-187: K key points of every cloud x (B, P, D) by farthest point sampling -- from the
    centroid (prepended as point 0, which then is the first key point) or from a random start point -- or a random subset"""
    if add_centroid:
        x = torch.cat([x.mean(dim=1, keepdim=True), x], dim=1)
    if random_subsample:
        assert not add_centroid
        idx = torch.randperm(x.shape[1], device=x.device)[:K]
        return x[:, idx, :], idx.unsqueeze(0)
    return _ext.sample_farthest_points(x, K=K, random_start_point=not add_centroid)


def find_max_ckpt(directory, prefix="pointnet_ckpt"):
    """iteration of the newest `<prefix>_<iter>.pkl` in `directory`, -1 if none (util.find_max_epoch, pointnet2/util.py)"""
    best = -1
    if os.path.isdir(directory):
        for f in os.listdir(directory):
            m = re.fullmatch(re.escape(prefix) + r"_(\d+)\.pkl", f)
            if m:
                best = max(best, int(m.group(1)))
    return best


def parse_ema_rate(v):
    """train_config['ema_rate'] is a string in the shipped configs ("[0.999, 0.9999]")"""
    if v is None:
        return None
    if isinstance(v, str):
        v = [float(t) for t in v.strip("[] ").split(",") if t.strip()]
    return [float(t) for t in v]


def train_ddpm(net, static, loss_fn, batches, n_iters, output_directory, learning_rate=2e-4, ema_rate=None, iters_per_ckpt=1000,
               iters_per_logging=50, ckpt_iter="max", prepare=None, log=print):
    """Adam on `loss_fn()` (which reads the STATIC tensors of `static`), one graphed step per batch.

    batches: iterable of dicts of arrays; prepare(batch) -> dict with the keys of `static` (default: the batch itself), copied into
    the static tensors before each step.  Checkpoints `pointnet_ckpt_<iter>.pkl` every iters_per_ckpt iterations: iter,
    model_state_dict, optimizer_state_dict, training_time_seconds, ema_state_list (train.py:243-255).  Returns the last iteration."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    opt = torch.optim.Adam(net.parameters(), lr=learning_rate, capturable=True)
    ema = EmaSet(net, ema_rate)
    time0 = time.time()
    if ckpt_iter == "max":
        ckpt_iter = find_max_ckpt(output_directory)
    n_iter = 0
    if ckpt_iter is not None and int(ckpt_iter) >= 0:
        ck = torch.load(os.path.join(output_directory, "pointnet_ckpt_%d.pkl" % int(ckpt_iter)), map_location="cpu")
        net.load_state_dict(ck["model_state_dict"])
        opt.load_state_dict(ck["optimizer_state_dict"])
        if ema.rates and "ema_state_list" in ck:
            ema.load_state_list(ck["ema_state_list"])
        time0 -= ck.get("training_time_seconds", 0)
        n_iter = int(ck["iter"]) + 1
        log("checkpoint of iteration %d loaded (trained for %d s)" % (int(ck["iter"]), ck.get("training_time_seconds", 0)))
    broadcast_parameters(net)
    step = None
    t_log = time.time()
    it = iter(batches)
    while n_iter < n_iters:
        try:
            batch = next(it)
        except StopIteration:
            it = iter(batches)  # next epoch
            batch = next(it)
        data = prepare(batch) if prepare is not None else batch
        for k, t in static.items():
            t.copy_(torch.as_tensor(data[k]).to(t.device, t.dtype), non_blocking=True)
        if step is None:
            # captured on the first batch (the static tensors then hold real data for the capture protocol's eager warm-up steps);
            # those steps must not count: weights, optimizer state and EMA sets are put back IN PLACE afterwards
            saved_p = [p.detach().clone() for p in net.parameters()]
            saved_e = [[t.clone() for t in sh] for sh in ema.shadows]
            saved_o = {id(p): {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()} for p, st in opt.state.items()}
            step = GraphedTrainingStep(net, opt, loss_fn, post_step=ema.update if ema.rates else None)
            with torch.no_grad():
                for p, q in zip(net.parameters(), saved_p):
                    p.copy_(q)
                for sh, sv in zip(ema.shadows, saved_e):
                    for t, q in zip(sh, sv):
                        t.copy_(q)
                for p, st in opt.state.items():
                    for k, v in st.items():
                        if torch.is_tensor(v):
                            old = saved_o.get(id(p), {}).get(k)
                            v.copy_(old.to(v.device)) if old is not None else v.zero_()
        loss = step()
        if n_iter % iters_per_logging == 0:
            log("iteration: %d \tloss: %.6f \ttime: %.2fs" % (n_iter, float(loss), time.time() - t_log))
            t_log = time.time()
        if n_iter > 0 and (n_iter + 1) % iters_per_ckpt == 0 and rank == 0:
            os.makedirs(output_directory, exist_ok=True)
            states = {"iter": n_iter, "model_state_dict": {k: v.detach().cpu() for k, v in net.state_dict().items()},
                      "optimizer_state_dict": opt.state_dict(), "training_time_seconds": int(time.time() - time0)}
            if ema.rates:
                states["ema_state_list"] = [{k: v.cpu() for k, v in sd.items()} for sd in ema.state_list()]
            torch.save(states, os.path.join(output_directory, "pointnet_ckpt_%d.pkl" % n_iter))
            log("model at iteration %d is saved" % n_iter)
        n_iter += 1
    return n_iter - 1


def npz_batches(path, batch_size, rank=0, world=1, seed=0):
    """epochs over an npz of clouds (`points` (n, P, 3), optional `normals`, `label`): a fresh permutation per epoch, this rank's
    contiguous share of it, full batches only (the stand-in for the reference's ShapeNet loader, pointnet2/dataset.py)"""
    with np.load(path) as z:  # (an NpzFile re-reads and decompresses an array on EVERY access: load each once -- ADVICE r4)
        d = {k: np.asarray(z[k]) for k in ("points", "normals", "label") if k in z.files}
    n = d["points"].shape[0]
    rs = np.random.RandomState(seed)

    class _Epochs:
        def __len__(self):  # full batches of this rank's share per epoch
            return (n // world) // batch_size

        def __iter__(self):
            perm = rs.permutation(n)
            per = n // world
            mine = perm[rank * per:(rank + 1) * per]
            for i in range(0, len(mine) - batch_size + 1, batch_size):
                j = np.sort(mine[i:i + batch_size])
                yield {k: v[j] for k, v in d.items()}

    return _Epochs()
