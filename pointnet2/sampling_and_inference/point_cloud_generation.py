#!/usr/bin/env python
"""Position-DDPM generation CLI on the HIP engine -- same flags, config format and output file as the reference's
pointnet2/sampling_and_inference/point_cloud_generation.py:40-129 (-> `<save_dir>/shapenet_psr_generated_data_16_pts.npz`
with keys points,label,category,category_name,timing).  Extensions: --random_init (synthetic weights when no checkpoint
is available), --prec, --seed; multi-GPU via `python -m torch.distributed.run --nproc-per-node N ...`.
The reference draws labels by instantiating the ShapeNet dataset (mesh_evaluation.py:54-59); the labels of the
configured categories are used directly here, exactly what its DummyShapes3dDataset provides."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-c", "--config", type=str, required=True, help="JSON file for configuration")
    ap.add_argument("--ckpt", type=str, default=None, help="the checkpoint to use")
    ap.add_argument("--ema_idx", type=int, default=1)
    ap.add_argument("--num_samples", type=int, default=32)
    ap.add_argument("--batch_size", type=int, default=32)
    ap.add_argument("--save_dir", type=str, default="ddpm_generated_point_clouds")
    ap.add_argument("--data_clamp_range", type=float, default=1)
    ap.add_argument("--model_var_type", type=str, default="fixedsmall")
    ap.add_argument("--random_init", action="store_true")
    ap.add_argument("--prec", default="mixed", choices=["mixed", "fp32", "fp16", "split"],
                    help="arithmetic: mixed (default) = position DDPM in two-term fp16 operand splits (fp32-grade), feature DDPM in fp16 "
                         "operands / fp32 accumulation -- every forward within 1e-3 of fp32; fp32 = the exact parity mode (fp32 MFMA); "
                         "split = both DDPMs fp32-grade; fp16 = both in fp16 operands (position forwards up to 3e-3 off)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--chains", type=int, default=3,
                    help="batches kept in flight: the position chain is launch-latency bound, so independent batches run as "
                         "concurrent chains replayed round-robin (the arrangement bench.py times)")
    ap.add_argument("--serial_chains", action="store_true", help="run the same chains one after the other (bit-identical; tests)")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    from slide_amd.checkpoint import load_denoiser_state
    from slide_amd.configs import CATEGORY_IDS
    from slide_amd.diffusion import PositionSampler
    from slide_amd.generation import save_generated
    from slide_amd.json_reader import read_json_file

    cfg = read_json_file(a.config)
    hp = cfg["pointnet_config"]
    if "diffusion_config" not in cfg:
        raise SystemExit("this CLI drives the `diffusion_config` (util.sampling) path of the shipped position configs")
    if a.ckpt is None and not a.random_init:
        raise SystemExit("--ckpt is required (or pass --random_init for synthetic weights)")
    from slide_amd.generation import init_distributed, start_noise
    rank, world, dev, gdev = init_distributed()  # one process per GPU over RCCL (SLIDE_SHARE_GPU=1: gloo on one GPU, tests)
    sd = load_denoiser_state(hp, None if a.random_init else a.ckpt, a.ema_idx)
    B = a.batch_size
    from slide_amd.diffusion import EagerChainsSampler
    from slide_amd.generation import batches, shard_range
    import time
    cats = cfg["shapenet_psr_dataset_config"]["categories"]
    labels = np.array([CATEGORY_IDS.index(cats[i % len(cats)]) for i in range(a.num_samples)], np.int64)
    s0, e0 = shard_range(a.num_samples, rank, world)
    my = list(batches(s0, e0, B))
    C = max(1, min(a.chains, len(my)))
    # chain c serves batches c, c + C, ...  Start noise and in-kernel noise of a shape are functions of (seed, its GLOBAL index)
    # only: the output does not depend on the number of ranks, the batch size or the chains in flight
    from slide_amd.generation import resolve_prec
    smps = [PositionSampler(hp, sd, B, dev, cfg["diffusion_config"], prec=resolve_prec(a.prec)[0], seed=a.seed, use_graph=False)
            for c in range(C)]
    outs, timing = {}, []
    torch.cuda.synchronize(dev)
    t_all = time.time()
    for g0 in range(0, len(my), C):
        grp = my[g0:g0 + C]
        t0 = time.time()
        for c, (lo, hi) in enumerate(grp):
            lab = np.concatenate([labels[lo:hi], np.zeros(B - (hi - lo), np.int64)])  # the plan is built for a fixed batch
            smps[c].begin(lab, start_noise(a.seed, 1, lo, lo + B, (16, 3), dev), nonce=1, sample_offset=lo)
        if a.serial_chains:
            for c in range(len(grp)):
                smps[c].advance(smps[c].T)
        else:
            EagerChainsSampler(smps[:len(grp)]).advance(smps[0].T)
        for c, (lo, hi) in enumerate(grp):
            outs[lo] = smps[c].state()[:hi - lo]
        dt = time.time() - t0
        timing += [dt / sum(hi - lo for lo, hi in grp)] * sum(hi - lo for lo, hi in grp)
    torch.cuda.synchronize(dev)
    dt_all = time.time() - t_all
    local = torch.cat([outs[lo] for lo, _ in my]) if my else torch.empty(0, 16, 3, device=dev)
    from slide_amd.generation import all_gather_rows
    pts = all_gather_rows(local, a.num_samples, world, device=gdev)
    timing = np.asarray(timing)
    if rank == 0 and my:
        print("position DDPM: %d shapes on this rank in %.2f s = %.1f shapes/s (%d chain(s) of %d, %s)" % (
            e0 - s0, dt_all, (e0 - s0) / dt_all, C, B, a.prec))
    if rank == 0:
        f = save_generated(a.save_dir, pts.cpu().numpy(), labels, np.resize(timing, a.num_samples), 16)
        print("Generated samples have been saved to", f)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
