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
    ap.add_argument("--prec", default="fp32", choices=["fp32", "fp16"])
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    from slide_amd.checkpoint import load_denoiser_state
    from slide_amd.configs import CATEGORY_IDS
    from slide_amd.diffusion import PositionSampler
    from slide_amd.generation import generate_latents, save_generated
    from slide_amd.json_reader import read_json_file

    cfg = read_json_file(a.config)
    hp = cfg["pointnet_config"]
    if "diffusion_config" not in cfg:
        raise SystemExit("this CLI drives the `diffusion_config` (util.sampling) path of the shipped position configs")
    if a.ckpt is None and not a.random_init:
        raise SystemExit("--ckpt is required (or pass --random_init for synthetic weights)")
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", 1), ("RANK", 0), ("LOCAL_RANK", 0)))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # RCCL, bound to this rank's GPU
    sd = load_denoiser_state(hp, None if a.random_init else a.ckpt, a.ema_idx)
    B = a.batch_size
    smp = PositionSampler(hp, sd, B, dev, cfg["diffusion_config"], prec=a.prec, seed=a.seed + rank)
    cats = cfg["shapenet_psr_dataset_config"]["categories"]
    labels = np.array([CATEGORY_IDS.index(cats[i % len(cats)]) for i in range(a.num_samples)], np.int64)
    rs = np.random.RandomState(a.seed + 7919 * rank)

    def run_batch(lab, lo, hi):
        n = hi - lo
        lab = np.concatenate([lab, np.zeros(B - n, np.int64)])  # the plan is built for a fixed batch; pad the last one
        return smp.sample(lab, rs.standard_normal((B, 16, 3)).astype(np.float32))[:n]

    pts, timing = generate_latents(a.num_samples, B, labels, run_batch, rank, world, gather_device=dev, row_shape=(16, 3))
    if rank == 0:
        f = save_generated(a.save_dir, pts.cpu().numpy(), labels, np.resize(timing, a.num_samples), 16)
        print("Generated samples have been saved to", f)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
