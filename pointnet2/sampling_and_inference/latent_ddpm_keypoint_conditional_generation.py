#!/usr/bin/env python
"""Feature-DDPM generation CLI on the HIP engine -- counterpart of the reference's
pointnet2/sampling_and_inference/latent_ddpm_keypoint_conditional_generation.py:39-177 up to the latent features: the
key points of --keypoint_file (the position CLI's npz) are the condition, the 48-dim features are generated and saved
as `keypoint_feature` next to `keypoint` in `<save_dir>/shapenet_psr_generated_latents_16_pts.npz`.
The autoencoder decode to 2048-point clouds (the reference's `points`) is SURVEY.md section 8 row a16 ("next")."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-c", "--config", type=str, required=True)
    ap.add_argument("--ckpt", type=str, default=None)
    ap.add_argument("--ema_idx", type=int, default=1)
    ap.add_argument("--keypoint_file", type=str, required=True)
    ap.add_argument("--save_dir", type=str, default="")
    ap.add_argument("--batch_size", type=int, default=32)
    ap.add_argument("--save_keypoint_feature", action="store_true")
    ap.add_argument("--random_init", action="store_true")
    ap.add_argument("--prec", default="fp32", choices=["fp32", "fp16"])
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    from slide_amd.checkpoint import load_denoiser_state
    from slide_amd.diffusion import FeatureSampler
    from slide_amd.generation import generate_latents, save_generated
    from slide_amd.json_reader import read_json_file

    cfg = read_json_file(a.config)
    hp = cfg["pointnet_config"]
    if a.ckpt is None and not a.random_init:
        raise SystemExit("--ckpt is required (or pass --random_init for synthetic weights)")
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", 1), ("RANK", 0), ("LOCAL_RANK", 0)))
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    kd = np.load(a.keypoint_file)
    keypoints, labels = kd["points"].astype(np.float32), kd["label"].astype(np.int64)
    n = keypoints.shape[0]
    B = a.batch_size
    sd = load_denoiser_state(hp, None if a.random_init else a.ckpt, a.ema_idx)
    smp = FeatureSampler(hp, sd, B, dev, cfg["standard_diffusion_config"], prec=a.prec, seed=a.seed + rank)
    rs = np.random.RandomState(a.seed + 7919 * rank)
    C = 3 + hp["in_fea_dim"]

    def run_batch(lab, lo, hi):
        m = hi - lo
        lab = np.concatenate([lab, np.zeros(B - m, np.int64)])
        kp = np.concatenate([keypoints[lo:hi], np.zeros((B - m, 16, 3), np.float32)])
        return smp.sample(lab, kp, rs.standard_normal((B, 16, C)).astype(np.float32))[:m]

    lat, timing = generate_latents(n, B, labels, run_batch, rank, world, gather_device=dev)
    if rank == 0:
        lat = lat.cpu().numpy()
        save_dir = a.save_dir or os.path.join(os.path.dirname(os.path.abspath(a.keypoint_file)), "latent_ddpm_generation")
        f = save_generated(save_dir, lat[:, :, :3], labels, np.resize(timing, n), 16, keypoint=lat[:, :, :3],
                           keypoint_feature=lat[:, :, 3:], ckpt_info="_latents")
        print("Generated latents have been saved to", f)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
