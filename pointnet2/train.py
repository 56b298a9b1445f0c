#!/usr/bin/env python
"""Position-DDPM training CLI on the HIP training path -- counterpart of the reference's pointnet2/train.py:33-300 for the task
`keypoint_generation` (the DDPM over the 16 key points of a shape): same JSON config (pointnet_config, diffusion_config,
train_config, shapenet_psr_dataset_config), same checkpoint files `<root>/<pointnet model name>/checkpoint/pointnet_ckpt_<iter>.pkl`
(model_state_dict, optimizer_state_dict, ema_state_list, iter, training_time_seconds), which `point_cloud_generation.py` loads.
One process per GPU under torch.distributed.run (gradient all-reduce over RCCL).
Not the reference's: the ShapeNet loader (out of scope) -- clouds come from `--dataset_npz` (`points` (n, P, 3), `label` (n,)); the
evaluation passes at the checkpoints (generation + metrics); tensorboard."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-c", "--config", type=str, required=True)
    ap.add_argument("--dataset_npz", type=str, required=True, help="training clouds: points (n, P, 3), label (n,)")
    ap.add_argument("--n_iters", type=int, default=None, help="default: train_config n_epochs x batches per epoch")
    ap.add_argument("--iters_per_ckpt", type=int, default=None, help="default: train_config epochs_per_ckpt x batches per epoch")
    ap.add_argument("--root_directory", type=str, default=None, help="default: train_config root_directory")
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()

    import numpy as np
    import torch
    from slide_amd.generation import init_distributed
    from slide_amd.json_reader import read_json_file
    from slide_amd.train.denoiser import TrainableDenoiser
    from slide_amd.train.losses import position_training_loss
    from slide_amd.train.trainer import npz_batches, parse_ema_rate, sample_keypoints, train_ddpm

    cfg = read_json_file(a.config)
    tc, dc, hp = cfg["train_config"], cfg["shapenet_psr_dataset_config"], cfg["pointnet_config"]
    if tc["task"] != "keypoint_generation":
        raise SystemExit("this CLI trains the key-point DDPM (task keypoint_generation); got %s" % tc["task"])
    rank, world, dev, _ = init_distributed()
    torch.manual_seed(a.seed + rank)
    B, K = int(dc["batch_size"]), int(dc["num_keypoints"])
    batches = npz_batches(a.dataset_npz, B, rank, world, seed=a.seed)
    per_epoch = max(1, len(batches))
    out_dir = os.path.join(a.root_directory or tc["root_directory"], hp.get("model_name", "pointnet"), tc["output_directory"])
    net = TrainableDenoiser(hp).reset_parameters(a.seed).to(dev)
    static = {"keypoint": torch.zeros(B, K, 3, device=dev), "label": torch.zeros(B, dtype=torch.int64, device=dev)}
    add_centroid = dc.get("add_centroid_to_keypoints", True)

    def prepare(batch):  # train.py:186-199: key points of the cloud by farthest point sampling are the DDPM's data
        pts = torch.as_tensor(batch["points"], dtype=torch.float32, device=dev)
        kp, _ = sample_keypoints(pts, K, add_centroid=add_centroid, random_subsample=dc.get("random_sample_keypoints", False))
        lab = batch["label"] if "label" in batch else np.zeros(B, np.int64)
        return {"keypoint": kp, "label": torch.as_tensor(lab)}

    last = train_ddpm(net, static, lambda: position_training_loss(net, static["keypoint"], cfg["diffusion_config"], static["label"]),
                      batches, a.n_iters or int(tc["n_epochs"]) * per_epoch, out_dir, learning_rate=tc["learning_rate"],
                      ema_rate=parse_ema_rate(tc.get("ema_rate")), iters_per_ckpt=a.iters_per_ckpt or int(tc["epochs_per_ckpt"]) * per_epoch,
                      iters_per_logging=int(tc.get("iters_per_logging", 50)), ckpt_iter=tc.get("ckpt_iter", "max"),
                      prepare=prepare, log=(print if rank == 0 else (lambda *_: None)))
    if rank == 0:
        print("trained to iteration %d; checkpoints in %s" % (last, out_dir))


if __name__ == "__main__":
    main()
