#!/usr/bin/env python
"""Feature-DDPM (latent DDPM) training CLI on the HIP training path -- counterpart of the reference's
pointnet2/train_latent_ddpm.py:36-290: the clouds are encoded to 16 x F latent features by the FROZEN autoencoder
(`PointAutoencoder.encode` on the HIP module path; key points by farthest point sampling, optional key-point noise) and the
key-point-conditional DDPM is trained on [key points | features] with `LatentDiffusion.train_loss`.  Same JSON config
(pointnet_config, standard_diffusion_config, autoencoder_config, train_config, shapenet_psr_dataset_config) and checkpoint files
as the reference; `latent_ddpm_keypoint_conditional_generation.py` loads them.  One process per GPU under torch.distributed.run.
Not the reference's: the ShapeNet loader (clouds come from `--dataset_npz`: points, normals, label), the evaluation passes."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-c", "--config", type=str, required=True)
    ap.add_argument("--dataset_npz", type=str, required=True, help="training clouds: points (n, P, 3), normals (n, P, 3), label (n,)")
    ap.add_argument("--ae_ckpt", type=str, default=None, help="autoencoder checkpoint (default: config['autoencoder_config']['ckpt'])")
    ap.add_argument("--random_init_ae", action="store_true", help="synthetic autoencoder weights (tests; no released checkpoint here)")
    ap.add_argument("--n_iters", type=int, default=None)
    ap.add_argument("--iters_per_ckpt", type=int, default=None)
    ap.add_argument("--root_directory", type=str, default=None)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()

    import numpy as np
    import torch
    from models.autoencoder import PointAutoencoder
    from slide_amd.generation import init_distributed
    from slide_amd.json_reader import autoencoder_read_config, read_json_file
    from slide_amd.synth import synth_state_dict
    from slide_amd.train.denoiser import TrainableDenoiser
    from slide_amd.train.losses import latent_training_loss
    from slide_amd.train.trainer import npz_batches, parse_ema_rate, sample_keypoints, train_ddpm

    cfg = read_json_file(a.config)
    tc, dc, hp, sdc = cfg["train_config"], cfg["shapenet_psr_dataset_config"], cfg["pointnet_config"], cfg["standard_diffusion_config"]
    rank, world, dev, _ = init_distributed()
    torch.manual_seed(a.seed + rank)
    # the frozen autoencoder (train_latent_ddpm.py:84-108)
    ae_cfg_file = cfg["autoencoder_config"]["config_file"]
    if not os.path.isabs(ae_cfg_file):
        ae_cfg_file = os.path.join(os.path.dirname(os.path.abspath(a.config)), "..", "..", "..", ae_cfg_file)
    ae_cfg = read_json_file(ae_cfg_file)
    enc, decs = autoencoder_read_config(os.path.dirname(ae_cfg_file), ae_cfg)
    ae = PointAutoencoder(enc, decs, apply_kl_regularization=ae_cfg["pointnet_config"].get("apply_kl_regularization", False),
                          kl_weight=ae_cfg["pointnet_config"].get("kl_weight", 0))
    if a.random_init_ae:
        ae.load_state_dict({k: torch.from_numpy(v) for k, v in
                            synth_state_dict([(k, tuple(t.shape)) for k, t in ae.state_dict().items()]).items()})
    else:
        ck = torch.load(a.ae_ckpt or cfg["autoencoder_config"]["ckpt"], map_location="cpu")["model_state_dict"]
        ae.load_state_dict({k: v for k, v in ck.items() if k in ae.state_dict()}, strict=True)
    ae = ae.to(dev).eval()

    B, K = int(dc["batch_size"]), int(dc["num_keypoints"])
    batches = npz_batches(a.dataset_npz, B, rank, world, seed=a.seed)
    per_epoch = max(1, len(batches))
    out_dir = os.path.join(a.root_directory or tc["root_directory"], hp.get("model_name", "pointnet"), tc["output_directory"])
    net = TrainableDenoiser(hp).reset_parameters(a.seed).to(dev)
    F = int(hp["in_fea_dim"])
    static = {"x": torch.zeros(B, K, 3 + F, device=dev), "keypoint": torch.zeros(B, K, 3, device=dev),
              "label": torch.zeros(B, dtype=torch.int64, device=dev)}
    add_centroid = dc.get("add_centroid_to_keypoints", True)
    noise = float(dc.get("keypoint_noise_magnitude", 0))

    def prepare(batch):  # train_latent_ddpm.py:182-196 + LatentDiffusion.train_loss's encode (diffusion.py:319-327)
        pts = torch.as_tensor(batch["points"], dtype=torch.float32, device=dev)
        kp, _ = sample_keypoints(pts, K, add_centroid=add_centroid, random_subsample=dc.get("random_sample_keypoints", False))
        if noise > 0:
            kp = kp + noise * torch.randn_like(kp)
        X = pts
        if dc.get("include_normals", True):
            X = torch.cat([pts, torch.as_tensor(batch["normals"], dtype=torch.float32, device=dev)], dim=2)
        lab = torch.as_tensor(batch["label"] if "label" in batch else np.zeros(B, np.int64)).to(dev)
        with torch.no_grad():
            feat = ae.encode(X, kp, ts=None, label=lab, sample_posterior=True)
        feat = feat[0] if isinstance(feat, (tuple, list)) else feat
        return {"x": torch.cat([kp, feat.reshape(B, K, -1)], dim=2), "keypoint": kp, "label": lab}

    last = train_ddpm(net, static, lambda: latent_training_loss(net, static["x"], static["keypoint"], static["label"], sdc).mean(),
                      batches, a.n_iters or int(tc["n_epochs"]) * per_epoch, out_dir, learning_rate=tc["learning_rate"],
                      ema_rate=parse_ema_rate(tc.get("ema_rate")), iters_per_ckpt=a.iters_per_ckpt or int(tc["epochs_per_ckpt"]) * per_epoch,
                      iters_per_logging=int(tc.get("iters_per_logging", 50)), ckpt_iter=tc.get("ckpt_iter", "max"),
                      prepare=prepare, log=(print if rank == 0 else (lambda *_: None)))
    if rank == 0:
        print("trained to iteration %d; checkpoints in %s" % (last, out_dir))


if __name__ == "__main__":
    main()
