#!/usr/bin/env python
"""Key points + latent features -> point clouds, on the HIP module path -- counterpart of the reference's
pointnet2/sampling_and_inference/autoencoder_decode_keypoint.py:23-147 (same flags, same npz schemas):

  --dataset_path : npz with `keypoint` (n,16,3), `keypoint_feature` (n,16,48), `label`, `category`, `category_name`
                   (what the latent-DDPM CLI writes with --save_keypoint_feature; GeneralNpzDataset keys,
                   pointnet2/shapenet_psr_dataloader/npz_dataset.py:58-125)
  --save_dir     : receives `reconstructed_pcd.npz` (points, normals, label, category, category_name, keypoint) and the
                   `reconstructed_pcd_visualization/*.xyz` text dumps of visualize_pcd
                   (pointnet2/visualization_tools/visualize_pcd.py:16-60)
Extensions: --random_init (synthetic weights instead of --ckpt); --encode_from <npz with `points`, `normals`, `keypoint`,
`label`> first ENCODES the clouds to latent features with `PointAutoencoder.encode` (posterior mode) and decodes those --
the reconstruction round trip of the autoencoder."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def visualize_pcd(pcd_file, include_idx_to_save_name=True):
    """text dumps, one `.xyz` per shape (+ `_keypoint.xyz`), named like the reference's"""
    path, name = os.path.split(pcd_file)
    save_dir = os.path.join(path, os.path.splitext(name)[0] + "_visualization")
    os.makedirs(save_dir, exist_ok=True)
    data = np.load(pcd_file, allow_pickle=True)
    points = data["points"]
    if "normals" in data.files:
        points = np.concatenate([points, data["normals"]], axis=2)
    label = data["label"] if "label" in data.files else -np.ones(points.shape[0])
    names = data["category_name"] if "category_name" in data.files else None
    for i in range(points.shape[0]):
        nm = str(names[i]) if names is not None else "shape"
        stem = ("pcd_%s_label_%s_%s" % (str(i).zfill(3), str(label[i]).zfill(2), nm)) if include_idx_to_save_name \
            else ("label_%s_%s" % (str(label[i]).zfill(2), nm))
        np.savetxt(os.path.join(save_dir, stem + ".xyz"), points[i], delimiter=" ")
        if "keypoint" in data.files:
            np.savetxt(os.path.join(save_dir, stem + "_keypoint.xyz"), data["keypoint"][i], delimiter=" ")
    return save_dir


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-c", "--config", type=str, required=True, help="autoencoder JSON configuration")
    ap.add_argument("--ckpt", type=str, default=None, help="autoencoder checkpoint (model_state_dict)")
    ap.add_argument("--dataset_path", type=str, default=None, help="npz with keypoint / keypoint_feature / label ...")
    ap.add_argument("--save_dir", type=str, required=True)
    ap.add_argument("--batch_size", type=int, default=32)
    ap.add_argument("--not_include_idx_to_save_name", action="store_true")
    ap.add_argument("--random_init", action="store_true")
    ap.add_argument("--encode_from", type=str, default=None, help="npz with points, normals, keypoint, label: encode first")
    ap.add_argument("--seed", type=int, default=0, help="seeds the start index of the decode's plain-FPS calls (per shape: a function "
                                                          "of (seed, shape index), independent of ranks and batches)")
    ap.add_argument("--prec", default="mixed", choices=["mixed", "fp32", "fp16", "split"],
                    help="module-path arithmetic, as in the generation CLIs: mixed / fp16 (default) = fp16 MFMA operands with fp32 "
                         "accumulation, GroupNorm statistics and soft-max (what bench.py's decode leg times; pinned to the reference's "
                         "decode by tests/test_hip_modules.py); fp32 / split = exact fp32 MFMA")
    a = ap.parse_args()
    from slide_amd.generation import module_prec_of
    os.environ["SLIDE_MODULE_PREC"] = module_prec_of(a.prec)  # (read when the layers are built and when they run)

    import torch
    from models.autoencoder import PointAutoencoder
    from slide_amd.configs import CATEGORY_IDS, CATEGORY_NAMES
    from slide_amd.json_reader import autoencoder_read_config, read_json_file
    from slide_amd.synth import synth_state_dict

    if a.ckpt is None and not a.random_init:
        raise SystemExit("--ckpt is required (or pass --random_init for synthetic weights)")
    if (a.dataset_path is None) == (a.encode_from is None):
        raise SystemExit("give exactly one of --dataset_path / --encode_from")
    cfg = read_json_file(a.config)
    enc, decs = autoencoder_read_config(os.path.dirname(os.path.abspath(a.config)), cfg)
    # one process per GPU under torch.distributed.run: every rank decodes its contiguous shard (the
    # reference's per-rank decode: pointnet2/mesh_evaluation.py:113-118), one all-gather of the clouds, rank 0 writes
    from slide_amd.generation import all_gather_rows, decode_shard, init_distributed, shard_range
    rank, world, dev, gdev = init_distributed()
    ae = PointAutoencoder(enc if a.encode_from else None, decs,
                          apply_kl_regularization=cfg["pointnet_config"].get("apply_kl_regularization", False),
                          kl_weight=cfg["pointnet_config"].get("kl_weight", 0))
    if a.random_init:
        ae.load_state_dict({k: torch.from_numpy(v) for k, v in
                            synth_state_dict([(k, tuple(t.shape)) for k, t in ae.state_dict().items()]).items()})
    else:
        ck = torch.load(a.ckpt, map_location="cpu")["model_state_dict"]
        ae.load_state_dict({k: v for k, v in ck.items() if k in ae.state_dict()}, strict=True)
    ae = ae.to(dev).eval()

    src = np.load(a.encode_from or a.dataset_path, allow_pickle=True)
    keypoint = src["keypoint"].astype(np.float32)
    n = keypoint.shape[0]
    label = src["label"].astype(np.int64) if "label" in src.files else np.zeros(n, np.int64)
    B = a.batch_size
    s0, e0 = shard_range(n, rank, world)
    feats = None
    if a.encode_from:
        def encode(lo, hi, kp, lab):
            pc = np.concatenate([src["points"][s0 + lo:s0 + hi], src["normals"][s0 + lo:s0 + hi]], axis=2).astype(np.float32)
            return ae.encode(torch.from_numpy(pc).to(dev), kp, ts=None, label=lab, sample_posterior=False)
        c_local, f_local = decode_shard(ae, keypoint[s0:e0], None, label[s0:e0], B, dev, seed=a.seed, global_offset=s0, encode=encode)
        if f_local is None:
            f_local = torch.empty(0, 16, 1, device=dev)
        if world > 1:  # (an empty shard learns the feature width from the others)
            import torch.distributed as dist
            widths = [None] * world
            dist.all_gather_object(widths, int(f_local.shape[2]) if e0 > s0 else None)
            wf = next(w for w in widths if w is not None)
            if e0 == s0:
                f_local = torch.empty(0, 16, wf, device=dev)
        feats = all_gather_rows(f_local, n, world, device=gdev).cpu().numpy()
    else:
        c_local = decode_shard(ae, keypoint[s0:e0], src["keypoint_feature"][s0:e0].astype(np.float32), label[s0:e0], B, dev,
                               seed=a.seed, global_offset=s0)
    clouds = all_gather_rows(c_local, n, world, device=gdev).cpu().numpy()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    category = list(src["category"]) if "category" in src.files else [CATEGORY_IDS[int(l)] for l in label]
    category_name = list(src["category_name"]) if "category_name" in src.files else [CATEGORY_NAMES[int(l)] for l in label]
    result = {"points": clouds[:, :, 0:3], "label": label, "category": np.array(category), "category_name":
              np.array(category_name), "keypoint": keypoint}
    if clouds.shape[2] == 6:
        result["normals"] = clouds[:, :, 3:6]
    if a.encode_from:
        result["keypoint_feature"] = feats
    os.makedirs(a.save_dir, exist_ok=True)
    f = os.path.join(a.save_dir, "reconstructed_pcd.npz")
    np.savez(f, **result)
    print("reconstructed_pcd has been saved to", f)
    visualize_pcd(f, include_idx_to_save_name=not a.not_include_idx_to_save_name)


if __name__ == "__main__":
    main()
