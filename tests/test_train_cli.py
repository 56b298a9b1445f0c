"""GPU: the two training CLIs (pointnet2/train.py, pointnet2/train_latent_ddpm.py) end to end on synthetic clouds -- configs in the
reference's JSON format, checkpoints in the reference's pickle layout (pointnet2/train.py:243-255), resume from the newest one,
and the trained checkpoints through the generation side (slide_amd/checkpoint.py, the generation CLI)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import REPO, load_golden
from slide_amd import configs, model_spec
from test_hip_cli import _stringify

pytestmark = pytest.mark.gpu


def _clouds(path, n, P=512, seed=0):
    rs = np.random.RandomState(seed)
    pts = rs.uniform(-0.8, 0.8, (n, P, 3)).astype(np.float32)
    nrm = rs.standard_normal((n, P, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=2, keepdims=True)
    np.savez(path, points=pts, normals=nrm, label=np.zeros(n, np.int64))


def _train_cfg(task):
    return {"task": task, "dataset": "shapenet_psr_dataset", "root_directory": "unused", "output_directory": "checkpoint",
            "ckpt_iter": "max", "epochs_per_ckpt": 1, "iters_per_logging": 2, "n_epochs": 2, "learning_rate": 0.0002,
            "ema_rate": "[0.999, 0.9999]"}


def test_position_training_cli_checkpoints_and_resume(gpu_device, tmp_path):
    pc = configs.position_ddpm_config()
    pc["train_config"] = _train_cfg("keypoint_generation")
    pc["shapenet_psr_dataset_config"] = {"dataset": "shapenet_psr_dataset", "categories": ["02691156"], "npoints": 512, "batch_size": 8,
                                         "num_keypoints": 16, "keypoints_source": "farthest_points_sampling"}
    cfg = tmp_path / "pos.json"
    cfg.write_text(json.dumps(_stringify(pc)))
    _clouds(tmp_path / "clouds.npz", 32)
    env = dict(os.environ, PYTHONPATH=REPO)
    cmd = [sys.executable, os.path.join(REPO, "pointnet2", "train.py"), "-c", str(cfg), "--dataset_npz", str(tmp_path / "clouds.npz"),
           "--root_directory", str(tmp_path / "exp")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True)   # 2 epochs x 4 batches, a checkpoint per epoch
    assert r.returncode == 0, r.stderr[-3000:]
    ckdir = tmp_path / "exp" / pc["pointnet_config"]["model_name"] / "checkpoint"
    assert sorted(os.listdir(ckdir)) == ["pointnet_ckpt_3.pkl", "pointnet_ckpt_7.pkl"], os.listdir(ckdir)
    losses = [float(l.split("loss: ")[1].split()[0]) for l in r.stdout.splitlines() if l.startswith("iteration:")]
    assert len(losses) == 4 and np.isfinite(losses).all()
    ck = torch.load(ckdir / "pointnet_ckpt_7.pkl", map_location="cpu")
    assert set(ck) == {"iter", "model_state_dict", "optimizer_state_dict", "training_time_seconds", "ema_state_list"} and ck["iter"] == 7
    spec = model_spec.denoiser_param_spec(pc["pointnet_config"])
    assert list(ck["model_state_dict"].keys()) == [n for n, _ in spec]
    assert len(ck["ema_state_list"]) == 2 and set(ck["ema_state_list"][1]) == set(n for n, _ in spec)
    # the slow EMA set stays closer to the initial weights than the fast one, both differ from the live weights
    w, e0, e1 = (ck["model_state_dict"]["fc_t1.weight"], ck["ema_state_list"][0]["fc_t1.weight"], ck["ema_state_list"][1]["fc_t1.weight"])
    assert 0 < float((w - e1).abs().max()) and float((w - e0).abs().max()) < float((w - e1).abs().max())
    # the generation side loads it: model_state_dict overlaid by an EMA set
    from slide_amd.checkpoint import load_denoiser_state
    sd = load_denoiser_state(pc["pointnet_config"], str(ckdir / "pointnet_ckpt_7.pkl"), ema_idx=1)
    assert np.allclose(sd["fc_t1.weight"], e1.numpy())
    # resume: one more epoch from the newest checkpoint
    r2 = subprocess.run(cmd + ["--n_iters", "12"], env=env, capture_output=True, text=True)
    assert r2.returncode == 0, r2.stderr[-3000:]
    assert "checkpoint of iteration 7 loaded" in r2.stdout and (ckdir / "pointnet_ckpt_11.pkl").exists()
    ck2 = torch.load(ckdir / "pointnet_ckpt_11.pkl", map_location="cpu")
    st = next(iter(ck2["optimizer_state_dict"]["state"].values()))
    assert int(st["step"]) == 12   # Adam's step count went on from the checkpoint's 8 (the capture's warm-up steps do not count)
    gen = subprocess.run([sys.executable, os.path.join(REPO, "pointnet2", "sampling_and_inference", "point_cloud_generation.py"), "-c", str(cfg),
                          "--ckpt", str(ckdir / "pointnet_ckpt_11.pkl"), "--ema_idx", "0", "--num_samples", "4", "--batch_size", "4",
                          "--save_dir", str(tmp_path / "gen")], env=env, capture_output=True, text=True)
    assert gen.returncode == 0, gen.stderr[-2000:]
    d = np.load(tmp_path / "gen" / "shapenet_psr_generated_data_16_pts.npz")
    assert d["points"].shape == (4, 16, 3) and np.isfinite(d["points"]).all()


def test_latent_training_cli(gpu_device, tmp_path):
    g, ge = load_golden("golden_decode.npz"), load_golden("golden_encode.npz")
    ae_dir = tmp_path / "configs" / "ae"
    os.makedirs(ae_dir / "lv")
    for i, dcfg in enumerate(json.loads(str(g["decoder_configs_json"]))):
        (ae_dir / "lv" / ("d%d.json" % i)).write_text(json.dumps({"pointnet_config": _stringify(dcfg)}))
    (ae_dir / "lv" / "enc.json").write_text(json.dumps({"pointnet_config": _stringify(json.loads(str(ge["encoder_config_json"])))}))
    (ae_dir / "ae.json").write_text(json.dumps({"pointnet_config": {"apply_kl_regularization": True, "kl_weight": 1e-5,
                                                                  "encoder_config_file": "lv/enc.json",
                                                                  "decoder_config_file": "['lv/d0.json', 'lv/d1.json', 'lv/d2.json']"}}))
    fc = configs.feature_ddpm_config()
    fc["autoencoder_config"] = {"config_file": str(ae_dir / "ae.json"), "ckpt": "unused"}
    fc["train_config"] = _train_cfg("latent_keypoint_conditional_generation")
    fc["shapenet_psr_dataset_config"] = {"dataset": "shapenet_psr_dataset", "categories": ["03001627"], "npoints": 2048, "batch_size": 4,
                                         "num_keypoints": 16, "keypoints_source": "farthest_points_sampling",
                                         "add_centroid_to_keypoints": False, "keypoint_noise_magnitude": 0.04}
    cfg = tmp_path / "feat.json"
    cfg.write_text(json.dumps(_stringify(fc)))
    _clouds(tmp_path / "clouds.npz", 8, P=2048, seed=1)
    env = dict(os.environ, PYTHONPATH=REPO)
    r = subprocess.run([sys.executable, os.path.join(REPO, "pointnet2", "train_latent_ddpm.py"), "-c", str(cfg), "--dataset_npz",
                        str(tmp_path / "clouds.npz"), "--root_directory", str(tmp_path / "exp"), "--random_init_ae"],
                       env=env, capture_output=True, text=True)   # 2 epochs x 2 batches
    assert r.returncode == 0, r.stderr[-3000:]
    ckdir = tmp_path / "exp" / fc["pointnet_config"]["model_name"] / "checkpoint"
    assert sorted(os.listdir(ckdir)) == ["pointnet_ckpt_1.pkl", "pointnet_ckpt_3.pkl"], os.listdir(ckdir)
    losses = [float(l.split("loss: ")[1].split()[0]) for l in r.stdout.splitlines() if l.startswith("iteration:")]
    assert len(losses) == 2 and np.isfinite(losses).all() and 0.5 < losses[0] < 2.0   # an untrained eps-prediction loss is ~1
    ck = torch.load(ckdir / "pointnet_ckpt_3.pkl", map_location="cpu")
    assert list(ck["model_state_dict"].keys()) == [n for n, _ in model_spec.denoiser_param_spec(fc["pointnet_config"])]
    from slide_amd.checkpoint import load_denoiser_state
    sd = load_denoiser_state(fc["pointnet_config"], str(ckdir / "pointnet_ckpt_3.pkl"), ema_idx=0)
    assert all(np.isfinite(v).all() for v in sd.values())
