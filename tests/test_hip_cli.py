"""GPU: the two generation CLIs end to end (random-init weights, configs written in the reference's JSON format with
string-encoded lists), npz schema of the reference harness (mesh_evaluation.py:135-150)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO, load_golden
from slide_amd import configs

pytestmark = pytest.mark.gpu


def _stringify(d):
    out = {}
    for k, v in d.items():
        out[k] = _stringify(v) if isinstance(v, dict) else (str(v) if isinstance(v, list) else v)
    return out


def test_generation_clis(gpu_device, tmp_path):
    cdir = tmp_path / "configs" / "a" / "b"
    os.makedirs(cdir)
    pc = configs.position_ddpm_config()
    pc["shapenet_psr_dataset_config"] = {"dataset": "shapenet_psr_dataset", "categories": ["02691156"], "num_keypoints": 16}
    pc["train_config"] = {"task": "keypoint_generation", "dataset": "shapenet_psr_dataset"}
    pos_cfg = cdir / "pos.json"
    pos_cfg.write_text(json.dumps(_stringify(pc)))
    env = dict(os.environ, PYTHONPATH=REPO)
    cli = os.path.join(REPO, "pointnet2", "sampling_and_inference")
    out1 = tmp_path / "gen"
    r = subprocess.run([sys.executable, os.path.join(cli, "point_cloud_generation.py"), "-c", str(pos_cfg), "--random_init",
                        "--num_samples", "6", "--batch_size", "4", "--save_dir", str(out1)], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    kpf = out1 / "shapenet_psr_generated_data_16_pts.npz"
    d = np.load(kpf)
    assert set(d.files) == {"points", "label", "category", "category_name", "timing"}
    assert d["points"].shape == (6, 16, 3) and np.isfinite(d["points"]).all() and (d["label"] == 0).all()
    assert d["category"][0] == "02691156" and d["category_name"][0] == "airplane" and d["timing"].shape == (6,)
    # feature DDPM + decode; the autoencoder config points to one encoder + three decoder level files
    g = load_golden("golden_decode.npz")
    decs = json.loads(str(g["decoder_configs_json"]))
    ae_dir = tmp_path / "configs" / "ae"
    os.makedirs(ae_dir / "lv")
    for i, dcfg in enumerate(decs):
        (ae_dir / "lv" / ("d%d.json" % i)).write_text(json.dumps({"pointnet_config": _stringify(dcfg)}))
    (ae_dir / "lv" / "enc.json").write_text(json.dumps({"pointnet_config": {"architecture": {"feature_dim": "[32, 64, 128, 256, 256]"}}}))
    (ae_dir / "ae.json").write_text(json.dumps({"pointnet_config": {"apply_kl_regularization": True, "encoder_config_file": "lv/enc.json",
                                                                  "decoder_config_file": "['lv/d0.json', 'lv/d1.json', 'lv/d2.json']"}}))
    fc = configs.feature_ddpm_config()
    fc["autoencoder_config"] = {"config_file": str(ae_dir / "ae.json"), "ckpt": "unused"}
    feat_cfg = cdir / "feat.json"
    feat_cfg.write_text(json.dumps(_stringify(fc)))
    out2 = tmp_path / "gen2"
    r = subprocess.run([sys.executable, os.path.join(cli, "latent_ddpm_keypoint_conditional_generation.py"), "-c", str(feat_cfg),
                        "--random_init", "--keypoint_file", str(kpf), "--batch_size", "4", "--save_dir", str(out2), "--decode",
                        "--save_keypoint_feature"], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    d2 = np.load(out2 / "shapenet_psr_generated_data_2048_pts.npz")
    assert {"points", "normals", "label", "category", "category_name", "timing", "keypoint", "keypoint_feature"} <= set(d2.files)
    assert d2["points"].shape == (6, 2048, 3) and d2["normals"].shape == (6, 2048, 3) and d2["keypoint_feature"].shape == (6, 16, 48)
    assert np.isfinite(d2["points"]).all() and np.allclose(d2["keypoint"], d["points"])
    # autoencoder_decode_keypoint.py on that file (reference flags / schemas), then the encode -> decode round trip
    ge = load_golden("golden_encode.npz")
    (ae_dir / "lv" / "enc_full.json").write_text(json.dumps({"pointnet_config": _stringify(json.loads(str(ge["encoder_config_json"])))}))
    (ae_dir / "ae_full.json").write_text(json.dumps({"pointnet_config": {"apply_kl_regularization": True, "kl_weight": 1e-5,
                                                                       "encoder_config_file": "lv/enc_full.json",
                                                                       "decoder_config_file": "['lv/d0.json', 'lv/d1.json', 'lv/d2.json']"}}))
    out3 = tmp_path / "rec"
    r = subprocess.run([sys.executable, os.path.join(cli, "autoencoder_decode_keypoint.py"), "-c", str(ae_dir / "ae_full.json"),
                        "--random_init", "--dataset_path", str(out2 / "shapenet_psr_generated_data_2048_pts.npz"),
                        "--save_dir", str(out3), "--batch_size", "4"], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    d3 = np.load(out3 / "reconstructed_pcd.npz")
    assert set(d3.files) == {"points", "normals", "label", "category", "category_name", "keypoint"}
    assert d3["points"].shape == (6, 2048, 3) and np.isfinite(d3["points"]).all() and np.allclose(d3["keypoint"], d["points"])
    vis = out3 / "reconstructed_pcd_visualization"
    assert (vis / "pcd_000_label_00_airplane.xyz").exists() and (vis / "pcd_005_label_00_airplane_keypoint.xyz").exists()
    assert np.loadtxt(vis / "pcd_000_label_00_airplane.xyz").shape == (2048, 6)
    src = tmp_path / "clouds.npz"
    np.savez(src, points=ge["pointcloud"][:, :, :3], normals=ge["pointcloud"][:, :, 3:], keypoint=ge["keypoint"], label=ge["label"])
    out4 = tmp_path / "rec2"
    r = subprocess.run([sys.executable, os.path.join(cli, "autoencoder_decode_keypoint.py"), "-c", str(ae_dir / "ae_full.json"),
                        "--random_init", "--encode_from", str(src), "--save_dir", str(out4), "--batch_size", "2"],
                       env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    d4 = np.load(out4 / "reconstructed_pcd.npz")
    assert d4["keypoint_feature"].shape == (2, 16, 48) and d4["points"].shape == (2, 2048, 3) and np.isfinite(d4["points"]).all()


def test_decode_is_sharded_over_the_ranks(gpu_device, tmp_path):
    """VERDICT r3 item 6 / BASELINE configs[4]: under torch.distributed.run every rank decodes ITS OWN latent shard and the clouds
    are all-gathered (the reference decodes per rank: mesh_evaluation.py:113-118) -- two ranks sharing this box's GPU (gloo,
    SLIDE_SHARE_GPU=1; RCCL over xGMI on a multi-GPU node) with another batch size write the SAME npz as one rank, bit for bit,
    for both CLIs that decode (per-sample arithmetic of the module path; per-shape FPS start indices keyed on the global index)."""
    g = load_golden("golden_decode.npz")
    decs = json.loads(str(g["decoder_configs_json"]))
    ae_dir = tmp_path / "configs" / "ae"
    os.makedirs(ae_dir / "lv")
    for i, dcfg in enumerate(decs):
        (ae_dir / "lv" / ("d%d.json" % i)).write_text(json.dumps({"pointnet_config": _stringify(dcfg)}))
    (ae_dir / "lv" / "enc.json").write_text(json.dumps({"pointnet_config": {"architecture": {"feature_dim": "[32, 64, 128, 256, 256]"}}}))
    (ae_dir / "ae.json").write_text(json.dumps({"pointnet_config": {"apply_kl_regularization": True, "encoder_config_file": "lv/enc.json",
                                                                  "decoder_config_file": "['lv/d0.json', 'lv/d1.json', 'lv/d2.json']"}}))
    cdir = tmp_path / "configs" / "a" / "b"
    os.makedirs(cdir)
    pc = configs.position_ddpm_config()
    pc["shapenet_psr_dataset_config"] = {"dataset": "shapenet_psr_dataset", "categories": ["02691156", "03001627"], "num_keypoints": 16}
    pc["train_config"] = {"task": "keypoint_generation", "dataset": "shapenet_psr_dataset"}
    (cdir / "pos.json").write_text(json.dumps(_stringify(pc)))
    fc = configs.feature_ddpm_config()
    fc["autoencoder_config"] = {"config_file": str(ae_dir / "ae.json"), "ckpt": "unused"}
    (cdir / "feat.json").write_text(json.dumps(_stringify(fc)))
    env = dict(os.environ, PYTHONPATH=REPO)
    env2 = dict(env, SLIDE_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cli = os.path.join(REPO, "pointnet2", "sampling_and_inference")
    two = lambda port: [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port)]
    gen = lambda out, bs: [os.path.join(cli, "latent_ddpm_keypoint_conditional_generation.py"), "-c", str(cdir / "feat.json"),
                           "--random_init", "--position_config", str(cdir / "pos.json"), "--num_samples", "7", "--batch_size", str(bs),
                           "--chains", "2", "--decode", "--save_keypoint_feature", "--save_dir", str(out)]
    r = subprocess.run([sys.executable] + gen(tmp_path / "g1", 8), env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run(two(29621) + gen(tmp_path / "g2", 3), env=env2, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    d1, d2 = (np.load(tmp_path / t / "shapenet_psr_generated_data_2048_pts.npz") for t in ("g1", "g2"))
    assert d1["points"].shape == (7, 2048, 3) and np.isfinite(d1["points"]).all()
    for k in ("points", "normals", "keypoint", "keypoint_feature", "label"):
        assert np.array_equal(d1[k], d2[k]), k
    dec = lambda out, bs: [os.path.join(cli, "autoencoder_decode_keypoint.py"), "-c", str(ae_dir / "ae.json"), "--random_init",
                           "--dataset_path", str(tmp_path / "g1" / "shapenet_psr_generated_data_2048_pts.npz"), "--save_dir", str(out),
                           "--batch_size", str(bs), "--seed", "3"]
    r = subprocess.run([sys.executable] + dec(tmp_path / "r1", 4), env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run(two(29622) + dec(tmp_path / "r2", 2), env=env2, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    e1, e2 = (np.load(tmp_path / t / "reconstructed_pcd.npz") for t in ("r1", "r2"))
    assert e1["points"].shape == (7, 2048, 3)
    for k in ("points", "normals", "keypoint", "label"):
        assert np.array_equal(e1[k], e2[k]), k


def test_five_category_end_to_end(gpu_device):
    """BASELINE configs[3] in miniature: 11 shapes over the five released categories (labels 0, 2, 3, 4, 6), one weight set
    per category, position DDPM -> feature DDPM conditioned on the generated positions, gathered latents"""
    import numpy as np
    import torch
    from slide_amd import configs, model_spec
    from slide_amd.generation import FIVE_CATEGORIES, CategoryChains, category_layout
    from slide_amd.synth import synth_state_dict
    pc, fc = configs.position_ddpm_config(), configs.feature_ddpm_config()
    spec_p, spec_f = model_spec.denoiser_param_spec(pc["pointnet_config"]), model_spec.denoiser_param_spec(fc["pointnet_config"])
    built = []

    def weights(c):
        built.append(c)
        return synth_state_dict(spec_p, seed=100 + c), synth_state_dict(spec_f, seed=200 + c)

    total = 11
    ch = CategoryChains(total, 0, 1, pc, fc, weights, gpu_device, prec="fp16", seed=3)
    assert built == list(FIVE_CATEGORIES) and [(c, lo, hi) for c, lo, hi, _, _ in ch.chains] == category_layout(total)
    lat, labels = ch.generate(steps=6)
    lat = lat.cpu().numpy()
    assert lat.shape == (total, 16, 51) and np.isfinite(lat).all()
    assert labels.tolist() == [0, 0, 0, 2, 2, 3, 3, 4, 4, 6, 6]
    # the feature chains were conditioned on the positions their position chains produced
    for c, lo, hi, ps, fs in ch.chains:
        assert np.array_equal(lat[lo:hi, :, :3], ps.state().cpu().numpy())
    # different categories = different weights: the same x_T gives different positions
    a = ch.chains[0][3].sample(np.zeros(3, np.int64), np.ones((3, 16, 3), np.float32) * 0.1, t_start=10, n_steps=2).cpu().numpy()
    b = ch.chains[1][3].sample(np.zeros(2, np.int64), np.ones((2, 16, 3), np.float32) * 0.1, t_start=10, n_steps=2).cpu().numpy()
    assert not np.allclose(a[:2], b)


def test_released_checkpoint_layout_through_the_clis(gpu_device, tmp_path):
    """the reference's pickle layout (train.py:243-255: model_state_dict + ema_state_list per EMA rate +
    optimizer state) loaded by --ckpt / --ema_idx through both generation CLIs and the autoencoder checkpoint through
    --ae_ckpt: the CLI output must equal a direct sampler run on the EMA-OVERLAID weights (and differ from the raw ones)"""
    import torch
    from slide_amd import model_spec
    from slide_amd.diffusion import PositionSampler
    from slide_amd.synth import synth_state_dict
    pc = configs.position_ddpm_config()
    hp = pc["pointnet_config"]
    spec = model_spec.denoiser_param_spec(hp)
    raw, ema0, ema1 = (synth_state_dict(spec, seed=s_) for s_ in (11, 12, 13))
    ck = tmp_path / "pointnet_ckpt_42.pkl"
    # EMA lists hold only trainable parameters (data_utils/ema.py:13-18): leave one tensor out of the EMA dicts on purpose
    skip = "class_emb.weight"
    torch.save({"model_state_dict": {k: torch.from_numpy(v) for k, v in raw.items()},
                "ema_state_list": [{k: torch.from_numpy(v) for k, v in e_.items() if k != skip} for e_ in (ema0, ema1)],
                "optimizer_state_dict": {}, "iter": 42, "training_time_seconds": 1.0}, ck)
    cdir = tmp_path / "configs" / "a" / "b"
    os.makedirs(cdir)
    pc["shapenet_psr_dataset_config"] = {"dataset": "shapenet_psr_dataset", "categories": ["02691156"], "num_keypoints": 16}
    pc["train_config"] = {"task": "keypoint_generation", "dataset": "shapenet_psr_dataset"}
    (cdir / "pos.json").write_text(json.dumps(_stringify(pc)))
    env = dict(os.environ, PYTHONPATH=REPO)
    cli = os.path.join(REPO, "pointnet2", "sampling_and_inference")
    out1 = tmp_path / "gen"
    r = subprocess.run([sys.executable, os.path.join(cli, "point_cloud_generation.py"), "-c", str(cdir / "pos.json"), "--ckpt", str(ck),
                        "--ema_idx", "1", "--num_samples", "4", "--batch_size", "4", "--save_dir", str(out1), "--seed", "5",
                        "--prec", "fp32"], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.load(out1 / "shapenet_psr_generated_data_16_pts.npz")["points"]
    # --prec split (round 4: fp32 storage, contractions as two-term fp16 operand splits): the same run within 1e-3 of the fp32 one
    # over the whole 1000-step chain (fp32-grade forwards; the chains still see different last-bit rounding at every step)
    out1s = tmp_path / "gen_split"
    r = subprocess.run([sys.executable, os.path.join(cli, "point_cloud_generation.py"), "-c", str(cdir / "pos.json"), "--ckpt", str(ck),
                        "--ema_idx", "1", "--num_samples", "4", "--batch_size", "4", "--save_dir", str(out1s), "--seed", "5",
                        "--prec", "split"], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    got_s = np.load(out1s / "shapenet_psr_generated_data_16_pts.npz")["points"]
    assert np.isfinite(got_s).all() and np.abs(got_s - got).max() <= 1e-3 * np.abs(got).max(), np.abs(got_s - got).max()
    want_sd = dict(raw)
    want_sd.update({k: v for k, v in ema1.items() if k != skip})
    smp = PositionSampler(hp, want_sd, 4, gpu_device, pc["diffusion_config"], prec="fp32", seed=5)
    from slide_amd.generation import start_noise
    # the CLI's conventions: start noise of shape i = start_noise(seed, 1 (position) | 2 (feature), i ...), chain nonce 1 | 2,
    # in-kernel noise keyed on the shape's global index
    def run_pos(s_, w_):
        s_.begin(np.zeros(4, np.int64), start_noise(5, 1, 0, 4, (16, 3), gpu_device), nonce=1, sample_offset=0)
        s_.advance(s_.T)
        return s_.state().cpu().numpy()
    want = run_pos(smp, want_sd)
    assert np.array_equal(got, want)
    other = run_pos(PositionSampler(hp, raw, 4, gpu_device, pc["diffusion_config"], prec="fp32", seed=5), raw)
    assert not np.allclose(got, other)
    # feature CLI: denoiser --ckpt (EMA 0) + autoencoder --ae_ckpt (model_state_dict of the whole autoencoder), --decode
    fc = configs.feature_ddpm_config()
    fspec = model_spec.denoiser_param_spec(fc["pointnet_config"])
    fraw, fema = synth_state_dict(fspec, seed=21), synth_state_dict(fspec, seed=22)
    fck = tmp_path / "latent_ckpt.pkl"
    torch.save({"model_state_dict": {k: torch.from_numpy(v) for k, v in fraw.items()},
                "ema_state_list": [{k: torch.from_numpy(v) for k, v in fema.items()}], "optimizer_state_dict": {}, "iter": 1}, fck)
    g = load_golden("golden_decode.npz")
    decs = json.loads(str(g["decoder_configs_json"]))
    ae_dir = tmp_path / "configs" / "ae"
    os.makedirs(ae_dir / "lv")
    for i, dcfg in enumerate(decs):
        (ae_dir / "lv" / ("d%d.json" % i)).write_text(json.dumps({"pointnet_config": _stringify(dcfg)}))
    (ae_dir / "lv" / "enc.json").write_text(json.dumps({"pointnet_config": {"architecture": {"feature_dim": "[32, 64, 128, 256, 256]"}}}))
    (ae_dir / "ae.json").write_text(json.dumps({"pointnet_config": {"apply_kl_regularization": True, "encoder_config_file": "lv/enc.json",
                                                                  "decoder_config_file": "['lv/d0.json', 'lv/d1.json', 'lv/d2.json']"}}))
    aspec = [(str(n), tuple(int(x) for x in str(s_).split(","))) for n, s_ in zip(g["spec_names"], g["spec_shapes"])]
    avals = synth_state_dict(aspec, seed=31)
    ack = tmp_path / "ae_ckpt.pkl"
    # a released autoencoder checkpoint also carries encoder weights: extra keys must be ignored by the decode-only model
    asd = {n: torch.from_numpy(v) for n, v in avals.items()}
    asd["encoder.some_unused.weight"] = torch.zeros(3)
    torch.save({"model_state_dict": asd, "iter": 7}, ack)
    fc["autoencoder_config"] = {"config_file": str(ae_dir / "ae.json"), "ckpt": str(ack)}
    (cdir / "feat.json").write_text(json.dumps(_stringify(fc)))
    out2 = tmp_path / "gen2"
    r = subprocess.run([sys.executable, os.path.join(cli, "latent_ddpm_keypoint_conditional_generation.py"), "-c", str(cdir / "feat.json"),
                        "--ckpt", str(fck), "--ema_idx", "0", "--keypoint_file", str(out1 / "shapenet_psr_generated_data_16_pts.npz"),
                        "--batch_size", "4", "--save_dir", str(out2), "--decode", "--save_keypoint_feature", "--seed", "3",
                        "--prec", "fp32", "--chains", "1"], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    d2 = np.load(out2 / "shapenet_psr_generated_data_2048_pts.npz")
    assert d2["points"].shape == (4, 2048, 3) and np.isfinite(d2["points"]).all() and np.allclose(d2["keypoint"], got)
    from slide_amd.diffusion import FeatureSampler
    fs = FeatureSampler(fc["pointnet_config"], fema, 4, gpu_device, fc["standard_diffusion_config"], prec="fp32", seed=3)
    fs.begin(np.zeros(4, np.int64), got, start_noise(3, 2, 0, 4, (16, 51), gpu_device), nonce=2, sample_offset=0)
    fs.advance(fs.T)
    lat = fs.state().cpu().numpy()
    assert np.array_equal(d2["keypoint_feature"], lat[:, :, 3:])


def test_clis_drive_the_benched_arrangement(gpu_device, tmp_path):
    """both generation CLIs run the arrangement bench.py times -- independent chains replayed round-robin by
    one library call (feature sub-batch chains of a batch; position batches in flight; on the fly: the position chain of batch
    i + 1 beside the feature chains of batch i), fp16 operands by default -- and their output is BIT-IDENTICAL to running the
    same chains one after the other (--serial_chains): the chains are independent objects of the sample partition."""
    cdir = tmp_path / "configs" / "a" / "b"
    os.makedirs(cdir)
    pc = configs.position_ddpm_config()
    pc["shapenet_psr_dataset_config"] = {"dataset": "shapenet_psr_dataset", "categories": ["02691156", "03001627"], "num_keypoints": 16}
    pc["train_config"] = {"task": "keypoint_generation", "dataset": "shapenet_psr_dataset"}
    (cdir / "pos.json").write_text(json.dumps(_stringify(pc)))
    fc = configs.feature_ddpm_config()
    (cdir / "feat.json").write_text(json.dumps(_stringify(fc)))
    env = dict(os.environ, PYTHONPATH=REPO)
    cli = os.path.join(REPO, "pointnet2", "sampling_and_inference")
    outs = {}
    for tag, extra in (("par", []), ("ser", ["--serial_chains"])):
        o1 = tmp_path / ("pos_" + tag)
        r = subprocess.run([sys.executable, os.path.join(cli, "point_cloud_generation.py"), "-c", str(cdir / "pos.json"), "--random_init",
                            "--num_samples", "40", "--batch_size", "8", "--chains", "3", "--save_dir", str(o1)] + extra,
                           env=env, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        # end-to-end rate of the run; default arithmetic "mixed" (round 5): position DDPM split (fp32-grade), feature DDPM fp16 operands
        assert "shapes/s" in r.stdout and "mixed" in r.stdout
        outs["pos_" + tag] = np.load(o1 / "shapenet_psr_generated_data_16_pts.npz")
        o2 = tmp_path / ("feat_" + tag)
        r = subprocess.run([sys.executable, os.path.join(cli, "latent_ddpm_keypoint_conditional_generation.py"), "-c", str(cdir / "feat.json"),
                            "--random_init", "--keypoint_file", str(o1 / "shapenet_psr_generated_data_16_pts.npz"), "--batch_size", "24",
                            "--chains", "3", "--save_dir", str(o2)] + extra, env=env, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "8/8/8 feature chain(s)" in r.stdout
        outs["feat_" + tag] = np.load(o2 / "shapenet_psr_generated_data_16_pts_latents.npz")
        o3 = tmp_path / ("fly_" + tag)
        r = subprocess.run([sys.executable, os.path.join(cli, "latent_ddpm_keypoint_conditional_generation.py"), "-c", str(cdir / "feat.json"),
                            "--random_init", "--position_config", str(cdir / "pos.json"), "--num_samples", "40", "--batch_size", "16",
                            "--chains", "2", "--save_dir", str(o3)] + extra, env=env, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "position + feature DDPM" in r.stdout
        outs["fly_" + tag] = np.load(o3 / "shapenet_psr_generated_data_16_pts_latents.npz")
    assert outs["pos_par"]["points"].shape == (40, 16, 3) and np.isfinite(outs["pos_par"]["points"]).all()
    assert (outs["pos_par"]["label"] == np.array([0, 4] * 20)).all()
    assert np.array_equal(outs["pos_par"]["points"], outs["pos_ser"]["points"])
    for k in ("feat", "fly"):
        a_, b_ = outs[k + "_par"], outs[k + "_ser"]
        assert a_["keypoint_feature"].shape == (40, 16, 48) and np.isfinite(a_["keypoint_feature"]).all()
        assert np.array_equal(a_["keypoint_feature"], b_["keypoint_feature"]) and np.array_equal(a_["points"], b_["points"])
    # the feature chains were conditioned on the key points of the file / of the on-the-fly position chains
    assert np.array_equal(outs["feat_par"]["points"], outs["pos_par"]["points"])
    # no two shapes coincide (noise keyed on the global shape index)
    p_ = outs["pos_par"]["points"].reshape(40, -1)
    assert len({p_[i].tobytes() for i in range(40)}) == 40
    # the same commands under torch.distributed.run with TWO ranks (sharing the one GPU of this box over gloo,
    # SLIDE_SHARE_GPU=1; RCCL over xGMI on a multi-GPU node) and with a different batch size / chain count give the SAME files:
    # a shape's start noise and in-kernel noise depend on (seed, global index) only
    env2 = dict(env, SLIDE_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    o1 = tmp_path / "pos_2rank"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29611", os.path.join(cli, "point_cloud_generation.py"), "-c", str(cdir / "pos.json"), "--random_init",
                        "--num_samples", "40", "--batch_size", "4", "--chains", "2", "--save_dir", str(o1)], env=env2, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert np.array_equal(np.load(o1 / "shapenet_psr_generated_data_16_pts.npz")["points"], outs["pos_par"]["points"])
    o3 = tmp_path / "fly_2rank"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29612", os.path.join(cli, "latent_ddpm_keypoint_conditional_generation.py"), "-c", str(cdir / "feat.json"),
                        "--random_init", "--position_config", str(cdir / "pos.json"), "--num_samples", "40", "--batch_size", "8",
                        "--chains", "3", "--save_dir", str(o3)], env=env2, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    d_ = np.load(o3 / "shapenet_psr_generated_data_16_pts_latents.npz")
    assert np.array_equal(d_["points"], outs["fly_par"]["points"]) and np.array_equal(d_["keypoint_feature"], outs["fly_par"]["keypoint_feature"])
