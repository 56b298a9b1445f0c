"""which arrays differ between the one-rank and the two-rank run of tests/test_hip_cli.py::test_decode_is_sharded_over_the_ranks
(all keys, max abs difference, per-shape)"""
import json, os, subprocess, sys, tempfile
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from test_hip_cli import load_golden, _stringify  # noqa
from slide_amd import configs
from pathlib import Path
tmp_path = Path(tempfile.mkdtemp())
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
                       "--chains", "2", "--decode", "--save_keypoint_feature", "--save_dir", str(out)] + sys.argv[1:]
runs = {"g1": ([sys.executable] + gen(tmp_path / "g1", 8), env), "g1b": ([sys.executable] + gen(tmp_path / "g1b", 8), env),
        "g3": ([sys.executable] + gen(tmp_path / "g3", 3), env), "g2": (two(29621) + gen(tmp_path / "g2", 3), env2)}
d = {}
for k, (cmd, e) in runs.items():
    r = subprocess.run(cmd, env=e, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    d[k] = np.load(tmp_path / k / "shapenet_psr_generated_data_2048_pts.npz")
for other in ("g1b", "g3", "g2"):
    print("== g1 (one rank, batch 8) vs %s" % other)
    for k in ("label", "keypoint", "keypoint_feature", "points", "normals"):
        a, b = d["g1"][k].astype(np.float64), d[other][k].astype(np.float64)
        diff = np.abs(a - b).reshape(a.shape[0], -1).max(1)
        print("  %-18s equal=%s  per-shape max abs diff: %s" % (k, np.array_equal(d["g1"][k], d[other][k]), np.array2string(diff, precision=2)))
