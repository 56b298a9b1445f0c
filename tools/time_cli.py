"""End-to-end rate of the generation CLIs in the benched arrangement: writes the shipped DDPM configs in the
reference's JSON format to a temp dir and runs
  latent_ddpm_keypoint_conditional_generation.py --position_config ... (position + feature DDPM, key points on the fly)
  point_cloud_generation.py                                         (position DDPM only)
with --random_init at batch 256; prints their own `shapes/s` lines (the timers exclude model construction)."""
import json, os, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from slide_amd import configs


def stringify(d):
    return {k: stringify(v) if isinstance(v, dict) else (str(v) if isinstance(v, list) else v) for k, v in d.items()}


n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
tmp = tempfile.mkdtemp()
pc = configs.position_ddpm_config()
pc["shapenet_psr_dataset_config"] = {"dataset": "shapenet_psr_dataset", "categories": ["02691156"], "num_keypoints": 16}
pc["train_config"] = {"task": "keypoint_generation", "dataset": "shapenet_psr_dataset"}
json.dump(stringify(pc), open(os.path.join(tmp, "pos.json"), "w"))
json.dump(stringify(configs.feature_ddpm_config()), open(os.path.join(tmp, "feat.json"), "w"))
env = dict(os.environ, PYTHONPATH=REPO)
cli = os.path.join(REPO, "pointnet2", "sampling_and_inference")
for cmd in ([os.path.join(cli, "latent_ddpm_keypoint_conditional_generation.py"), "-c", os.path.join(tmp, "feat.json"), "--random_init",
             "--position_config", os.path.join(tmp, "pos.json"), "--num_samples", str(n), "--batch_size", "256", "--save_dir", os.path.join(tmp, "o1")],
            [os.path.join(cli, "point_cloud_generation.py"), "-c", os.path.join(tmp, "pos.json"), "--random_init", "--num_samples", str(n),
             "--batch_size", "256", "--save_dir", os.path.join(tmp, "o2")]):
    r = subprocess.run([sys.executable] + cmd, env=env, capture_output=True, text=True)
    print("\n".join(l for l in r.stdout.split("\n") if "shapes/s" in l) or r.stderr[-1500:])
