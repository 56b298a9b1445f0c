"""wall-clock of graph-replayed reverse steps: feature chain alone, position chain alone, both on two streams"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from slide_amd import configs, model_spec
from slide_amd.diffusion import FeatureSampler, JointSampler, PositionSampler
from slide_amd.synth import synth_keypoints, synth_state_dict
dev = torch.device("cuda:0"); B = int(os.environ.get("B", 256)); prec = os.environ.get("PREC", "fp16")
pc, fc = configs.position_ddpm_config(), configs.feature_ddpm_config()
pos = PositionSampler(pc["pointnet_config"], synth_state_dict(model_spec.denoiser_param_spec(pc["pointnet_config"])), B, dev, pc["diffusion_config"], prec=prec)
feat = FeatureSampler(fc["pointnet_config"], synth_state_dict(model_spec.denoiser_param_spec(fc["pointnet_config"])), B, dev, fc["standard_diffusion_config"], prec=prec)
rs = np.random.RandomState(0)
def reset():
    pos.begin(np.zeros(B, np.int64), rs.standard_normal((B, 16, 3)).astype(np.float32))
    feat.begin(np.full(B, 4, np.int64), synth_keypoints(B), rs.standard_normal((B, 16, 51)).astype(np.float32))
def sync():
    pos.stream.synchronize(); feat.stream.synchronize(); torch.cuda.synchronize()
N = 300
joint = JointSampler(pos, feat)
for name, fn in (("feat", lambda: feat.advance(N)), ("pos", lambda: pos.advance(N)), ("both", lambda: (pos.advance(N), feat.advance(N))),
                 ("joint", lambda: joint.advance(N))):
    reset(); fn(); sync(); reset(); sync()
    t0 = time.perf_counter(); fn(); sync(); dt = time.perf_counter() - t0
    print("%-5s %.3f ms/step" % (name, dt * 1e3 / N))
