"""two half-batch joint samplers replayed concurrently (separate stream pairs) vs one full-batch joint sampler"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from slide_amd import configs, model_spec
from slide_amd.diffusion import FeatureSampler, JointSampler, PositionSampler
from slide_amd.synth import synth_keypoints, synth_state_dict
dev = torch.device("cuda:0"); prec = "fp16"
pc, fc = configs.position_ddpm_config(), configs.feature_ddpm_config()
psd = synth_state_dict(model_spec.denoiser_param_spec(pc["pointnet_config"]))
fsd = synth_state_dict(model_spec.denoiser_param_spec(fc["pointnet_config"]))
rs = np.random.RandomState(0)
def make(B):
    pos = PositionSampler(pc["pointnet_config"], psd, B, dev, pc["diffusion_config"], prec=prec)
    feat = FeatureSampler(fc["pointnet_config"], fsd, B, dev, fc["standard_diffusion_config"], prec=prec)
    def reset():
        pos.begin(np.zeros(B, np.int64), rs.standard_normal((B, 16, 3)).astype(np.float32))
        feat.begin(np.full(B, 4, np.int64), synth_keypoints(B), rs.standard_normal((B, 16, 51)).astype(np.float32))
    return pos, feat, JointSampler(pos, feat), reset
N = 300
for parts in (1, 2, 3):
    B = 256 // parts
    S = [make(256 - B * (parts - 1) if i == 0 else B) for i in range(parts)]
    def run(n):
        # interleave launches so the graphs of the sub-batches are in flight together
        for _ in range(n):
            for s in S: s[2].advance(1)
    for s in S: s[3]()
    run(5); torch.cuda.synchronize()
    for s in S: s[3]()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); run(N); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%d sub-batches of %3d: %.3f ms/step (256 shapes)" % (parts, B, dt * 1e3 / N))
