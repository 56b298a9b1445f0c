"""position plan as its OWN step graph on its own stream beside the two feature sub-batch graphs, against the default
(position plan as a branch of the first sub-batch's graph)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from slide_amd import configs, model_spec
from slide_amd.diffusion import FeatureSampler, JointSampler, PositionSampler, SplitJointSampler
from slide_amd.synth import synth_keypoints, synth_state_dict
dev = torch.device("cuda:0"); prec = "fp16"; B = 256
pc, fc = configs.position_ddpm_config(), configs.feature_ddpm_config()
psd = synth_state_dict(model_spec.denoiser_param_spec(pc["pointnet_config"]))
fsd = synth_state_dict(model_spec.denoiser_param_spec(fc["pointnet_config"]))
rs = np.random.RandomState(0)
pos = PositionSampler(pc["pointnet_config"], psd, B, dev, pc["diffusion_config"], prec=prec)
fs = [FeatureSampler(fc["pointnet_config"], fsd, B // 2, dev, fc["standard_diffusion_config"], prec=prec, seed=i) for i in range(2)]
def reset():
    pos.begin(np.zeros(B, np.int64), rs.standard_normal((B, 16, 3)).astype(np.float32))
    for f in fs:
        f.begin(np.full(B // 2, 4, np.int64), synth_keypoints(B // 2), rs.standard_normal((B // 2, 16, 51)).astype(np.float32))
def sync():
    pos.stream.synchronize(); [f.stream.synchronize() for f in fs]; torch.cuda.synchronize()
N = 500
# (a) three independent graphs, launched round-robin
reset(); pos.advance(5); [f.advance(5) for f in fs]; sync()
reset(); sync(); t0 = time.perf_counter()
for _ in range(N):
    fs[0].advance(1); pos.advance(1); fs[1].advance(1)
sync(); ta = (time.perf_counter() - t0) / N
# (b) default: position plan as a branch of sub-batch 0's graph
joint = SplitJointSampler([JointSampler(pos, fs[0]), JointSampler(None, fs[1])])
reset(); joint.advance(5); sync()
reset(); sync(); t0 = time.perf_counter()
joint.advance(N)
sync(); tb = (time.perf_counter() - t0) / N
print("three graphs %.4f ms/step   branch-of-first-graph %.4f ms/step" % (ta * 1e3, tb * 1e3))
