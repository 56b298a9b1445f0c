"""wall-clock ms/step of bench.py's eager round-robin arrangement with / without the position chain and for several feature
sub-batch counts (marginal cost of each chain)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from slide_amd import configs, model_spec
from slide_amd.diffusion import EagerChainsSampler, FeatureSampler, PositionSampler
from slide_amd.synth import synth_keypoints, synth_state_dict
dev = torch.device("cuda:0"); B = 256
pc, fc = configs.position_ddpm_config(), configs.feature_ddpm_config()
sd_p = synth_state_dict(model_spec.denoiser_param_spec(pc["pointnet_config"]))
sd_f = synth_state_dict(model_spec.denoiser_param_spec(fc["pointnet_config"]))
rs = np.random.RandomState(0)
pos = PositionSampler(pc["pointnet_config"], sd_p, B, dev, pc["diffusion_config"], prec="fp16", use_graph=False)
N = 300
for sizes in ([256], [128, 128], [88, 88, 80], [64, 64, 64, 64]):
    fs = [FeatureSampler(fc["pointnet_config"], sd_f, b, dev, fc["standard_diffusion_config"], prec="fp16", use_graph=False, seed=i) for i, b in enumerate(sizes)]
    for with_pos in (False, True):
        chains = fs[:1] + ([pos] if with_pos else []) + fs[1:]
        j = EagerChainsSampler(chains)
        def reset():
            pos.begin(np.zeros(B, np.int64), rs.standard_normal((B, 16, 3)).astype(np.float32))
            for f, b in zip(fs, sizes):
                f.begin(np.full(b, 4, np.int64), synth_keypoints(b), rs.standard_normal((b, 16, 51)).astype(np.float32))
        reset(); j.advance(20); j.synchronize(); reset(); j.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter(); j.advance(N); j.synchronize(); dt = time.perf_counter() - t0
        print("feature sub-batches %-18s pos %-5s  %.3f ms/step" % (sizes, with_pos, dt * 1e3 / N))
    del fs
