"""fp16-mode position chains against the exact-fp32 mode (equal in-kernel noise): per-shape relative max distance after n steps"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from slide_amd import configs, model_spec
from slide_amd.diffusion import PositionSampler
from slide_amd.synth import synth_state_dict
dev = torch.device("cuda:0")
pc = configs.position_ddpm_config()
sd = synth_state_dict(model_spec.denoiser_param_spec(pc["pointnet_config"]))
B = 256
for seed in (77, 5):
    for n in (20, 200, 1000):
        res = {}
        for prec in ("fp16", "fp32"):
            ps = PositionSampler(pc["pointnet_config"], sd, B, dev, pc["diffusion_config"], prec=prec, seed=seed, use_graph=True)
            xT = np.random.RandomState(seed).standard_normal((B, 16, 3)).astype(np.float32)
            res[prec] = ps.sample(np.zeros(B, np.int64), xT, t_start=n - 1, n_steps=n).cpu().numpy().reshape(B, -1)
        per = np.abs(res["fp16"] - res["fp32"]).max(axis=1) / np.abs(res["fp32"]).max()
        print("seed %d, last %4d steps: median %.2e  95%% %.2e  max %.2e" % (seed, n, np.median(per), np.quantile(per, 0.95), per.max()), flush=True)
