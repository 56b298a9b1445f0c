import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from slide_amd import configs, model_spec
from slide_amd.experiments.resident import ResidentPositionSampler
from slide_amd.synth import synth_state_dict
dev = torch.device("cuda:0")
B = 256
pc = configs.position_ddpm_config()
sd = synth_state_dict(model_spec.denoiser_param_spec(pc["pointnet_config"]))
rs = np.random.RandomState(0)
xT = rs.standard_normal((B, 16, 3)).astype(np.float32)
r = ResidentPositionSampler(pc["pointnet_config"], sd, B, dev, pc["diffusion_config"], seed=5)
r.begin(np.zeros(B, np.int64), xT)
r.advance(20)
r.stream.synchronize()
