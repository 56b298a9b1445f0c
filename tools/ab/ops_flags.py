import sys, ctypes
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from slide_amd import configs, model_spec
from slide_amd.diffusion import FeatureSampler
from slide_amd.engine import SlideEpi
from slide_amd.synth import synth_keypoints, synth_state_dict
dev = torch.device("cuda:0"); B = 88
c = configs.feature_ddpm_config()
s = FeatureSampler(c["pointnet_config"], synth_state_dict(model_spec.denoiser_param_spec(c["pointnet_config"])), B, dev, c["standard_diffusion_config"], prec="split")
for i in range(len(s.step_ops)):
    o = s.step_ops[i]
    if o.kind != 1: continue
    n_cob = o.i[3]
    raw = (ctypes.c_char * (ctypes.sizeof(SlideEpi) * n_cob))()
    torch.cuda.synchronize()
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy(raw, ctypes.c_void_p(o.p[2]), ctypes.sizeof(raw), 2)
    eps = (SlideEpi * n_cob).from_buffer_copy(bytes(raw))
    modes = sorted(set((e.mode, e.flags, bool(e.pre_add), bool(e.residual), bool(e.addvec), e.pre_add_shift) for e in eps))
    print(i, "rows", o.i[0], "K", o.i[2], "N", n_cob * 32, "npx", o.i[4], "in_scale", bool(o.p[3]), "f", list(o.f), "epi (mode, flags, pre_add, resid, addvec, pre_shift):", modes)
