import numpy as np, torch, sys, os
sys.path.insert(0, "/root/repo")
from slide_amd import configs, model_spec
from slide_amd.diffusion import PositionSampler, FeatureSampler
from slide_amd.engine import DenoiserEngine
from slide_amd.generation import start_noise
from slide_amd.synth import synth_state_dict, synth_keypoints
dev = torch.device("cuda:0")
for name, cfgf in (("pos", configs.position_ddpm_config), ("feat", configs.feature_ddpm_config)):
    c = cfgf(); hp = c["pointnet_config"]
    sd = synth_state_dict(model_spec.denoiser_param_spec(hp))
    rs = np.random.RandomState(0)
    x = rs.standard_normal((8, 16, 3 + hp["in_fea_dim"])).astype(np.float32)
    ts = np.full(8, 500, np.float32); lab = np.zeros(8, np.int64)
    for prec in ("fp32", "fp16"):
        y8 = DenoiserEngine(hp, sd, 8, dev, prec=prec).forward(x, ts, lab).cpu().numpy()
        y4 = DenoiserEngine(hp, sd, 4, dev, prec=prec).forward(x[:4], ts[:4], lab[:4]).cpu().numpy()
        y3 = DenoiserEngine(hp, sd, 3, dev, prec=prec).forward(x[1:4], ts[1:4], lab[1:4]).cpu().numpy()
        print(name, prec, "forward B=8 vs B=4 equal:", np.array_equal(y8[:4], y4), np.abs(y8[:4]-y4).max(), " B=3 shifted:", np.array_equal(y8[1:4], y3))
c = configs.position_ddpm_config(); hp = c["pointnet_config"]; sd = synth_state_dict(model_spec.denoiser_param_spec(hp))
outs = {}
for B in (4, 8):
    s = PositionSampler(hp, sd, B, dev, c["diffusion_config"], prec="fp16", seed=0, use_graph=False)
    s.begin(np.zeros(B, np.int64), start_noise(0, 1, 0, B, (16, 3), dev), nonce=1, sample_offset=0)
    s.advance(30)
    outs[B] = s.state().cpu().numpy()
print("sampler 30 steps B=8 vs 4:", np.array_equal(outs[8][:4], outs[4]), np.abs(outs[8][:4]-outs[4]).max())
s = PositionSampler(hp, sd, 4, dev, c["diffusion_config"], prec="fp16", seed=0, use_graph=False)
s.begin(np.zeros(4, np.int64), start_noise(0, 1, 4, 8, (16, 3), dev), nonce=1, sample_offset=4)
s.advance(30)
print("sampler offset 4:", np.array_equal(outs[8][4:], s.state().cpu().numpy()))
