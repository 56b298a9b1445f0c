"""Odd batch sizes through the default fp16 plans (ragged row tiles, partial XCD groups): forward vs the fp32 engine of the same
batch, and a few sampler steps of both DDPMs (finite, batch-size independent)."""
import numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from slide_amd import configs, model_spec
from slide_amd.diffusion import PositionSampler, FeatureSampler
from slide_amd.engine import DenoiserEngine
from slide_amd.generation import start_noise
from slide_amd.synth import synth_state_dict, synth_keypoints
dev = torch.device("cuda:0")
worst = 0.0
for name, cfgf in (("pos", configs.position_ddpm_config), ("feat", configs.feature_ddpm_config)):
    c = cfgf(); hp = c["pointnet_config"]
    sd = synth_state_dict(model_spec.denoiser_param_spec(hp))
    rs = np.random.RandomState(0)
    for B in (1, 3, 5, 17, 37, 100):
        x = rs.standard_normal((B, 16, 3 + hp["in_fea_dim"])).astype(np.float32)
        ts = rs.randint(0, 1000, B).astype(np.float32); lab = np.zeros(B, np.int64)
        y16 = DenoiserEngine(hp, sd, B, dev, prec="fp16").forward(x, ts, lab).cpu().numpy()
        y32 = DenoiserEngine(hp, sd, B, dev, prec="fp32").forward(x, ts, lab).cpu().numpy()
        rel = float(np.linalg.norm(y16 - y32) / np.linalg.norm(y32))
        worst = max(worst, rel)
        print(name, "B=%d" % B, "fp16 vs fp32 rel L2 %.2e" % rel, "finite", bool(np.isfinite(y16).all()))
        assert np.isfinite(y16).all() and rel < 5e-3
c = configs.feature_ddpm_config(); hp = c["pointnet_config"]; sd = synth_state_dict(model_spec.denoiser_param_spec(hp))
ref = None
for B in (5, 37):
    s = FeatureSampler(hp, sd, B, dev, c["standard_diffusion_config"], prec="fp16", seed=3, use_graph=False)
    kp = synth_keypoints(37)[:B]
    s.begin(np.full(B, 4, np.int64), kp, start_noise(0, 2, 0, B, (16, 51), dev), nonce=2, sample_offset=0)
    s.advance(12)
    st = s.state().cpu().numpy()
    assert np.isfinite(st).all()
    if ref is None:
        ref = st
    else:
        print("feature sampler 12 steps, B=37 vs B=5 (first five shapes): equal", np.array_equal(st[:5], ref), float(np.abs(st[:5] - ref).max()))
print("worst fp16 forward rel L2", worst)
