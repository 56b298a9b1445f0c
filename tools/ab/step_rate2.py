"""what state carries over a reset + sync?  pre-phase = a list of advance() call sizes, then reset + sync, then 4 blocks of 10 steps
timed with HIP events on every chain's stream.  usage: python tools/ab/step_rate2.py "55" "5" "10x6" "5,50" ..."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from slide_amd import configs, model_spec
from slide_amd.diffusion import EagerChainsSampler, FeatureSampler, PositionSampler
from slide_amd.synth import synth_keypoints, synth_state_dict
dev = torch.device("cuda:0")
pc, fc = configs.position_ddpm_config(), configs.feature_ddpm_config()
sd_p = synth_state_dict(model_spec.denoiser_param_spec(pc["pointnet_config"]))
sd_f = synth_state_dict(model_spec.denoiser_param_spec(fc["pointnet_config"]))
pos = PositionSampler(pc["pointnet_config"], sd_p, 256, dev, pc["diffusion_config"], prec="fp16", seed=1, use_graph=False)
feats = [FeatureSampler(fc["pointnet_config"], sd_f, b, dev, fc["standard_diffusion_config"], prec="fp16", seed=2 + i, use_graph=False)
         for i, b in enumerate((88, 88, 80))]
chains = feats[:1] + [pos] + feats[1:]
joint = EagerChainsSampler(chains)
gen = torch.Generator(device=dev); gen.manual_seed(0)
def reset():
    pos.begin(torch.zeros(256, dtype=torch.int64, device=dev), torch.randn(256, 16, 3, device=dev, generator=gen))
    for f_, b in zip(feats, (88, 88, 80)):
        f_.begin(torch.full((b,), 4, dtype=torch.int64, device=dev), torch.as_tensor(synth_keypoints(b), device=dev),
                 torch.randn(b, 16, 51, device=dev, generator=gen))
def sync():
    for s_ in chains:
        s_.stream.synchronize()
    torch.cuda.synchronize(dev)
reset(); joint.advance(100); sync()   # cold start out of the way
for spec in sys.argv[1:]:
    for rep in range(3):
        reset(); sync(); time.sleep(0.2)   # settle: every variant starts from the same idle device
        reset()
        for part in spec.split(","):
            if "x" in part:
                n, k = part.split("x")
                for _ in range(int(k)):
                    joint.advance(int(n))
            elif int(part) > 0:
                joint.advance(int(part))
        reset(); sync()
        t0 = time.perf_counter()
        joint.advance(20)
        sync()
        dt = (time.perf_counter() - t0) * 1e3 / 20
        print("pre [%s] rep %d: 20 timed steps %.4f ms per step" % (spec, rep, dt), flush=True)
