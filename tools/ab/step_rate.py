"""Rate of bench.py's arrangement over time from a cold start: HIP events on one feature chain's stream every 10 joint steps
(the host enqueues ahead; the events time the GPU's progress).  usage: python tools/ab/step_rate.py [steps=600] [idle_ms_before=0]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from slide_amd import configs, model_spec
from slide_amd.diffusion import EagerChainsSampler, FeatureSampler, PositionSampler
from slide_amd.synth import synth_keypoints, synth_state_dict
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
pc, fc = configs.position_ddpm_config(), configs.feature_ddpm_config()
sd_p = synth_state_dict(model_spec.denoiser_param_spec(pc["pointnet_config"]))
sd_f = synth_state_dict(model_spec.denoiser_param_spec(fc["pointnet_config"]))
pos = PositionSampler(pc["pointnet_config"], sd_p, 256, dev, pc["diffusion_config"], prec="fp16", seed=1, use_graph=False)
feats = [FeatureSampler(fc["pointnet_config"], sd_f, b, dev, fc["standard_diffusion_config"], prec="fp16", seed=2 + i, use_graph=False)
         for i, b in enumerate((88, 88, 80))]
joint = EagerChainsSampler(feats[:1] + [pos] + feats[1:])
gen = torch.Generator(device=dev); gen.manual_seed(0)
def reset():
    pos.begin(torch.zeros(256, dtype=torch.int64, device=dev), torch.randn(256, 16, 3, device=dev, generator=gen))
    for f_, b in zip(feats, (88, 88, 80)):
        f_.begin(torch.full((b,), 4, dtype=torch.int64, device=dev), torch.as_tensor(synth_keypoints(b), device=dev),
                 torch.randn(b, 16, 51, device=dev, generator=gen))
def sync():
    for s_ in [pos] + feats:
        s_.stream.synchronize()
    torch.cuda.synchronize(dev)
for trial in range(2):
    reset(); sync()
    if len(sys.argv) > 2:
        time.sleep(float(sys.argv[2]) * 1e-3)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(N // 10 + 1)]
    ev[0].record(feats[0].stream)
    for i in range(N // 10):
        joint.advance(10)
        ev[i + 1].record(feats[0].stream)
    sync()
    ms = [ev[i].elapsed_time(ev[i + 1]) / 10 for i in range(N // 10)]
    print("trial %d: ms per joint step, per block of 10 steps (feature chain 0's clock):" % trial)
    print(" ".join("%.3f" % v for v in ms))
    print("  first 20 steps %.3f | steps 20-100 %.3f | 100-300 %.3f | 300-end %.3f" % (
        np.mean(ms[:2]), np.mean(ms[2:10]), np.mean(ms[10:30]), np.mean(ms[30:])))
