"""per-group wall time of PipelinedGenerator.run (diagnostic)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from slide_amd import configs, model_spec, generation as G
from slide_amd.synth import synth_state_dict
from slide_amd import diffusion as D
dev = torch.device("cuda", 0)
pc, fc = configs.position_ddpm_config(), configs.feature_ddpm_config()
sd_p = synth_state_dict(model_spec.denoiser_param_spec(pc["pointnet_config"]))
sd_f = synth_state_dict(model_spec.denoiser_param_spec(fc["pointnet_config"]))
orig = D.EagerChainsSampler.advance
def timed(self, n):
    for s_ in self.samplers: s_.stream.synchronize()
    t0 = time.perf_counter(); orig(self, n)
    for s_ in self.samplers: s_.stream.synchronize()
    dt = time.perf_counter() - t0
    print("group of %d chains (every %s, batches %s): %.3f s = %.3f ms per round" % (len(self.samplers), list(self._every) if self._every is not None else None, [s_.B for s_ in self.samplers], dt, dt * 1e3 / n), flush=True)
D.EagerChainsSampler.advance = timed
g = G.PipelinedGenerator(256, dev, pos=(pc["pointnet_config"], sd_p, pc["diffusion_config"]), feat=(fc["pointnet_config"], sd_f, fc["standard_diffusion_config"]), prec="mixed", seed=1)
print("pos_mult", g.pos_mult)
t0 = time.perf_counter()
out = g.run(1536, np.zeros(1536, np.int64))
torch.cuda.synchronize()
print("total %.2f s, %s finite %s" % (time.perf_counter() - t0, tuple(out.shape), bool(torch.isfinite(out).all())))
