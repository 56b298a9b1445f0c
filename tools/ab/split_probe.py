"""split mode (fp32 storage, 3 x fp16 MFMA per product) against the exact-fp32 mode: forward error, per-op and chain timing"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from slide_amd import configs, model_spec
from slide_amd.engine import DenoiserEngine
from slide_amd.diffusion import FeatureSampler, PositionSampler
from slide_amd.synth import synth_keypoints, synth_state_dict
dev = torch.device("cuda:0")
rs = np.random.RandomState(0)
for nm in ("pos", "feat"):
    cfg = configs.position_ddpm_config() if nm == "pos" else configs.feature_ddpm_config()
    hp = cfg["pointnet_config"]
    sd = synth_state_dict(model_spec.denoiser_param_spec(hp))
    B = 64
    for fam in ("normal", "keypoints", "scaled0.1"):
        x = rs.standard_normal((B, 16, 3 + hp["in_fea_dim"])).astype(np.float32)
        if fam == "scaled0.1":
            x *= 0.1
        if nm == "feat" or fam == "keypoints":
            x[:, :, :3] = synth_keypoints(B, seed=99)
        ts = np.linspace(0, 999, B).astype(np.float32)
        lab = np.full(B, 4 if nm == "feat" else 0, np.int64)
        y = {p: DenoiserEngine(hp, sd, B, dev, prec=p).forward(x, ts, lab).double().cpu().numpy() for p in ("fp32", "split", "fp16")}
        rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
        print("[%s] %-10s split vs fp32 %.2e (max-norm %.2e)   fp16 vs fp32 %.2e" % (
            nm, fam, rel(y["split"], y["fp32"]), np.abs(y["split"] - y["fp32"]).max() / np.abs(y["fp32"]).max(), rel(y["fp16"], y["fp32"])), flush=True)
    Bc = 256
    for prec in ("fp32", "split", "fp16"):
        if nm == "pos":
            s = PositionSampler(hp, sd, Bc, dev, cfg["diffusion_config"], prec=prec, seed=1, use_graph=False)
            beg = lambda: s.begin(np.zeros(Bc, np.int64), rs.standard_normal((Bc, 16, 3)).astype(np.float32))
        else:
            s = FeatureSampler(hp, sd, Bc, dev, cfg["standard_diffusion_config"], prec=prec, seed=1, use_graph=False)
            beg = lambda: s.begin(np.full(Bc, 4, np.int64), synth_keypoints(Bc), rs.standard_normal((Bc, 16, 51)).astype(np.float32))
        beg(); s.advance(20); s.stream.synchronize()
        t0 = time.perf_counter(); s.advance(100); s.stream.synchronize(); dt = (time.perf_counter() - t0) / 100
        print("[%s] chain alone, batch %d, %s: %.1f us/step (%d launches)" % (nm, Bc, prec, dt * 1e6, len(s.step_ops)), flush=True)
