"""Training step (forward + loss + backward + Adam) of the two latent-DDPM denoisers on the HIP training path: ms per step and
samples/s at the reference's training batch size (32, config_*_batchsize_32_*).  usage: python tools/time_train.py [batch=32] [--eager-only]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slide_amd import configs, model_spec
from slide_amd.synth import synth_keypoints, synth_state_dict
from slide_amd.train.denoiser import TrainableDenoiser
from slide_amd.train.dp import training_step
from slide_amd.train.graph import GraphedTrainingStep
from slide_amd.train.losses import latent_training_loss, position_training_loss
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 32
EAGER_ONLY = "--eager-only" in sys.argv  # (for rocprofv3 --kernel-trace: per-kernel times of the eager step)
for name in ("pos", "feat"):
    cfg = configs.position_ddpm_config() if name == "pos" else configs.feature_ddpm_config()
    hp = cfg["pointnet_config"]
    net = TrainableDenoiser(hp, synth_state_dict(model_spec.denoiser_param_spec(hp))).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=2e-4, capturable=True)
    kp = torch.as_tensor(synth_keypoints(B), device=dev)
    x0 = torch.cat([kp, 0.5 * torch.randn(B, 16, hp["in_fea_dim"], device=dev)], dim=2) if name == "feat" else kp
    lab = torch.zeros(B, dtype=torch.int64, device=dev)
    if name == "pos":
        fn = lambda: position_training_loss(net, x0, cfg["diffusion_config"], lab)
    else:
        fn = lambda: latent_training_loss(net, x0, kp, lab, cfg["standard_diffusion_config"]).mean()
    bucket = None
    for _ in range(3):
        l_, bucket = training_step(net, opt, fn, bucket)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        l_, bucket = training_step(net, opt, fn, bucket)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    with torch.no_grad():
        net(x0, torch.zeros(B, device=dev), lab); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            net(x0, torch.zeros(B, device=dev), lab)
        torch.cuda.synchronize()
        df = (time.perf_counter() - t0) / n
    if EAGER_ONLY:
        print("%s denoiser, batch %d: training step eager %.2f ms (%.0f samples/s), forward only %.2f ms" % (name, B, dt * 1e3, B / dt, df * 1e3), flush=True)
        continue
    step = GraphedTrainingStep(net, opt, fn)
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        lg = step()
    torch.cuda.synchronize()
    dg = (time.perf_counter() - t0) / n
    print("%s denoiser, batch %d: training step eager %.2f ms (%.0f samples/s), as one HIP graph %.2f ms (%.0f samples/s), forward only %.2f ms; loss %.4f / %.4f"
          % (name, B, dt * 1e3, B / dt, dg * 1e3, B / dg, df * 1e3, float(l_), float(lg)), flush=True)
