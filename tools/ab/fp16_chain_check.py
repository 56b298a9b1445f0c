import sys, json, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from conftest import load_golden
import test_hip_engine as T
from slide_amd.diffusion import PositionSampler
dev = torch.device("cuda:0")
g = load_golden("golden_sampler_pos.npz")
_, hp, sd = T._load("pos")
ns = T.NoiseStream(g["full_seed"]); size = g["full_x0"].shape
xT = ns(size); noise = np.stack([ns(size) for _ in range(999)])
for prec in ("fp32", "fp16"):
    smp = PositionSampler(hp, sd, size[0], dev, T._pos_cfg(), prec=prec, noise=noise, use_graph=True)
    x0 = smp.sample(g["label"], xT).cpu().numpy()
    d = np.abs(x0 - g["full_x0"]).reshape(size[0], -1).max(1) / np.abs(g["full_x0"]).max()
    print(prec, "batch", size[0], "rel max err vs reference: %.3e" % T._rel(x0, g["full_x0"]), "per-sample:", np.round(d, 4))
