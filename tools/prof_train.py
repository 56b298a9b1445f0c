import os, sys
sys.path.insert(0, "/root/repo")
import torch
from torch.profiler import profile, ProfilerActivity
from slide_amd import configs, model_spec
from slide_amd.synth import synth_keypoints, synth_state_dict
from slide_amd.train.denoiser import TrainableDenoiser
from slide_amd.train.dp import training_step
from slide_amd.train.losses import position_training_loss
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
NAME = sys.argv[2] if len(sys.argv) > 2 else "pos"
cfg = configs.position_ddpm_config() if NAME == "pos" else configs.feature_ddpm_config(); hp = cfg["pointnet_config"]
net = TrainableDenoiser(hp, synth_state_dict(model_spec.denoiser_param_spec(hp))).to(dev)
opt = torch.optim.Adam(net.parameters(), lr=2e-4)
kp = torch.as_tensor(synth_keypoints(B), device=dev); lab = torch.zeros(B, dtype=torch.int64, device=dev)
from slide_amd.train.losses import latent_training_loss
x0 = torch.cat([kp, 0.5 * torch.randn(B, 16, hp["in_fea_dim"], device=dev)], dim=2) if NAME == "feat" else kp
fn = (lambda: position_training_loss(net, kp, cfg["diffusion_config"], lab)) if NAME == "pos" else (lambda: latent_training_loss(net, x0, kp, lab, cfg["standard_diffusion_config"]).mean())
b = None
for _ in range(5):
    _, b = training_step(net, opt, fn, b)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(5):
        _, b = training_step(net, opt, fn, b)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=30, max_name_column_width=60))
