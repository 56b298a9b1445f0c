"""Training losses of the two latent DDPMs, restated from the reference (forward = slide_amd.train.denoiser.TrainableDenoiser or any
callable net(x_t, ts, label) -> eps prediction).  Random timesteps and noise are drawn here unless they are passed in (the parity
tests inject the reference's)."""
import numpy as np
import torch

from ..diffusion import calc_diffusion_hyperparams, get_beta_schedule


def position_training_loss(net, X, diffusion_config, label, steps=None, z=None):
    """util.training_loss (pointnet2/util.py:262-300) with nn.MSELoss: x_t = sqrt(abar_t) x_0 + sqrt(1 - abar_t) z,
    loss = mean((eps_theta(x_t, t, label) - z)^2) over every element of the batch.  X (B, N, 3)."""
    dh = calc_diffusion_hyperparams(**diffusion_config)
    abar = torch.as_tensor(dh["Alpha_bar"], device=X.device, dtype=torch.float32)
    B = X.shape[0]
    if steps is None:
        steps = torch.randint(dh["T"], size=(B,), device=X.device)
    if z is None:
        z = torch.randn_like(X)
    a = abar[steps.long()].reshape(B, 1, 1)
    x_t = torch.sqrt(a) * X + torch.sqrt(1 - a) * z
    eps = net(x_t, steps.reshape(B).float(), label)
    return torch.nn.functional.mse_loss(eps, z)


def latent_training_loss(net, x, keypoint, label, standard_diffusion_config, steps=None, z=None):
    """LatentDiffusion.train_loss (pointnet2/diffusion_utils/diffusion.py:319-341) on given latents x (B, N, kd + F) = [key points |
    features] (the reference gets them from its frozen autoencoder's encode()): with keypoint_conditional the key points are not
    diffused and their loss weight is 0; per-sample loss = mean over points of
    w_kp * sum_{c < kd} (eps - z)^2 + w_feat * mean_{c >= kd} (eps - z)^2.  -> (B,)"""
    cfg = standard_diffusion_config
    betas = get_beta_schedule(cfg["beta_schedule"], cfg["beta_start"], cfg["beta_end"], cfg["num_diffusion_timesteps"])
    ac = torch.as_tensor(np.cumprod(1.0 - betas, axis=0), device=x.device).float()  # extract(): float64 table cast to float32
    conditional = cfg.get("keypoint_conditional", False)
    w_kp = 0.0 if conditional else cfg.get("keypoint_position_loss_weight", 1.0)
    w_f = cfg.get("feature_loss_weight", 1.0)
    B = x.shape[0]
    kd = keypoint.shape[2]
    if steps is None:
        steps = torch.randint(int(betas.shape[0]), size=(B,), device=x.device)
    if z is None:
        z = torch.randn_like(x)
    a = ac[steps.long()].reshape(B, 1, 1)
    x_t = torch.sqrt(a) * x + torch.sqrt(1 - a) * z
    if conditional:
        x_t = torch.cat([keypoint, x_t[:, :, kd:]], dim=2)
    out = net(x_t, steps.float(), label) * cfg.get("model_output_scale_factor", 1.0)
    mse = (out - z) ** 2
    loss = w_kp * mse[:, :, :kd].sum(dim=2) + w_f * mse[:, :, kd:].mean(dim=2)
    return loss.mean(dim=1)
