"""Training losses of the two latent DDPMs, restated from the reference (forward = slide_amd.train.denoiser.TrainableDenoiser or any
callable net(x_t, ts, label) -> eps prediction).  Random timesteps and noise are drawn here unless they are passed in (the parity
tests inject the reference's)."""
import numpy as np
import torch

from ..diffusion import calc_diffusion_hyperparams, get_beta_schedule

_TABLES = {}  # (schedule, device) -> alpha_bar table on the device (an upload per step would be a blocking host -> device copy)


def _table(key, device, make):
    key = (key, str(device))
    t = _TABLES.get(key)
    if t is None:
        t = _TABLES[key] = torch.as_tensor(np.asarray(make(), dtype=np.float64), device=device).float()
    return t


def position_training_loss(net, X, diffusion_config, label, steps=None, z=None):
    """util.training_loss (pointnet2/util.py:262-300) with nn.MSELoss: x_t = sqrt(abar_t) x_0 + sqrt(1 - abar_t) z,
    loss = mean((eps_theta(x_t, t, label) - z)^2) over every element of the batch.  X (B, N, 3)."""
    T = int(diffusion_config["T"])
    abar = _table(("ddpm",) + tuple(sorted(diffusion_config.items())), X.device,
                  lambda: np.asarray(calc_diffusion_hyperparams(**diffusion_config)["Alpha_bar"]))
    B = X.shape[0]
    if steps is None:
        steps = torch.randint(T, size=(B,), device=X.device)
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
    sched = (cfg["beta_schedule"], cfg["beta_start"], cfg["beta_end"], cfg["num_diffusion_timesteps"])
    ac = _table(sched, x.device, lambda: np.cumprod(1.0 - get_beta_schedule(*sched), axis=0))  # extract(): float64 table cast to float32
    conditional = cfg.get("keypoint_conditional", False)
    w_kp = 0.0 if conditional else cfg.get("keypoint_position_loss_weight", 1.0)
    w_f = cfg.get("feature_loss_weight", 1.0)
    B = x.shape[0]
    kd = keypoint.shape[2]
    if steps is None:
        steps = torch.randint(int(cfg["num_diffusion_timesteps"]), size=(B,), device=x.device)
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
