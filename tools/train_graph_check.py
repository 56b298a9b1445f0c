"""Gradients of the graphed training step (slide_amd/train/graph.py) against the eager backward on a FIXED batch, replay after replay,
with no host synchronisation between the replays: the largest deviation of any parameter's gradient over all replays.
(How the garbage bias gradients of torch's multi-block reductions inside replayed HIP graphs were found -- functions.col_sums.)
usage: python tools/train_graph_check.py [batch=256] [pos|feat] [replays=100]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slide_amd import configs, model_spec
from slide_amd.synth import synth_keypoints, synth_state_dict
from slide_amd.train.denoiser import TrainableDenoiser
from slide_amd.train.graph import GraphedTrainingStep
from slide_amd.train.losses import latent_training_loss, position_training_loss


def check(B, name, reps, dev):
    cfg = configs.position_ddpm_config() if name == "pos" else configs.feature_ddpm_config()
    hp = cfg["pointnet_config"]
    kp = torch.as_tensor(synth_keypoints(B), device=dev)
    x0 = torch.cat([kp, 0.5 * torch.randn(B, 16, hp["in_fea_dim"], device=dev)], dim=2) if name == "feat" else kp
    lab = torch.zeros(B, dtype=torch.int64, device=dev)
    steps, z = torch.randint(1000, (B,), device=dev), torch.randn_like(x0)
    net = TrainableDenoiser(hp, synth_state_dict(model_spec.denoiser_param_spec(hp))).to(dev)
    opt = torch.optim.SGD(net.parameters(), lr=0.0)  # the weights stay put: every replay must reproduce the same gradients
    if name == "pos":
        fn = lambda: position_training_loss(net, x0, cfg["diffusion_config"], lab, steps=steps, z=z)
    else:
        fn = lambda: latent_training_loss(net, x0, kp, lab, cfg["standard_diffusion_config"], steps=steps, z=z).mean()
    opt.zero_grad(set_to_none=True)
    fn().backward()
    ref = {k: p.grad.clone() for k, p in net.named_parameters()}
    scale = max(float(g.abs().max()) for g in ref.values())
    step = GraphedTrainingStep(net, opt, fn)
    names = [k for k, _ in net.named_parameters()]
    worst = torch.zeros(len(names), device=dev)
    for _ in range(reps):
        step()
        cur = torch.stack([(p.grad - ref[k]).abs().max() for k, p in net.named_parameters()])
        worst = torch.maximum(worst, torch.nan_to_num(cur, nan=1e30, posinf=1e30))
    torch.cuda.synchronize()
    return scale, dict(zip(names, worst.tolist()))


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    name = sys.argv[2] if len(sys.argv) > 2 else "pos"
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    scale, worst = check(B, name, reps, torch.device("cuda:0"))
    print("%s denoiser, batch %d, %d replays: gradient scale %.3g; largest deviations from the eager gradients:" % (name, B, reps, scale))
    for k in sorted(worst, key=lambda k: -worst[k])[:8]:
        print("  %-60s %.3g" % (k, worst[k]))
