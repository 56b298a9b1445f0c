"""tools/gen_golden.py -- AUTHORING-CONTAINER ONLY.  Generates tests/golden/*.npz.

Runs the REFERENCE's own Python (imported from /root/reference via tools/ref_shims.py, never
copied) on seeded inputs and records inputs + outputs as small fixtures:
  golden_denoiser_{pos,feat}.npz  PointNet2CloudCondition.forward at t in {0,1,500,999}, per-level
                                  features (forward hooks), state-dict spec (names + shapes)
  golden_sampler_{pos,feat}.npz   util.sampling / LatentDiffusion.denoise_and_reconstruct segments with
                                  the noise stream injected (seeded numpy RandomState), schedule tables
  golden_denoiser_variants.npz    the same forward with the FP-module / layer-order branches switched (use_knn_FP False, bn_first, bn False)
  golden_denoiser_condition.npz   the same forward WITH a condition cloud (local feature transfer, global feature, both; retained features)
  golden_blocks.npz               stand-alone reference modules (QueryAndGroup, group_knn, Mlp_plus_t_emb,
                                  AttentionModule, PointnetSAModule w/ FPS, PointnetFPModule (three_nn path))
  golden_ops.npz                  op-level adversarial cases (computed by the C oracle -- the reference has
                                  no CPU path for `_ext`; the oracle itself is pinned by the files above)
Weights are NOT stored: slide_amd.synth.synth_state_dict(spec) regenerates them from the names.

usage:  python tools/gen_golden.py [--out tests/golden]
"""
import argparse
import copy
import io
import contextlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
from tools import ref_shims  # noqa: E402

ref_shims.install()
from slide_amd.synth import synth_state_dict, synth_keypoints  # noqa: E402

CFG = "/root/reference/pointnet2/configs/shapenet_psr_configs/"
POS_CFG = CFG + "ddpm_keypoint_training_configs/config_standard_attention_batchsize_32_s3_ema_model_keypoint_airplane_02691156.json"
FEAT_CFG = CFG + "latent_ddpm_training_configs/config_latent_ddpm_s3_dim_16_32_ae_kp_noise_0.04_keypoint_conditional_chair_ae_trained_on_chair.json"


def load_cfg(path):
    from data_utils.json_reader import restore_string_to_list_in_a_dict
    with open(path) as f:
        return restore_string_to_list_in_a_dict(json.loads(f.read()))


def build_net(cfg):
    from models.pointnet2_with_pcld_condition import PointNet2CloudCondition
    net = PointNet2CloudCondition(copy.deepcopy(cfg["pointnet_config"]))
    spec = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    sd = synth_state_dict(spec)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval()
    return net, spec


def spec_arrays(spec):
    names = np.array([n for n, _ in spec])
    shapes = np.array([",".join(str(d) for d in s) for _, s in spec])
    return names, shapes


class NoiseStream:
    """seeded standard-normal stream shared (by construction) with the tests"""

    def __init__(self, seed):
        self.rs = np.random.RandomState(seed)
        self.count = 0

    def __call__(self, size):
        self.count += 1
        return torch.from_numpy(self.rs.standard_normal(tuple(size)).astype(np.float32))


def gen_denoiser(name, cfg_path, out, B=3):
    cfg = load_cfg(cfg_path)
    net, spec = build_net(cfg)
    hp = cfg["pointnet_config"]
    C = 3 + hp["in_fea_dim"]
    rs = np.random.RandomState(7)
    res = {}
    names, shapes = spec_arrays(spec)
    res["spec_names"], res["spec_shapes"] = names, shapes
    res["config_json"] = np.array(json.dumps(hp))
    inter = {}

    def hook(tag):
        def fn(mod, inp, outp):
            inter[tag] = (outp[1] if isinstance(outp, tuple) else outp).detach().numpy().copy()
        return fn

    for i, m in enumerate(net.SA_modules):
        m.register_forward_hook(hook("sa%d" % i))
    for i, m in enumerate(net.FP_modules):
        m.register_forward_hook(hook("fp%d" % i))
    for t in [0, 1, 500, 999]:
        x = rs.standard_normal((B, 16, C)).astype(np.float32)
        if name == "feat":
            x[:, :, 0:3] = synth_keypoints(B, 16, seed=t)
        elif t < 500:
            x[:, :, 0:3] *= 0.5
        label = rs.randint(0, 13, size=(B,)).astype(np.int64)
        ts = np.full((B,), t, np.float32)
        with torch.no_grad():
            y = net(torch.from_numpy(x), ts=torch.from_numpy(ts), label=torch.from_numpy(label))
        res["x_t%d" % t] = x; res["label_t%d" % t] = label; res["ts_t%d" % t] = ts
        res["eps_t%d" % t] = y.numpy()
        for k, v in inter.items():
            res["%s_t%d" % (k, t)] = v
    # mixed per-sample timesteps in one batch
    x = rs.standard_normal((B, 16, C)).astype(np.float32)
    ts = np.array([3, 250, 998][:B], np.float32)
    label = np.array([0, 4, 12][:B], np.int64)
    with torch.no_grad():
        y = net(torch.from_numpy(x), ts=torch.from_numpy(ts), label=torch.from_numpy(label))
    res["x_mixed"], res["ts_mixed"], res["label_mixed"], res["eps_mixed"] = x, ts, label, y.numpy()
    np.savez_compressed(os.path.join(out, "golden_denoiser_%s.npz" % name), **res)
    print("denoiser", name, "params", sum(int(np.prod(s)) for _, s in spec), "tensors", len(spec))
    return net, cfg


DENOISER_VARIANTS = {  # configuration branches no shipped latent-DDPM config takes (reference pointnet2_with_pcld_condition.py:226-241, :259-277)
    "fp3nn": dict(use_knn_FP=False),                                 # PointnetFPModule: three_nn / three_interpolate + one Mlp
    "bnfirst": dict(bn_first=True),                                  # GroupNorm -> ReLU -> conv order, activation + conv output head
    "fp3nn_bnfirst": dict(use_knn_FP=False, bn_first=True),
    "nobn": dict(bn=False),                                          # no GroupNorm in the Mlps; conv -> ReLU -> conv head
}


def variant_config(base_hp, name):
    hp = copy.deepcopy(base_hp)
    v = DENOISER_VARIANTS[name]
    if "use_knn_FP" in v:
        hp["architecture"]["use_knn_FP"] = v["use_knn_FP"]
    for k in ("bn_first", "bn"):
        if k in v:
            hp[k] = v[k]
    if name == "nobn":  # (the attention modules carry their own switch)
        hp["attention_setting"] = dict(hp["attention_setting"], attention_bn=False)
    return hp


def gen_denoiser_variants(out, B=2):
    """PointNet2CloudCondition.forward of the position net's configuration with the FP-module / layer-order branches switched"""
    base = load_cfg(POS_CFG)
    res = {}
    rs = np.random.RandomState(11)
    for name in DENOISER_VARIANTS:
        cfg = copy.deepcopy(base)
        cfg["pointnet_config"] = variant_config(base["pointnet_config"], name)
        net, spec = build_net(cfg)
        names, shapes = spec_arrays(spec)
        res[name + "_spec_names"], res[name + "_spec_shapes"] = names, shapes
        res[name + "_config_json"] = np.array(json.dumps(cfg["pointnet_config"]))
        x = rs.standard_normal((B, 16, 3)).astype(np.float32)
        x[1] *= 0.4
        ts = np.array([999, 7][:B], np.float32)
        label = np.array([0, 4][:B], np.int64)
        with torch.no_grad():
            y = net(torch.from_numpy(x), ts=torch.from_numpy(ts), label=torch.from_numpy(label))
        res[name + "_x"], res[name + "_ts"], res[name + "_label"], res[name + "_eps"] = x, ts, label, y.numpy()
        print("variant", name, "params", sum(int(np.prod(s_)) for _, s_ in spec), "eps rms %.3f" % float(np.sqrt((y.numpy() ** 2).mean())))
    np.savez_compressed(os.path.join(out, "golden_denoiser_variants.npz"), **res)


def condition_config(base_hp, local, glob):
    """the position net's configuration with a CONDITION CLOUD (the two-stream form of pointnet2_with_pcld_condition.py:94-260; no
    shipped configuration sets it -- a small architecture of this repo's choosing, the reference's keys)"""
    hp = copy.deepcopy(base_hp)
    hp["include_local_feature"], hp["include_global_feature"] = local, glob
    hp["pnet_global_feature_architecture"] = [[3, 32, 64], [64, 96]]
    hp["condition_net_architecture"] = {"npoint": [32, 16], "radius": [0, 0], "neighbor_definition": "nn", "nsample": [8, 8],
                                        "feature_dim": [16, 32, 64], "mlp_depth": 3, "decoder_feature_dim": [16, 32, 64],
                                        "decoder_mlp_depth": 2, "use_knn_FP": True, "K": 3}
    hp["feature_mapper_architecture"] = {"neighbor_definition": "nn", "encoder_feature_map_dim": [16, 32], "encoder_mlp_depth": 2,
                                         "encoder_radius": [0, 0], "encoder_nsample": [8, 8],
                                         "decoder_feature_map_dim": [16, 32, 64], "decoder_mlp_depth": 2,
                                         "decoder_radius": [0, 0, 0], "decoder_nsample": [8, 8, 8]}
    return hp


def gen_denoiser_condition(out, B=2, M=64):
    """PointNet2CloudCondition.forward WITH a condition cloud: local features (two-stream encoder / decoder with feature transfer
    modules), the global feature (Pnet2Stage), both; and the retained-condition-feature path (second call re-uses the first's)"""
    base = load_cfg(POS_CFG)
    res = {}
    rs = np.random.RandomState(13)
    for name, (local, glob) in {"local": (True, False), "global": (False, True), "both": (True, True)}.items():
        cfg = copy.deepcopy(base)
        cfg["pointnet_config"] = condition_config(base["pointnet_config"], local, glob)
        net, spec = build_net(cfg)
        names, shapes = spec_arrays(spec)
        res[name + "_spec_names"], res[name + "_spec_shapes"] = names, shapes
        res[name + "_config_json"] = np.array(json.dumps(cfg["pointnet_config"]))
        x = rs.standard_normal((B, 16, 3)).astype(np.float32)
        cond = (0.6 * rs.standard_normal((B, M, 3))).astype(np.float32)
        ts = np.array([999, 7][:B], np.float32)
        label = np.array([0, 4][:B], np.int64)
        with torch.no_grad():
            y = net(torch.from_numpy(x), condition=torch.from_numpy(cond), ts=torch.from_numpy(ts), label=torch.from_numpy(label))
            # retained condition features: the first call stores them, the second (other x, other t) re-uses them
            net.reset_cond_features()
            y1 = net(torch.from_numpy(x), condition=torch.from_numpy(cond), ts=torch.from_numpy(ts), label=torch.from_numpy(label),
                     use_retained_condition_feature=True)
            x2 = rs.standard_normal((B, 16, 3)).astype(np.float32)
            ts2 = np.array([500, 3][:B], np.float32)
            y2 = net(torch.from_numpy(x2), condition=torch.from_numpy(cond), ts=torch.from_numpy(ts2), label=torch.from_numpy(label),
                     use_retained_condition_feature=True)
        assert np.allclose(y.numpy(), y1.numpy(), atol=1e-6)
        for k, v in (("x", x), ("cond", cond), ("ts", ts), ("label", label), ("eps", y.numpy()), ("x2", x2), ("ts2", ts2), ("eps2", y2.numpy())):
            res[name + "_" + k] = v
        print("condition", name, "params", sum(int(np.prod(s_)) for _, s_ in spec), "eps rms %.3f %.3f" % (
            float(np.sqrt((y.numpy() ** 2).mean())), float(np.sqrt((y2.numpy() ** 2).mean()))))
    np.savez_compressed(os.path.join(out, "golden_denoiser_condition.npz"), **res)


DENOISER_SWITCHES = {  # the completion / refinement parent project's switches of the same class (reference :47-93, :119-126, :243-257, :302-347)
    "swish_pe_ga": dict(activation="swish", use_position_encoding=True, position_encoding_multires=4,
                        global_attention_setting={"use_global_attention_module": True, "global_attention_layer_index": [0, 1],
                                                  "attention_bn": True, "last_activation": True},
                        point_upsample_factor=2, first_refine_coarse_points=False, include_displacement_center_to_final_output=False),
    "concat_partial": dict(concate_partial_with_noisy_input=True, in_fea_dim=1, attach_position_to_input_feature=False),
}


def gen_denoiser_switches(out, B=2, M=24):
    base = load_cfg(POS_CFG)
    res = {}
    rs = np.random.RandomState(17)
    for name, sw in DENOISER_SWITCHES.items():
        cfg = copy.deepcopy(base)
        cfg["pointnet_config"].update(copy.deepcopy(sw))
        res[name + "_config_json"] = np.array(json.dumps(cfg["pointnet_config"]))  # (before the constructor edits it in place)
        net, spec = build_net(cfg)
        names, shapes = spec_arrays(spec)
        res[name + "_spec_names"], res[name + "_spec_shapes"] = names, shapes
        x = rs.standard_normal((B, 16, 3)).astype(np.float32)
        ts = np.array([999, 7][:B], np.float32)
        label = np.array([0, 4][:B], np.int64)
        kw = {}
        if name == "concat_partial":
            cond = (0.6 * rs.standard_normal((B, M, 3))).astype(np.float32)
            kw["condition"] = torch.from_numpy(cond)
            res[name + "_cond"] = cond
        with torch.no_grad():
            y = net(torch.from_numpy(x), ts=torch.from_numpy(ts), label=torch.from_numpy(label), **kw)
        res[name + "_x"], res[name + "_ts"], res[name + "_label"], res[name + "_eps"] = x, ts, label, y.numpy()
        print("switches", name, "params", sum(int(np.prod(s_)) for _, s_ in spec), "out", tuple(y.shape), "rms %.3f" % float(np.sqrt((y.numpy() ** 2).mean())))
    np.savez_compressed(os.path.join(out, "golden_denoiser_switches.npz"), **res)


def gen_sampler_pos(net, cfg, out, B=2):
    import util
    res = {}
    dh = util.calc_diffusion_hyperparams(**cfg["diffusion_config"])
    for k in ["Beta", "Alpha", "Alpha_bar", "Sigma"]:
        res["sched_" + k] = dh[k].numpy()
    label = torch.tensor([0, 4][:B]).long()
    res["label"] = label.numpy()

    def run(seed, **kw):
        ns = NoiseStream(seed)
        old = util.std_normal
        util.std_normal = ns
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                x = util.sampling(net, (B, 16, 3), dh, label=label, verbose=False, **kw)
        finally:
            util.std_normal = old
        return x.numpy(), ns.count

    # (a) the last 20 reverse steps from a supplied X_T (reference: use_a_precomputed_XT, util.py:228-230)
    XT = (0.5 * np.random.RandomState(11).standard_normal((B, 16, 3))).astype(np.float32)
    res["tail_XT"] = XT; res["tail_step"] = np.array(20); res["tail_seed"] = np.array(101)
    res["tail_x0"], res["tail_ndraws"] = run(101, use_a_precomputed_XT=True, step=20, XT=torch.from_numpy(XT))
    # (b) the full 1000-step chain
    res["full_seed"] = np.array(202)
    res["full_x0"], res["full_ndraws"] = run(202)
    np.savez_compressed(os.path.join(out, "golden_sampler_pos.npz"), **res)
    print("sampler pos: tail draws", res["tail_ndraws"], "full draws", res["full_ndraws"])


def gen_sampler_feat(net, cfg, out, B=2):
    from diffusion_utils import diffusion as D
    res = {}
    dcfg = copy.deepcopy(cfg["standard_diffusion_config"])
    with contextlib.redirect_stdout(io.StringIO()):
        dm = D.LatentDiffusion(dcfg, autoencoder=None, device=torch.device("cpu"))
    dm.decode = lambda latent, keypoint_dim, label: latent  # skip the autoencoder; the loop is untouched
    for k in ["logvar", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_mean_coef1",
              "posterior_mean_coef2"]:
        res["sched_" + k] = np.asarray(getattr(dm, k), np.float64)
    res["config_json"] = np.array(json.dumps(dcfg))
    label = torch.tensor([4, 4][:B]).long()
    keypoint = torch.from_numpy(synth_keypoints(B, 16, seed=5))
    res["label"] = label.numpy(); res["keypoint"] = keypoint.numpy()

    def run(seed, **kw):
        ns = NoiseStream(seed)
        o1, o2 = torch.randn, torch.randn_like
        torch.randn = lambda *size, **k: ns(size)
        torch.randn_like = lambda x, **k: ns(x.shape)
        try:
            with torch.no_grad():
                _, kp, feat = dm.denoise_and_reconstruct(B, net, 3, (16, 51), label=label, keypoint=keypoint,
                                                         return_keypoint_feature=True, **kw)
        finally:
            torch.randn, torch.randn_like = o1, o2
        return torch.cat([kp, feat], dim=2).numpy(), ns.count

    # (a) first 20 steps t=999..980 from x_T ~ N(0,1)
    res["head_seed"] = np.array(303); res["head_nsteps"] = np.array(20)
    res["head_x"], res["head_ndraws"] = run(303, n_steps=20)
    # (b) last 20 steps t=19..0 from a supplied x_20
    x20 = (0.7 * np.random.RandomState(12).standard_normal((B, 16, 51))).astype(np.float32)
    res["tail_x_in"] = x20; res["tail_seed"] = np.array(404); res["tail_curr_step"] = np.array(20)
    res["tail_x0"], res["tail_ndraws"] = run(404, x=torch.from_numpy(x20), curr_step=20)
    np.savez_compressed(os.path.join(out, "golden_sampler_feat.npz"), **res)
    print("sampler feat: head draws", res["head_ndraws"], "tail draws", res["tail_ndraws"])


def gen_sampler_feat_full(net, cfg, out, B=2):
    """COMPLETE 1000-step LatentDiffusion.denoise_and_reconstruct chain (diffusion.py:346-404) with the noise stream injected:
    pins the fp32 AND the fp16 mode of the HIP feature sampler over a whole generation"""
    from diffusion_utils import diffusion as D
    res = {}
    dcfg = copy.deepcopy(cfg["standard_diffusion_config"])
    with contextlib.redirect_stdout(io.StringIO()):
        dm = D.LatentDiffusion(dcfg, autoencoder=None, device=torch.device("cpu"))
    dm.decode = lambda latent, keypoint_dim, label: latent
    res["config_json"] = np.array(json.dumps(dcfg))
    label = torch.tensor([4, 4][:B]).long()
    keypoint = torch.from_numpy(synth_keypoints(B, 16, seed=6))
    res["label"] = label.numpy(); res["keypoint"] = keypoint.numpy()
    ns = NoiseStream(505)
    o1, o2 = torch.randn, torch.randn_like
    torch.randn = lambda *size, **k: ns(size)
    torch.randn_like = lambda x, **k: ns(x.shape)
    try:
        with torch.no_grad():
            _, kp, feat = dm.denoise_and_reconstruct(B, net, 3, (16, 51), label=label, keypoint=keypoint, return_keypoint_feature=True)
    finally:
        torch.randn, torch.randn_like = o1, o2
    res["full_seed"] = np.array(505)
    res["full_x0"] = torch.cat([kp, feat], dim=2).numpy()
    res["full_ndraws"] = np.array(ns.count)
    np.savez_compressed(os.path.join(out, "golden_sampler_feat_full.npz"), **res)
    print("sampler feat full chain: draws", ns.count)


def gen_train(out, B=2):
    """TRAINING-STEP fixtures: the reference's own losses and autograd on CPU --
    util.training_loss (pointnet2/util.py:262-300, position DDPM, nn.MSELoss) and LatentDiffusion.train_loss
    (pointnet2/diffusion_utils/diffusion.py:319-341, feature DDPM on given latents: encode() is an identity here) -- with the
    random timesteps and the noise INJECTED; recorded: inputs, loss, the L2 norm of every parameter's gradient and the full
    gradients of a handful of parameters of every kind."""
    import util as U
    from diffusion_utils import diffusion as D
    keep = ("class_emb.weight", "fc_t1.weight", "fc_t2.bias", "SA_modules.0.mlps.0.first_mlp.0.weight",
            "SA_modules.0.mlps.0.first_mlp.1.group_norm.weight", "SA_modules.0.mlps.0.fc.weight", "SA_modules.1.mlps.0.res_connect.weight",
            "SA_modules.1.attention_modules.0.weight_conv.2.weight", "SA_modules.1.attention_modules.0.weight_conv.4.group_norm.bias",
            "SA_modules.0.attention_modules.0.feat_conv.weight", "FP_modules.1.mlp1.second_mlp.0.weight",
            "FP_modules.0.attention_module.grouped_feat_conv.weight", "FP_modules.0.mlp2.fc_condition.bias",
            "FP_modules.0.attention_module.feat_out_conv.1.group_norm.weight", "fc_lyaer.0.weight", "fc_lyaer.1.weight", "fc_lyaer.3.bias")
    for name, cfg_path in (("pos", POS_CFG), ("feat", FEAT_CFG)):
        cfg = load_cfg(cfg_path)
        net, spec = build_net(cfg)
        net.train()
        hp = cfg["pointnet_config"]
        C = 3 + hp["in_fea_dim"]
        rs = np.random.RandomState(31 if name == "pos" else 32)
        x0 = (0.6 * rs.standard_normal((B, 16, C))).astype(np.float32)
        x0[:, :, :3] = synth_keypoints(B, 16, seed=9)
        label = np.array([0, 4][:B], np.int64)
        steps = np.array([37, 812][:B], np.int64)
        z = rs.standard_normal((B, 16, C)).astype(np.float32)
        res = {"x0": x0, "label": label, "steps": steps, "z": z, "config_json": np.array(json.dumps(cfg))}
        names, shapes = spec_arrays(spec)
        res["spec_names"], res["spec_shapes"] = names, shapes
        o_randint, o_randn_like = torch.randint, torch.randn_like
        try:
            if name == "pos":
                dh = U.calc_diffusion_hyperparams(**cfg["diffusion_config"])
                torch.randint = lambda *a, **k: torch.from_numpy(steps).reshape(B, 1, 1)
                o_std = U.std_normal
                U.std_normal = lambda size: torch.from_numpy(z)
                try:
                    loss = U.training_loss(net, torch.nn.MSELoss(), torch.from_numpy(x0), dh, label=torch.from_numpy(label))
                finally:
                    U.std_normal = o_std
            else:
                dcfg = copy.deepcopy(cfg["standard_diffusion_config"])
                with contextlib.redirect_stdout(io.StringIO()):
                    dm = D.LatentDiffusion(dcfg, autoencoder=None, device=torch.device("cpu"))
                dm.encode = lambda x, keypoint, label: x  # the latents are given: [key points | features]
                torch.randint = lambda *a, **k: torch.from_numpy(steps)
                torch.randn_like = lambda t, **k: torch.from_numpy(z)
                res["diffusion_config_json"] = np.array(json.dumps(dcfg))
                loss_b = dm.train_loss(net, torch.from_numpy(x0), torch.from_numpy(x0[:, :, :3].copy()), torch.from_numpy(label))
                res["loss_per_sample"] = loss_b.detach().numpy()
                loss = loss_b.mean()
        finally:
            torch.randint, torch.randn_like = o_randint, o_randn_like
        loss.backward()
        res["loss"] = np.array(loss.item(), np.float64)
        grads = {k: p.grad.detach().numpy() for k, p in net.named_parameters()}
        assert list(grads) == [n for n, _ in spec] and all(g is not None for g in grads.values())
        res["grad_norms"] = np.array([np.linalg.norm(grads[n].astype(np.float64)) for n, _ in spec])
        for k in keep:  # (big tensors: every stride-th entry of the flattened gradient, ~8k values)
            g_ = grads[k].reshape(-1)
            stride = max(1, g_.size // 8192)
            res["grad__" + k] = g_[::stride].copy()
            res["stride__" + k] = np.array(stride)
        np.savez_compressed(os.path.join(out, "golden_train_%s.npz" % name), **res)
        print("train", name, "loss", float(loss), "grad norm", float(np.sqrt((res["grad_norms"] ** 2).sum())))


def gen_schedules(out):
    """the schedule tables of Diffusion.init_diffusion_parameters (diffusion.py:158-208) for every beta schedule the
    reference's get_beta_schedule can produce ('warmup10' / 'warmup50' raise NameError there: `_warmup_beta` is never
    defined) and both variance types, T = 1000 and a short T = 50"""
    from diffusion_utils import diffusion as D
    res, cfgs = {}, []
    base = copy.deepcopy(load_cfg(FEAT_CFG)["standard_diffusion_config"])
    for sched in ("linear", "quad", "const", "jsd"):
        for vt in ("fixedsmall", "fixedlarge"):
            for T in (1000, 50):
                c = copy.deepcopy(base)
                c.update(beta_schedule=sched, model_var_type=vt, num_diffusion_timesteps=T)
                with contextlib.redirect_stdout(io.StringIO()):
                    dm = D.LatentDiffusion(c, autoencoder=None, device=torch.device("cpu"))
                k = len(cfgs)
                cfgs.append(c)
                for nm in ["logvar", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_mean_coef1",
                           "posterior_mean_coef2"]:
                    res["c%d_%s" % (k, nm)] = np.asarray(getattr(dm, nm), np.float64)
    for sched in ("warmup10", "warmup50"):
        try:
            D.get_beta_schedule(sched, beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=10)
            raise SystemExit("the reference now defines _warmup_beta: record its tables too")
        except NameError:
            pass
    res["configs_json"] = np.array(json.dumps(cfgs))
    np.savez_compressed(os.path.join(out, "golden_schedules.npz"), **res)
    print("schedules:", len(cfgs), "configs")


def gen_sampler_feat_resample(out, B=2):
    """LatentDiffusion.denoise_and_reconstruct(local_resampling=True) (diffusion.py:346-359, :76-79): features are
    re-generated only on the points with keypoint_mask == 1, the predicted x0 of the others is pinned to complete_x0.
    Recorded: a 20-step head from x_T and the last 20 steps are not reachable with x given (the reference asserts
    x is None in this mode), so the fixture is a 12-step chain from x_T on a diffusion object with T shrunk to 12 for the
    tail behaviour (t = 11..0, includes the noise-free last step) plus a 20-step head at T = 1000."""
    from diffusion_utils import diffusion as D
    cfg = load_cfg(FEAT_CFG)
    net, _ = build_net(cfg)
    res = {}
    label = torch.tensor([4, 4][:B]).long()
    keypoint = torch.from_numpy(synth_keypoints(B, 16, seed=5))
    rs = np.random.RandomState(77)
    complete_x0 = np.concatenate([keypoint.numpy(), (0.5 * rs.standard_normal((B, 16, 48))).astype(np.float32)], axis=2)
    mask = (rs.uniform(size=(B, 16)) < 0.4).astype(np.float32)
    mask[0, 0], mask[0, 1] = 1.0, 0.0
    res["label"], res["keypoint"], res["complete_x0"], res["keypoint_mask"] = label.numpy(), keypoint.numpy(), complete_x0, mask
    for tag, T, nsteps, seed in (("head", 1000, 20, 505), ("short", 12, 12, 606)):
        dcfg = copy.deepcopy(cfg["standard_diffusion_config"])
        dcfg["num_diffusion_timesteps"] = T
        with contextlib.redirect_stdout(io.StringIO()):
            dm = D.LatentDiffusion(dcfg, autoencoder=None, device=torch.device("cpu"))
        dm.decode = lambda latent, keypoint_dim, label: latent
        ns = NoiseStream(seed)
        o1, o2 = torch.randn, torch.randn_like
        torch.randn = lambda *size, **k: ns(size)
        torch.randn_like = lambda x, **k: ns(x.shape)
        try:
            with torch.no_grad():
                _, kp, feat = dm.denoise_and_reconstruct(B, net, 3, (16, 51), label=label, keypoint=keypoint,
                                                         return_keypoint_feature=True, n_steps=nsteps, local_resampling=True,
                                                         complete_x0=torch.from_numpy(complete_x0),
                                                         keypoint_mask=torch.from_numpy(mask))
        finally:
            torch.randn, torch.randn_like = o1, o2
        res[tag + "_config_json"] = np.array(json.dumps(dcfg))
        res[tag + "_seed"], res[tag + "_nsteps"], res[tag + "_ndraws"] = np.array(seed), np.array(nsteps), np.array(ns.count)
        res[tag + "_x"] = torch.cat([kp, feat], dim=2).numpy()
    np.savez_compressed(os.path.join(out, "golden_sampler_feat_resample.npz"), **res)
    print("sampler feat resample: draws", res["head_ndraws"], res["short_ndraws"])


def gen_blocks(out):
    """stand-alone reference modules on seeded inputs, covering branches the DDPM configs skip"""
    from pointnet2_ops import pointnet2_utils as PU
    from pointnet2_ops import pointnet2_modules as PM
    from pointnet2_ops.attention import AttentionModule
    rs = np.random.RandomState(21)
    res = {}
    B, N, npoint, C = 2, 40, 12, 5
    xyz = rs.uniform(-1, 1, (B, N, 3)).astype(np.float32)
    feats = rs.standard_normal((B, C, N)).astype(np.float32)
    res["xyz"], res["feats"] = xyz, feats
    txyz, tf = torch.from_numpy(xyz), torch.from_numpy(feats)
    # FPS + gather + QueryAndGroup in both neighbour definitions
    fidx = PU.furthest_point_sample(txyz, npoint)
    new_xyz = PU.gather_operation(txyz.transpose(1, 2).contiguous(), fidx).transpose(1, 2).contiguous()
    res["fps_idx"] = fidx.numpy(); res["new_xyz"] = new_xyz.numpy()
    g_nn = PU.QueryAndGroup(0, 8, use_xyz=True, include_abs_coordinate=True, include_center_coordinate=True,
                            neighbor_def="nn")
    o, c = g_nn(txyz, new_xyz, tf, subset=True, return_counts=True)
    res["qg_nn"], res["qg_nn_counts"] = o.numpy(), c.numpy()
    g_r = PU.QueryAndGroup(0.6, 8, use_xyz=True, include_abs_coordinate=True, include_center_coordinate=False,
                           neighbor_def="radius")
    o, c = g_r(txyz, new_xyz, tf, subset=True, return_counts=True)
    res["qg_radius"], res["qg_radius_counts"] = o.numpy(), c.numpy()
    q2 = rs.uniform(-1.5, 1.5, (B, 9, 3)).astype(np.float32)
    res["q2"] = q2
    o, c = g_r(txyz, torch.from_numpy(q2), tf, subset=False, return_counts=True)
    res["qg_radius_nosubset"], res["qg_radius_nosubset_counts"] = o.numpy(), c.numpy()
    # group_knn
    res["group_knn"] = PU.group_knn(new_xyz, txyz, tf, 6, transpose=True).numpy()
    # three_nn / three_interpolate through PointnetFPModule
    fp = PM.PointnetFPModule(mlp=[C + 7, 16, 16], bn=True, include_t=False, bias=True, res_connect=True)
    spec = [(k, tuple(v.shape)) for k, v in fp.state_dict().items()]
    vals = synth_state_dict([("fpmod." + n, s) for n, s in spec])
    fp.load_state_dict({n: torch.from_numpy(vals["fpmod." + n]) for n, _ in spec})
    fp.eval()
    uf = rs.standard_normal((B, 7, N)).astype(np.float32)
    kf = rs.standard_normal((B, C, npoint)).astype(np.float32)
    res["fp_unknown_feats"], res["fp_known_feats"] = uf, kf
    res["fp_spec_names"], res["fp_spec_shapes"] = spec_arrays(spec)
    with torch.no_grad():
        res["fp_out"] = fp(txyz, new_xyz, torch.from_numpy(uf), torch.from_numpy(kf)).numpy()
    d, i = PU.three_nn(txyz, new_xyz)
    res["three_nn_dist"], res["three_nn_idx"] = d.numpy(), i.numpy()
    # PointnetSAModule with FPS (N > npoint), t-embedding + condition + attention
    att = {"use_attention_module": True, "attention_bn": True, "transform_grouped_feat_out": True,
           "last_activation": True}
    sa = PM.PointnetSAModule(mlp=[C, 16, 16, 32], npoint=npoint, radius=0, nsample=8, bn=True, use_xyz=True,
                             t_dim=24, include_t=True, include_abs_coordinate=True, include_center_coordinate=True,
                             bias=True, res_connect=True, include_condition=True, condition_dim=10,
                             neighbor_def="nn", attention_setting=att)
    spec = [(k, tuple(v.shape)) for k, v in sa.state_dict().items()]
    vals = synth_state_dict([("samod." + n, s) for n, s in spec])
    sa.load_state_dict({n: torch.from_numpy(vals["samod." + n]) for n, _ in spec})
    sa.eval()
    temb = rs.standard_normal((B, 24)).astype(np.float32); cemb = rs.standard_normal((B, 10)).astype(np.float32)
    res["sa_t_emb"], res["sa_cond_emb"] = temb, cemb
    res["sa_spec_names"], res["sa_spec_shapes"] = spec_arrays(spec)
    with torch.no_grad():
        nx, nf = sa(txyz, tf, t_emb=torch.from_numpy(temb), condition_emb=torch.from_numpy(cemb))
    res["sa_new_xyz"], res["sa_new_features"] = nx.numpy(), nf.numpy()
    # AttentionModule with a count mask (radius grouping path)
    am = AttentionModule(C, C + 6, C, C + 6, 32, attention_bn=True, transform_grouped_feat_out=True,
                         last_activation=True)
    spec = [(k, tuple(v.shape)) for k, v in am.state_dict().items()]
    vals = synth_state_dict([("attmod." + n, s) for n, s in spec])
    am.load_state_dict({n: torch.from_numpy(vals["attmod." + n]) for n, _ in spec})
    am.eval()
    gfo = rs.standard_normal((B, 32, npoint, 8)).astype(np.float32)
    res["att_grouped_feat_out"] = gfo
    res["att_spec_names"], res["att_spec_shapes"] = spec_arrays(spec)
    qfeat = PU.gather_operation(tf, fidx)
    res["att_query"] = qfeat.numpy()
    with torch.no_grad():
        res["att_out"] = am(qfeat, torch.from_numpy(res["qg_radius"]), torch.from_numpy(gfo),
                            torch.from_numpy(res["qg_radius_counts"])).numpy()
    np.savez_compressed(os.path.join(out, "golden_blocks.npz"), **res)
    print("blocks done")


def gen_decode(out, B=2):
    """PointAutoencoder.decode of the reference (airplane AE config) on synthetic latents; FPS start index 0 (shim)"""
    from data_utils.json_reader import read_json_file, autoencoder_read_config
    from models.autoencoder import PointAutoencoder
    d = CFG + "autoencoder_configs/"
    cfg = read_json_file(d + "config_autoencoder_s3_kl_1e-5_16_keypoints_latent_dim_16_32_normal_weight_0_0_0.1_with_augm_kp_noise_0.04_airplane.json")
    enc, decs = autoencoder_read_config(d, cfg)
    ae = PointAutoencoder(enc, decs, apply_kl_regularization=True, kl_weight=1e-5)
    spec_all = [(k, tuple(v.shape)) for k, v in ae.state_dict().items()]
    vals = synth_state_dict([("ae." + n, s) for n, s in spec_all])
    ae.load_state_dict({n: torch.from_numpy(vals["ae." + n]) for n, _ in spec_all})
    ae.eval()
    spec = [(n, s) for n, s in spec_all if n.startswith("keypoint_encoder.fc_layer") or n.startswith("decoder.")]
    rs = np.random.RandomState(31)
    kp = synth_keypoints(B, 16, seed=9)
    feat = (0.5 * rs.standard_normal((B, 16, 48))).astype(np.float32)
    label = np.zeros(B, np.int64)
    with torch.no_grad():
        new_xyz = ae.keypoint_encoder.upsample_points(torch.from_numpy(feat), torch.from_numpy(kp))
        l_xyz = ae.decoder(torch.from_numpy(kp), torch.from_numpy(feat), new_xyz, ts=None, label=torch.from_numpy(label))
    res = {"keypoint": kp, "feature": feat, "label": label, "decoder_configs_json": np.array(json.dumps(decs)),
           "level1": l_xyz[1].numpy(), "level2": l_xyz[2].numpy(), "level3": l_xyz[3].numpy()}
    res["spec_names"], res["spec_shapes"] = spec_arrays(spec)
    np.savez_compressed(os.path.join(out, "golden_decode.npz"), **res)
    print("decode", [tuple(x.shape) for x in l_xyz], "params", sum(int(np.prod(s)) for _, s in spec))


def gen_encode(out, B=2, N=2048):
    """PointAutoencoder.encode of the reference (airplane AE config, posterior mode) on a synthetic surface-like cloud:
    PointNet2Encoder (FPS 2048 -> 1024 -> 256 -> 64 -> 32, kNN-32 SA stack) + keypoint_encoder.propagate_feature."""
    from data_utils.json_reader import read_json_file, autoencoder_read_config
    from models.autoencoder import PointAutoencoder
    d = CFG + "autoencoder_configs/"
    cfg = read_json_file(d + "config_autoencoder_s3_kl_1e-5_16_keypoints_latent_dim_16_32_normal_weight_0_0_0.1_with_augm_kp_noise_0.04_airplane.json")
    enc, decs = autoencoder_read_config(d, cfg)
    ae = PointAutoencoder(enc, decs, apply_kl_regularization=True, kl_weight=1e-5)
    spec_all = [(k, tuple(v.shape)) for k, v in ae.state_dict().items()]
    vals = synth_state_dict([("ae." + n, s) for n, s in spec_all])
    ae.load_state_dict({n: torch.from_numpy(vals["ae." + n]) for n, _ in spec_all})
    ae.eval()
    spec = [(n, s) for n, s in spec_all if n.startswith("encoder.") or n.startswith("keypoint_encoder.")]
    rs = np.random.RandomState(41)
    # points on a few random ellipsoid shells in [-1,1]^3 with unit normals (in_fea_dim = 3)
    u = rs.standard_normal((B, N, 3)).astype(np.float32)
    u /= np.linalg.norm(u, axis=2, keepdims=True)
    radii = rs.uniform(0.3, 0.9, (B, 1, 3)).astype(np.float32)
    pts = (u * radii).astype(np.float32)
    nrm = u / radii
    nrm = (nrm / np.linalg.norm(nrm, axis=2, keepdims=True)).astype(np.float32)
    pc = np.concatenate([pts, nrm], axis=2).astype(np.float32)
    from oracle import ops as O
    kidx = O.furthest_point_sampling(pts, 16)
    kp = np.take_along_axis(pts, kidx[..., None].astype(np.int64), axis=1).astype(np.float32)
    label = np.zeros(B, np.int64)
    with torch.no_grad():
        out_e, l_xyz, _ = ae.encoder(torch.from_numpy(pc), ts=None, label=torch.from_numpy(label))
        feat = ae.encode(torch.from_numpy(pc), torch.from_numpy(kp), ts=None, label=torch.from_numpy(label),
                         sample_posterior=False)
    res = {"pointcloud": pc, "keypoint": kp, "label": label, "encoder_config_json": np.array(json.dumps(enc)),
           "decoder_configs_json": np.array(json.dumps(decs)), "encoder_out": out_e.numpy(),
           "encoder_xyz_last": l_xyz[-1].numpy(), "feature_at_keypoint": feat.numpy()}
    res["spec_names"], res["spec_shapes"] = spec_arrays(spec)
    np.savez_compressed(os.path.join(out, "golden_encode.npz"), **res)
    print("encode", tuple(out_e.shape), tuple(feat.shape), "params", sum(int(np.prod(s)) for _, s in spec))


def gen_encoder_switches(out, B=2, N=128):
    """PointNet2Encoder.forward (the autoencoder's encoder class) with the parent project's switches: swish, position encoding, bn_first
    (leading convolution), global attention behind level 0 -- a small architecture under the airplane AE encoder's other settings
    (reference pointnet2_feature_extractor.py:25-218; no global feature here: with position encoding the reference widens the global
    PointNet's input twice, :73-78, and cannot run)"""
    from data_utils.json_reader import read_json_file, autoencoder_read_config
    from models.pointnet2_feature_extractor import PointNet2Encoder
    d = CFG + "autoencoder_configs/"
    cfg = read_json_file(d + "config_autoencoder_s3_kl_1e-5_16_keypoints_latent_dim_16_32_normal_weight_0_0_0.1_with_augm_kp_noise_0.04_airplane.json")
    enc, _ = autoencoder_read_config(d, cfg)
    hp = copy.deepcopy(enc)
    hp["architecture"] = dict(hp["architecture"], npoint=[64, 16], radius=[0, 0], nsample=[8, 8], feature_dim=[16, 32, 64])
    hp.update(activation="swish", use_position_encoding=True, position_encoding_multires=3, bn_first=True, include_global_feature=False,
              global_attention_setting={"use_global_attention_module": True, "global_attention_layer_index": [0],
                                        "attention_bn": True, "last_activation": True})
    res = {"config_json": np.array(json.dumps(hp))}
    net = PointNet2Encoder(copy.deepcopy(hp))
    spec = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    sd = synth_state_dict(spec)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval()
    res["spec_names"], res["spec_shapes"] = spec_arrays(spec)
    rs = np.random.RandomState(23)
    pc = rs.standard_normal((B, N, 3 + hp["in_fea_dim"])).astype(np.float32)
    pc[:, :, :3] *= 0.5
    label = np.array([0, 4][:B], np.int64)
    with torch.no_grad():
        y, l_xyz, _ = net(torch.from_numpy(pc), ts=None, label=torch.from_numpy(label))
    res["pointcloud"], res["label"], res["out"], res["xyz_last"] = pc, label, y.numpy(), l_xyz[-1].numpy()
    np.savez_compressed(os.path.join(out, "golden_encoder_switches.npz"), **res)
    print("encoder switches", tuple(y.shape), "params", sum(int(np.prod(s_)) for _, s_ in spec))


def gen_ops(out):
    from oracle import ops as O
    rs = np.random.RandomState(3)
    res = {}

    def cloud(B, N, dup=0, origin=0, scale=1.0):
        p = rs.uniform(-1, 1, (B, N, 3)).astype(np.float32) * np.float32(scale)
        for b in range(B):
            for _ in range(dup):  # exact duplicates -> exact distance ties
                i, j = rs.randint(0, N, 2)
                p[b, j] = p[b, i]
            for _ in range(origin):  # inside the 1e-3 origin ball (|p|^2 <= 1e-3): FPS skips them
                p[b, rs.randint(0, N)] = rs.uniform(-0.015, 0.015, 3).astype(np.float32)
        return p

    fps_cases = [(2, 16, 16, 0, 0), (2, 64, 16, 3, 2), (2, 100, 37, 5, 3), (1, 256, 128, 4, 4), (2, 1024, 256, 8, 6),
                 (1, 700, 64, 0, 0), (1, 5, 5, 1, 1), (1, 1500, 40, 10, 5), (1, 33, 1, 0, 0)]
    for ci, (B, N, m, dup, org) in enumerate(fps_cases):
        p = cloud(B, N, dup, org)
        if ci == 2:
            p[:, 0] = 0.0  # index 0 itself invalid
        idx, temp = O.furthest_point_sampling(p, m, return_temp=True)
        res["fps%d_in" % ci], res["fps%d_idx" % ci], res["fps%d_temp" % ci] = p, idx, temp
    # grid cloud: massive exact ties exercise the reduction-tree tie-break
    g = np.stack(np.meshgrid(np.arange(8), np.arange(8), np.arange(4), indexing="ij"), -1).reshape(1, -1, 3)
    g = (g.astype(np.float32) - 2.0) * 0.25
    res["fps_grid_in"] = g
    res["fps_grid_idx"], res["fps_grid_temp"] = O.furthest_point_sampling(g, 100, return_temp=True)
    res["n_fps"] = np.array(len(fps_cases))

    bq_cases = [(2, 64, 16, 0.4, 8), (2, 100, 33, 0.25, 16), (1, 16, 16, 5.0, 32), (1, 300, 50, 0.05, 4),
                (2, 1024, 256, 0.2, 32)]
    for ci, (B, N, M, r, ns) in enumerate(bq_cases):
        p = cloud(B, N, 2, 0); q = cloud(B, M, 0, 0, 1.2)
        q[:, 0] = p[:, 3]
        idx, cnt = O.ball_query(q, p, r, ns)
        res["bq%d_xyz" % ci], res["bq%d_new" % ci] = p, q
        res["bq%d_r" % ci], res["bq%d_ns" % ci] = np.float32(r), np.array(ns)
        res["bq%d_idx" % ci], res["bq%d_cnt" % ci] = idx, cnt
    res["n_bq"] = np.array(len(bq_cases))

    knn_cases = [(2, 16, 16, 16), (2, 16, 16, 8), (2, 40, 100, 32), (1, 256, 1024, 32), (2, 64, 16, 8), (1, 7, 3, 3),
                 (1, 300, 257, 4)]
    for ci, (B, N1, N2, K) in enumerate(knn_cases):
        p2 = cloud(B, N2, 2 if N2 > 8 else 0, 0)
        p1 = p2.copy() if N1 == N2 else cloud(B, N1, 0, 0)
        d, i = O.knn_points(p1, p2, K)
        res["knn%d_p1" % ci], res["knn%d_p2" % ci], res["knn%d_K" % ci] = p1, p2, np.array(K)
        res["knn%d_d" % ci], res["knn%d_i" % ci] = d, i
    res["n_knn"] = np.array(len(knn_cases))
    # three_nn incl. m < 3 (trailing idx 0, dist inf)
    tn_cases = [(2, 50, 20), (1, 16, 2), (1, 300, 64), (1, 9, 1)]
    for ci, (B, n, m) in enumerate(tn_cases):
        u = cloud(B, n); k = cloud(B, m, 1 if m > 4 else 0)
        d, i = O.three_nn(u, k)
        res["tn%d_u" % ci], res["tn%d_k" % ci], res["tn%d_d" % ci], res["tn%d_i" % ci] = u, k, d, i
    res["n_tn"] = np.array(len(tn_cases))
    np.savez_compressed(os.path.join(out, "golden_ops.npz"), **res)
    print("ops done")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    torch.manual_seed(0)
    want = set(a.only.split(",")) if a.only else {"ops", "blocks", "pos", "feat", "variants", "condition", "switches", "resample", "sched", "train", "decode", "encode"}
    if "ops" in want:
        gen_ops(a.out)
    if "blocks" in want:
        gen_blocks(a.out)
    if "pos" in want:
        net, cfg = gen_denoiser("pos", POS_CFG, a.out)
        gen_sampler_pos(net, cfg, a.out)
    if "feat" in want:
        net, cfg = gen_denoiser("feat", FEAT_CFG, a.out)
        gen_sampler_feat(net, cfg, a.out)
    if "featfull" in want:  # (not in the default set: ~10 min of reference CPU time; the file is committed)
        import tempfile
        net, cfg = gen_denoiser("feat", FEAT_CFG, tempfile.mkdtemp())  # (the denoiser fixture itself is not rewritten)
        gen_sampler_feat_full(net, cfg, a.out)
    if "variants" in want:
        gen_denoiser_variants(a.out)
    if "condition" in want:
        gen_denoiser_condition(a.out)
    if "switches" in want:
        gen_denoiser_switches(a.out)
        gen_encoder_switches(a.out)
    if "resample" in want:
        gen_sampler_feat_resample(a.out)
    if "sched" in want:
        gen_schedules(a.out)
    if "train" in want:
        gen_train(a.out)
    if "decode" in want:
        gen_decode(a.out)
    if "encode" in want:
        gen_encode(a.out)
