"""CPU: host-side logic -- config reader, built-in configs vs the reference's shipped configs (via the golden fixture),
parameter spec, schedules, plan structure / FLOP accounting, checkpoint layout."""
import json

import numpy as np
import pytest
import torch

from conftest import golden_spec, load_golden
from slide_amd import configs, model_spec
from slide_amd.json_reader import read_json_file


def test_builtin_configs_match_reference_configs():
    for name, cfg in (("pos", configs.position_ddpm_config()), ("feat", configs.feature_ddpm_config())):
        g = load_golden("golden_denoiser_%s.npz" % name)
        assert json.loads(str(g["config_json"])) == cfg["pointnet_config"]
        assert dict(golden_spec(g)) == dict(model_spec.denoiser_param_spec(cfg["pointnet_config"]))
    g = load_golden("golden_sampler_feat.npz")
    assert json.loads(str(g["config_json"])) == configs.feature_ddpm_config()["standard_diffusion_config"]


def test_json_reader_restores_string_lists(tmp_path):
    p = tmp_path / "c.json"
    p.write_text(json.dumps({"a": {"npoint": "[16, 16]", "name": "x", "cats": "['02691156']", "n": 3, "bad": "[1,"}}))
    c = read_json_file(str(p))
    assert c["a"]["npoint"] == [16, 16] and c["a"]["cats"] == ["02691156"] and c["a"]["name"] == "x" and c["a"]["bad"] == "[1,"


def test_schedules_match_reference_tables():
    from slide_amd.diffusion import calc_diffusion_hyperparams, latent_diffusion_params
    g = load_golden("golden_sampler_pos.npz")
    dh = calc_diffusion_hyperparams(1000, 1e-4, 0.02)
    for k in ("Beta", "Alpha", "Alpha_bar"):
        assert np.array_equal(dh[k], g["sched_" + k])
    assert np.abs(dh["Sigma"] - g["sched_Sigma"]).max() <= 1.5e-8  # torch's CPU sqrt is 1 ulp low on 9 entries
    g = load_golden("golden_sampler_feat.npz")
    dp = latent_diffusion_params(json.loads(str(g["config_json"])))
    for k in ("logvar", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_mean_coef1", "posterior_mean_coef2"):
        assert np.array_equal(dp[k], g["sched_" + k])


def test_every_beta_schedule_matches_the_reference_tables():
    """golden_schedules.npz: Diffusion.init_diffusion_parameters of the reference (diffusion.py:158-208) for every schedule
    name its get_beta_schedule (:12-28) can produce x both variance types x T in {1000, 50}; product and oracle restatements"""
    import warnings
    from oracle import denoiser_np as D
    from slide_amd.diffusion import get_beta_schedule, latent_diffusion_params
    g = load_golden("golden_schedules.npz")
    cfgs = json.loads(str(g["configs_json"]))
    assert {c["beta_schedule"] for c in cfgs} == {"linear", "quad", "const", "jsd"}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # ('jsd' ends at beta = 1: alpha_bar = 0 -> inf / nan entries, as in the reference)
        for k, c in enumerate(cfgs):
            for name, dp in (("product", latent_diffusion_params(c)), ("oracle", D.latent_diffusion_params(c))):
                assert dp["T"] == c["num_diffusion_timesteps"]
                for nm in ("logvar", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_mean_coef1",
                           "posterior_mean_coef2"):
                    assert np.array_equal(dp[nm], g["c%d_%s" % (k, nm)], equal_nan=True), (name, c["beta_schedule"], nm)
    # the two names whose helper the reference never defines: DDPM's warm-up ramp
    for nm, frac in (("warmup10", 0.1), ("warmup50", 0.5)):
        b = get_beta_schedule(nm, 1e-4, 0.02, 1000)
        n = int(1000 * frac)
        assert b.shape == (1000,) and b[0] == 1e-4 and np.all(b[n - 1:] == 0.02) and np.all(np.diff(b[:n]) > 0)
    with pytest.raises(NotImplementedError):
        get_beta_schedule("cosine", 1e-4, 0.02, 10)


def test_plan_structure_and_flops():
    """the plan can be built without a GPU; its GEMMs cover every conv MAC of the reference network, minus the
    query half of attention weight_conv.2 which is evaluated per point instead of per neighbour"""
    from slide_amd.engine import OP_GEMM, DenoiserEngine
    from slide_amd.synth import synth_state_dict
    for cfg, conv_macs in ((configs.position_ddpm_config(), 38977024 - 471040), (configs.feature_ddpm_config(), None)):
        hp = cfg["pointnet_config"]
        sd = synth_state_dict(model_spec.denoiser_param_spec(hp))
        e = DenoiserEngine(hp, sd, 2, torch.device("cpu"), prec="fp32")
        assert sum(1 for o in e.ops if o.kind == OP_GEMM) == 34  # (the two FP query GEMMs ride on the SA blocks that read the same table)
        saved = 0
        for pfx, K in (("SA_modules.0.attention_modules.0", 16), ("SA_modules.1.attention_modules.0", 16),
                       ("FP_modules.0.attention_module", 8), ("FP_modules.1.attention_module", 8)):
            C1 = sd[pfx + ".feat_conv.weight"].shape[0]
            inter = sd[pfx + ".weight_conv.2.weight"].shape[0]
            saved += C1 * inter * 16 * (K - 1)
        if conv_macs is not None:  # SURVEY.md appendix A.1 total minus the linear (fc / fc_t) layers
            assert e.flops // 2 // 2 + saved == conv_macs
        try:
            e.forward(np.zeros((2, 16, e.cx)), np.zeros(2), np.zeros(2))
            assert False, "a plan must not run without a GPU"
        except RuntimeError:
            pass


def test_split_plan_structure_and_fallback():
    """round 5: the split plan is the pair decomposition in the split arithmetic (generated-X GEMMs, dual launches, the SA blocks'
    second_mlp -> rest_mlp chained, split attention tails): the position step is 28 launches (round 4: 49).  Its kernels scale the
    weights' high terms by 2^11 in fp16, so a checkpoint with a convolution weight >= 31 falls back to the fp32-structured plan."""
    import warnings
    from slide_amd.engine import OP_ATTN_TAIL, OP_GEMM_GX, OP_GEMM_GX_DUAL, DenoiserEngine
    from slide_amd.synth import synth_state_dict
    hp = configs.position_ddpm_config()["pointnet_config"]
    sd = synth_state_dict(model_spec.denoiser_param_spec(hp))
    e = DenoiserEngine(hp, sd, 4, torch.device("cpu"), prec="split", per_sample_t=False, t_table=10)
    kinds = [o.kind for o in e.ops]
    assert e.use_gxs and len(kinds) == 28 and kinds.count(OP_GEMM_GX_DUAL) == 4 and kinds.count(OP_ATTN_TAIL) == 4
    chained = [o for o in e._dual_keep if o[1].p[12]]          # mode-0 ops of the dual launches that carry a chained layer
    assert len(chained) == 2 and all(o[1].i[4] == 8 for o in chained)  # the two SA blocks (16 x 16-row samples)
    big = dict(sd)
    k = "SA_modules.0.mlps.0.second_mlp.0.weight"
    big[k] = sd[k].copy(); big[k].flat[0] = 40.0
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        e2 = DenoiserEngine(hp, big, 4, torch.device("cpu"), prec="split", per_sample_t=False, t_table=10)
    assert not e2.use_gxs and any("falls back" in str(x.message) for x in w)
    assert not any(o.kind in (OP_GEMM_GX, OP_GEMM_GX_DUAL, OP_ATTN_TAIL) for o in e2.ops) and len(e2.ops) == 49


def test_checkpoint_layout(tmp_path):
    from slide_amd.checkpoint import load_denoiser_state
    from slide_amd.synth import synth_state_dict
    hp = configs.position_ddpm_config()["pointnet_config"]
    spec = model_spec.denoiser_param_spec(hp)
    sd = {k: torch.from_numpy(v) for k, v in synth_state_dict(spec).items()}
    ema = {k: v + 1 for k, v in list(sd.items())[:5]}
    f = tmp_path / "pointnet_ckpt_1.pkl"
    torch.save({"model_state_dict": sd, "ema_state_list": [{}, ema], "iter": 1}, str(f))
    out = load_denoiser_state(hp, str(f), ema_idx=1)
    k0 = list(ema)[0]
    assert np.array_equal(out[k0], ema[k0].numpy()) and np.array_equal(out["fc_lyaer.3.bias"], sd["fc_lyaer.3.bias"].numpy())


def test_bench_self_launch_argv(monkeypatch):
    """`python bench.py --gpus N` outside a launcher re-executes itself as N ranks under torch.distributed.run (the
    reference's launcher spawns its per-GPU workers itself: pointnet2/distributed.py:171-182)"""
    import importlib.util
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    argv = bench.relaunch_argv(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], port=29999)
    assert argv[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert argv[argv.index("--nproc-per-node") + 1] == "8" and "--nnodes=1" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1" and argv[argv.index("--master-port") + 1] == "29999"
    k = argv.index(os.path.join(root, "bench.py"))
    assert argv[k + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    # main(): with --gpus 2 and no WORLD_SIZE it must spawn (not assert) and pass the child's exit status on
    calls = []
    import subprocess
    monkeypatch.setattr(subprocess, "call", lambda cmd, **kw: (calls.append(cmd), 7)[1])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    try:
        bench.main()
        raise AssertionError("main() should have exited with the launcher's status")
    except SystemExit as e:
        assert e.code == 7
    assert len(calls) == 1 and calls[0][calls[0].index("--nproc-per-node") + 1] == "2" and calls[0][-4:] == ["--gpus", "2", "--steps", "3"]


@pytest.mark.parametrize("name", ["fp3nn", "bnfirst", "fp3nn_bnfirst", "nobn", "local", "global", "both", "swish_pe_ga", "concat_partial"])
def test_denoiser_configuration_branches_are_checkpoint_compatible(name):
    """CPU: the module tree PointNet2CloudCondition builds for the non-shipped configuration branches (three-nearest-neighbour FP
    module, bn_first with its leading convolution and activation + conv head, bn False; the condition-cloud forms: local feature
    transfer, global feature, both) has the reference's state-dict names and shapes (golden_denoiser_{variants,condition}.npz:
    generated by importing the reference, tools/gen_golden.py `--only variants,condition`); the forward
    itself is a -m gpu test (tests/test_hip_modules.py)."""
    import os
    import sys
    from conftest import REPO
    sys.path.insert(0, os.path.join(REPO, "pointnet2"))
    from models.pointnet2_with_pcld_condition import PointNet2CloudCondition
    g = load_golden("golden_denoiser_condition.npz" if name in ("local", "global", "both") else
                    "golden_denoiser_switches.npz" if name in ("swish_pe_ga", "concat_partial") else "golden_denoiser_variants.npz")
    hp = json.loads(str(g[name + "_config_json"]))
    net = PointNet2CloudCondition(hp)
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == dict(golden_spec(g, name + "_spec"))
    assert g[name + "_eps"].shape == (2, 16, 6 if name == "swish_pe_ga" else 3) and np.isfinite(g[name + "_eps"]).all()
