#!/usr/bin/env python
"""bench.py -- latent shapes/sec of the position + feature DDPM sampling hot path on N MI355X.

Workload (BASELINE.json configs[1]/[2], per GPU): batch 256 latent-point sets (16 points each);
  step = ONE reverse-diffusion step of the position DDPM (airplane config, 3-dim)  +
         ONE reverse-diffusion step of the feature DDPM (chair config, 48-dim feature + 3-dim key points)
  over the whole batch: denoiser forward + DDPM update + in-kernel noise.  The position plan (one chain of 256) and the
  feature plan (three independent sub-batches) are replayed concurrently on their own streams, launched eagerly and
  round-robin per step by one host thread inside the library (in steady-state generation batch i's feature chains
  overlap batch i+1's position chain); `--replay threads` gives every chain its own host thread, `--replay graph`
  replays captured hipGraphs.
A generated shape needs 1000 + 1000 such steps, so  value = n_gpus * batch / (1000 * seconds_per_step).
`--steps 1000` is therefore exactly one complete generation of the batch.  Synthetic random-init weights,
synthetic key points, inputs resident in HBM.  N > 1: one process per GPU (torchrun), batch shards are
independent (weak scaling), one RCCL all-gather of the (B,16,51) latents closes the timed region.

Extra objects on the JSON line:
  roofline     the dominant kernel (the MFMA kernel with the largest share of a feature step's device time): algorithmic
               FLOPs of its launches / their device time, measured with HIP events on the launch stream in an
               instrumented eager replay right after the timed region (the timed region launches without events)
  cpu_baseline the numpy/C oracle ("port" of the reference algorithm; the reference has no CPU path for its native
               ops) timed on the host cores on a bounded sample of the same workload
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

# MI355X dense MFMA peaks (MI355X_MICROARCH.md); "split" = three fp16 MFMAs per logical product
PEAK_TFLOPS = {"fp16": 2500.0, "fp32": 157.3, "split": 2500.0 / 3}


def cpu_baseline(batch=16, budget_s=25.0):
    """the numpy/C oracle ("port": the reference has no CPU path for its native ops) on the host cores: forwards of each
    denoiser on a bounded sample (batch 16; numpy's per-sample cost does not improve with the batch -- its GroupNorm /
    broadcast-matmul temporaries grow with it -- and the oracle's arithmetic is frozen by the bit-sensitive decode goldens),
    repeated while the budget lasts, x1000 steps"""
    from oracle import denoiser_np as D
    from slide_amd import configs, model_spec
    from slide_amd.synth import synth_keypoints, synth_state_dict
    try:
        from threadpoolctl import threadpool_info
        threads = max([int(i.get("num_threads", 1)) for i in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count()
    rs = np.random.RandomState(0)
    per = {}
    for name, cfg in (("pos", configs.position_ddpm_config()), ("feat", configs.feature_ddpm_config())):
        hp = cfg["pointnet_config"]
        sd = synth_state_dict(model_spec.denoiser_param_spec(hp))
        x = rs.standard_normal((batch, 16, 3 + hp["in_fea_dim"])).astype(np.float32)
        x[:, :, :3] = synth_keypoints(batch)
        ts = np.full((batch,), 500, np.float32)
        label = np.zeros((batch,), np.int64)
        D.denoiser_forward(hp, sd, x[:4], ts[:4], label[:4])  # warm (library load, BLAS thread pool)
        n, t0 = 0, time.time()
        while n < 1 or (time.time() - t0 < budget_s / 2 and n < 20):
            D.denoiser_forward(hp, sd, x, ts, label)
            n += 1
        per[name] = (time.time() - t0) / n
    sec_per_step = per["pos"] + per["feat"]
    return {"value": batch / (1000.0 * sec_per_step), "unit": "shapes/s", "cores": threads, "kind": "port",
            "sample": "numpy/C oracle denoiser forward (pos+feat) at batch %d, %.2f+%.2f s/step (BLAS threads %d of %d host cores), "
                      "extrapolated x1000 steps" % (batch, per["pos"], per["feat"], threads, os.cpu_count())}


def relaunch_argv(gpus, argv, port=None):
    """`python bench.py --gpus N` outside a launcher: the command line that re-runs this script as N ranks, one process
    per GPU (the reference's launcher spawns its per-GPU workers itself too: pointnet2/distributed.py:171-182).  rank 0
    owns stdout (the JSON line); the other ranks' stdout goes to stderr inside main()."""
    if port is None:
        import socket
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def _decode_profile_note():
    """dominant kernel of the decode leg from the newest committed rocprofv3 summary (profiles/r*_decode_fp16_kernel_stats.md)"""
    import glob
    import re
    fs = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_decode_fp16_kernel_stats.md")))
    if not fs:
        return None
    for line in open(fs[-1]):
        m = re.match(r"\| ([^|]+) \| (\d+) \| ([0-9.]+) \| ([0-9.]+) \| ([0-9.]+) \|", line)
        if m and m.group(1).strip() != "kernel":
            return {"source": os.path.relpath(fs[-1], REPO), "dominant_kernel": m.group(1).strip(), "share_pct": float(m.group(5)),
                    "avg_us": float(m.group(4)), "calls": int(m.group(2))}
    return {"source": os.path.relpath(fs[-1], REPO)}


def _decode_hbm_note():
    """HBM-side rate of the decode leg from the newest committed PMC summary (profiles/r*_decode_hbm_pmc.md, tools/decode_hbm.sh):
    the leg's GEMMs run on millions of rows at K, N <= 256 -- it is HBM-bound, not matrix-pipe-bound"""
    import glob
    import re
    fs = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_decode_hbm_pmc.md")))
    if not fs:
        return None
    m = re.search(r"All kernels: (\d+) MB in ([0-9.]+) ms of kernel time = (\d+) GB/s = ([0-9.]+) of the 8 TB/s peak", open(fs[-1]).read())
    if not m:
        return {"source": os.path.relpath(fs[-1], REPO)}
    return {"source": os.path.relpath(fs[-1], REPO), "bound": "hbm", "moved_MB": int(m.group(1)), "kernel_ms": float(m.group(2)),
            "achieved": int(m.group(3)), "peak": 8000, "unit": "GB/s", "frac": float(m.group(4)),
            "note": "FETCH_SIZE x2 + WRITE_SIZE over every kernel of tools/time_decode.py (fp16 operands), from the committed profile"}


def decode_leg(dev, B):
    """BASELINE configs[4] on the driver's record: PointAutoencoder.decode of B synthetic latents to
    (B, 2048, 6) on the HIP module path (fp16 MFMA operands), shapes/s over three timed passes after two warm-up passes; not part
    of `value`.  Round 6: this arithmetic (SLIDE_MODULE_PREC=fp16) is the one the generation CLIs decode in by default (`--prec
    mixed`, slide_amd.generation.module_prec_of) and is pinned to the reference's decode of golden_decode.npz by
    tests/test_hip_modules.py::test_autoencoder_decode_fp16_operands_matches_reference (per level: reference points -> own FPS
    candidates <= 1e-4, Chamfer <= 1e-5; measured 1e-5 / 2e-7); `shapes_per_s_fp32_mode` = the same decode with exact fp32 MFMA
    (`--prec fp32`; test_autoencoder_decode_matches_reference)."""
    import torch
    from slide_amd.synth import synth_keypoints, synth_state_dict
    sys.path.insert(0, os.path.join(REPO, "pointnet2"))
    prev = os.environ.get("SLIDE_MODULE_PREC")
    os.environ["SLIDE_MODULE_PREC"] = "fp16"
    try:
        from models.autoencoder import PointAutoencoder
        g = np.load(os.path.join(REPO, "tests", "golden", "golden_decode.npz"))
        decs = json.loads(str(g["decoder_configs_json"]))
        spec = [(str(n), tuple(int(x) for x in str(s_).split(","))) for n, s_ in zip(g["spec_names"], g["spec_shapes"])]
        vals = synth_state_dict([("ae." + n, s_) for n, s_ in spec])
        ae = PointAutoencoder(None, decs, True)
        ae.load_state_dict({n: torch.from_numpy(vals["ae." + n]) for n, _ in spec})
        ae = ae.to(dev).eval()
        kp = torch.from_numpy(synth_keypoints(B)).to(dev)
        feat = 0.5 * torch.randn(B, 16, 48, device=dev)
        lab = torch.zeros(B, dtype=torch.long, device=dev)
        for _ in range(2):
            o = ae.decode(kp, feat, label=lab)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(3):
            o = ae.decode(kp, feat, label=lab)
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / 3
        # the exact-fp32 module mode on the same latents (a fresh model: the layers read the mode when they are built)
        os.environ["SLIDE_MODULE_PREC"] = "fp32"
        ae32 = PointAutoencoder(None, decs, True)
        ae32.load_state_dict({n: torch.from_numpy(vals["ae." + n]) for n, _ in spec})
        ae32 = ae32.to(dev).eval()
        ae32.decode(kp, feat, label=lab)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        ae32.decode(kp, feat, label=lab)
        torch.cuda.synchronize(dev)
        dt32 = time.perf_counter() - t1
        del ae32
        os.environ["SLIDE_MODULE_PREC"] = "fp16"
        return {"workload": "BASELINE configs[4]: autoencoder decode of %d latents (16 x 51) to %d x 2048 x 6, HIP module path, fp16 MFMA "
                            "operands / fp32 accumulate (the generation CLIs' default arithmetic, --prec mixed)" % (B, B),
                "shapes_per_s": round(B / dt, 1), "ms_per_batch": round(dt * 1e3, 2), "shapes_per_s_fp32_mode": round(B / dt32, 1),
                "parity_test": "tests/test_hip_modules.py::test_autoencoder_decode_fp16_operands_matches_reference",
                "gflop_per_shape": 16.6, "tflops": round(16.6e9 * B / dt / 1e12, 1), "finite": bool(torch.isfinite(o).all()),
                # whole-leg fraction of the dense fp16 MFMA peak (the leg also holds FPS / kNN / grouping kernels, which are not
                # MFMA work); its per-kernel breakdown is the committed rocprofv3 summary
                "roofline": {"bound": "mfma", "achieved": round(16.6e9 * B / dt / 1e12, 1), "peak": PEAK_TFLOPS["fp16"],
                             "unit": "TFLOP/s", "frac": round(16.6e9 * B / dt / 1e12 / PEAK_TFLOPS["fp16"], 4),
                             "kernel_breakdown": _decode_profile_note(), "hbm": _decode_hbm_note()}}
    finally:
        if prev is None:
            os.environ.pop("SLIDE_MODULE_PREC", None)
        else:
            os.environ["SLIDE_MODULE_PREC"] = prev


def configs_leg(dev, B, a, pc, fc, sd_p, sd_f, gen, feat_chains, pos_sampler, steps=20, warmup=5):
    """`configs` object of the JSON line (round 6): every BASELINE config on the driver's record, each timed like
    the headline (W untimed steps, barrier + synchronize, K timed steps, synchronize; the reference's convention: wall time of the
    whole sampling loop, pointnet2/mesh_evaluation.py:102,126), in the benched arithmetic, after the headline's timed region:
      config2_position_ddpm: BASELINE configs[1] ALONE -- the airplane position DDPM, one chain of B shapes (`pos_batch_multiple` 1) and
        one chain of 2 B shapes (multiple 2, the headline arrangement's position chain), shapes/s of the position DDPM only
        (a shape = 1000 position steps);
      config3_feature_ddpm: BASELINE configs[2] ALONE -- the chair feature DDPM on fixed key points, B shapes as the headline's
        sub-batches, shapes/s of the feature DDPM only;
      joint_pos_batch_multiple_1: the headline arrangement with the position chain over B shapes stepping every round (config 2 at
        its literal batch beside config 3) -- the headline runs it over 2 B shapes at half cadence;
      config4_five_category_shard: BASELINE configs[3]'s per-GPU shard -- B shapes as five category segments (labels 0, 2, 3, 4, 6; one
        position + one feature weight set per category; ten chains on four streams), shapes/s, and per category the forward error of
        the benched arithmetic against the fp32 mode (relative L2, batch 8) for both nets;
      config4_rank1_of_8_shard: the same config as the 8-GPU run shards it -- the B shapes of rank 1 of an 8 B-shape run in
        category-major order (two segments: four chains)."""
    import torch
    from slide_amd.diffusion import EagerChainsSampler, FeatureSampler, PositionSampler
    from slide_amd.engine import DenoiserEngine
    from slide_amd.generation import POS_CU_SHARE, CategoryChains
    from slide_amd.synth import synth_keypoints, synth_state_dict
    from slide_amd import model_spec
    out = {"steps": steps, "warmup": warmup}

    def timed(joint, begin, streams):
        begin()
        joint.advance(warmup)
        for st_ in streams:
            st_.synchronize()
        torch.cuda.synchronize(dev)
        begin()
        for st_ in streams:
            st_.synchronize()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        joint.advance(steps)
        for st_ in streams:
            st_.synchronize()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / steps

    # The stream -> hardware-queue assignment of the runtime decides whether concurrent chains overlap (four queues; two chains on one
    # queue serialise: DESIGN.md section 9 item 5a'), and it depends on the order in which streams were created and first used.  The
    # concurrent legs therefore run on the HEADLINE's own chains and streams: its feature samplers as they are, a batch-B position
    # chain on the headline position chain's stream.
    feats = list(feat_chains)
    sizes = [f_.B for f_, _, _ in feats]
    P = len(feats)

    def begin_feats():
        for f_, l_, k_ in feats:
            f_.begin(l_, k_, torch.randn(f_.B, 16, 51, device=dev, generator=gen))

    # config 3 alone
    s = timed(EagerChainsSampler([f_ for f_, _, _ in feats]), begin_feats, [f_.stream for f_, _, _ in feats])
    out["config3_feature_ddpm"] = {"batch": B, "sub_batches": sizes, "prec": a.prec, "ms_per_step": round(s * 1e3, 5),
                                   "shapes_per_s": round(B / (1000.0 * s), 2)}
    # config 2 alone, at its literal batch and at the headline's multiple
    c2 = {}
    for mult in (1, 2):
        ps = PositionSampler(pc["pointnet_config"], sd_p, B * mult, dev, pc["diffusion_config"], prec=a.pos_prec, seed=3100 + mult, use_graph=False)
        lab = torch.zeros(ps.B, dtype=torch.int64, device=dev)
        s = timed(EagerChainsSampler([ps]), lambda: ps.begin(lab, torch.randn(ps.B, 16, 3, device=dev, generator=gen)), [ps.stream])
        c2["pos_batch_multiple_%d" % mult] = {"batch": ps.B, "prec": a.pos_prec, "ms_per_step": round(s * 1e3, 5),
                                              "shapes_per_s": round(ps.B / (1000.0 * s), 2)}
        del ps
    out["config2_position_ddpm"] = c2
    # the joint arrangement with the position chain at config 2's literal batch
    ps = PositionSampler(pc["pointnet_config"], sd_p, B, dev, pc["diffusion_config"], prec=a.pos_prec, seed=3200, use_graph=False)
    ps.stream = ps.stream2 = pos_sampler.stream  # (the headline position chain's stream: its CU mask and its hardware queue)
    lab = torch.zeros(B, dtype=torch.int64, device=dev)

    def begin_joint():
        ps.begin(lab, torch.randn(B, 16, 3, device=dev, generator=gen))
        begin_feats()

    order = [feats[0][0], ps] + [f_ for f_, _, _ in feats[1:]]
    s = timed(EagerChainsSampler(order), begin_joint, [o_.stream for o_ in order])
    out["joint_pos_batch_multiple_1"] = {"batch": B, "ms_per_step": round(s * 1e3, 5), "shapes_per_s": round(B / (1000.0 * s), 2),
                                         "pos_stream_cus": getattr(pos_sampler, "n_cus", 0) or "all"}
    del ps, feats, order
    # config 4: (i) B shapes as ALL five category segments (a 1-GPU run of the config), (ii) the shard of rank 1 of 8 of the 8 B-shape
    # run -- BASELINE configs[3] as quoted: category-major order, contiguous shards, so a rank's B shapes span at most two categories
    spec_p, spec_f = model_spec.denoiser_param_spec(pc["pointnet_config"]), model_spec.denoiser_param_spec(fc["pointnet_config"])
    wts = lambda c: (synth_state_dict(spec_p, seed=100 + c), synth_state_dict(spec_f, seed=200 + c))
    for key, total, rank_, world_ in (("config4_five_category_shard", B, 0, 1), ("config4_rank1_of_8_shard", 8 * B, 1, 8)):
        cc = CategoryChains(total, rank_, world_, pc, fc, wts, dev, prec="mixed" if (a.prec == "fp16" and a.pos_prec == "split") else a.prec, seed=11)
        chains = []
        for k, (c, lo, hi, ps, fs) in enumerate(cc.chains):
            chains.append((ps, fs, torch.full((hi - lo,), c, dtype=torch.int64, device=dev), torch.as_tensor(synth_keypoints(hi - lo, seed=60 + k), device=dev)))

        def begin_cat():
            for ps, fs, l_, k_ in chains:
                ps.begin(l_, torch.randn(ps.B, 16, 3, device=dev, generator=gen))
                fs.begin(l_, k_, torch.randn(fs.B, 16, 51, device=dev, generator=gen))

        order = []
        for ps, fs, _, _ in chains:
            order += [fs, ps]
        s = timed(EagerChainsSampler(order), begin_cat, [o_.stream for o_ in order])
        finite = all(bool(torch.isfinite(o_.state()).all().item()) for o_ in order)
        n_sh = sum(hi - lo for _, lo, hi, _, _ in cc.chains)
        out[key] = {"shapes": n_sh, "segments": [{"label": c, "shapes": hi - lo} for c, lo, hi, _, _ in cc.chains], "chains": len(order),
                    "ms_per_step": round(s * 1e3, 5), "shapes_per_s": round(n_sh / (1000.0 * s), 2), "finite": finite}
        del cc, chains, order
    c4 = out["config4_five_category_shard"]
    c4["forward_rel_l2_vs_fp32_mode"] = {}
    nb = 8
    rs = np.random.RandomState(321)
    tsb = np.linspace(0, 999, nb).astype(np.float32)
    for c in (0, 2, 3, 4, 6):
        sd_pc, sd_fc = wts(c)
        e = {}
        for nm, cfg_, sd_, mode in (("pos", pc, sd_pc, a.pos_prec), ("feat", fc, sd_fc, a.prec)):
            hp_ = cfg_["pointnet_config"]
            xb = rs.standard_normal((nb, 16, 3 + hp_["in_fea_dim"])).astype(np.float32)
            if nm == "feat":
                xb[:, :, :3] = synth_keypoints(nb, seed=99)
            lb = np.full(nb, c, np.int64)
            y32 = DenoiserEngine(hp_, sd_, nb, dev, prec="fp32").forward(xb, tsb, lb).double()
            yb = DenoiserEngine(hp_, sd_, nb, dev, prec=mode).forward(xb, tsb, lb).double()
            e[nm] = round(float(((yb - y32).norm() / y32.norm()).item()), 7)
        c4["forward_rel_l2_vs_fp32_mode"]["label_%d" % c] = e
    return out


def parity_leg(dev, B, a, pc, fc, sd_p, sd_f, gen):
    """`parity` object of the JSON line: ONLY numbers measured in this run, on this GPU.
    forward_rel_l2_vs_fp32_mode: relative L2 of one denoiser forward IN THE BENCHED ARITHMETIC (position plan: --pos-prec,
      feature plan: --prec) against the exact-fp32 mode of the same plan (which tests/ pin to the reference goldens at <= 2e-4),
      batch 32, timesteps spread over the schedule, for two input families: x ~ N(0, 1) (the start of a chain) and x =
      key-point-like coordinates (|x| rms 0.37: the end of a position chain).  north_star's bar is 1e-3.
    forward_rel_l2_fp16_position_plan_vs_fp32_mode: the same for the fp16 POSITION plan when it is not the benched one -- why it is
      not (its error grows as the coordinates shrink).
    chain: complete 1000-step chains of 64 shapes in the benched arithmetic against the fp32 mode with equal in-kernel noise:
      per-shape relative max distance (median / max) and the NUMBER of shapes above 1e-3 -- north_star's criterion on generated latents.
    forward_max_norm_vs_fp32_mode: the forwards above in the max-norm (max |difference| / max |fp32 output|).
    fp32_mode_shapes_per_s / split_mode_shapes_per_s: throughput of the two whole-path fp32-grade modes (--prec fp32 / split)."""
    import torch
    from slide_amd.diffusion import FeatureSampler, JointSampler, PositionSampler
    from slide_amd.engine import DenoiserEngine
    from slide_amd.synth import synth_keypoints
    rs = np.random.RandomState(123)
    par = {"forward_rel_l2_vs_fp32_mode": {"pos_prec": a.pos_prec, "feat_prec": a.prec}, "chain_1000_steps_vs_fp32_mode": {}}
    nb = 32
    for nm, cfg_, sd_, mode in (("pos", pc, sd_p, a.pos_prec), ("feat", fc, sd_f, a.prec)):
        hp_ = cfg_["pointnet_config"]
        e32 = DenoiserEngine(hp_, sd_, nb, dev, prec="fp32")
        eb = DenoiserEngine(hp_, sd_, nb, dev, prec=mode)
        e16 = DenoiserEngine(hp_, sd_, nb, dev, prec="fp16") if (nm == "pos" and mode != "fp16") else None
        tsb, lb = np.linspace(0, 999, nb).astype(np.float32), np.full(nb, 4 if nm == "feat" else 0, np.int64)
        for fam in ("normal", "keypoints"):
            xb = rs.standard_normal((nb, 16, 3 + hp_["in_fea_dim"])).astype(np.float32)
            if nm == "feat" or fam == "keypoints":
                xb[:, :, :3] = synth_keypoints(nb, seed=99)
            if nm == "feat" and fam == "keypoints":
                continue  # (the feature net's coordinates are key points in both families)
            y32 = e32.forward(xb, tsb, lb).double()
            yb = eb.forward(xb, tsb, lb).double()
            par["forward_rel_l2_vs_fp32_mode"]["%s_%s" % (nm, fam)] = round(float(((yb - y32).norm() / y32.norm()).item()), 7)
            # the same forward in the MAX-norm (max |difference| / max |fp32 output|): the metric is stated next to every 1e-3 claim
            par.setdefault("forward_max_norm_vs_fp32_mode", {"pos_prec": a.pos_prec, "feat_prec": a.prec})["%s_%s" % (nm, fam)] = round(
                float(((yb - y32).abs().max() / y32.abs().max()).item()), 7)
            if e16 is not None:
                y16 = e16.forward(xb, tsb, lb).double()
                par.setdefault("forward_rel_l2_fp16_position_plan_vs_fp32_mode", {})["%s_%s" % (nm, fam)] = round(
                    float(((y16 - y32).norm() / y32.norm()).item()), 6)
        del e32, eb, e16
    nc = 64
    if a.prec == "fp16" and a.fp32_steps > 0:
        res = {}
        for prec in ("bench", "fp32"):
            ps = PositionSampler(pc["pointnet_config"], sd_p, nc, dev, pc["diffusion_config"], prec=a.pos_prec if prec == "bench" else prec,
                                 seed=77, use_graph=True)
            xT = np.random.RandomState(4).standard_normal((nc, 16, 3)).astype(np.float32)
            res["pos", prec] = ps.sample(np.zeros(nc, np.int64), xT).cpu().numpy()
            fs = FeatureSampler(fc["pointnet_config"], sd_f, nc, dev, fc["standard_diffusion_config"], prec=a.prec if prec == "bench" else prec,
                                seed=78, use_graph=True)
            xT = np.random.RandomState(5).standard_normal((nc, 16, 51)).astype(np.float32)
            res["feat", prec] = fs.sample(np.full(nc, 4, np.int64), synth_keypoints(nc), xT).cpu().numpy()
            del ps, fs
        for nm in ("pos", "feat"):
            x16, x32 = res[nm, "bench"].reshape(nc, -1), res[nm, "fp32"].reshape(nc, -1)
            per = np.abs(x16 - x32).max(axis=1) / np.abs(x32).max()
            par["chain_1000_steps_vs_fp32_mode"][nm] = {"prec": a.pos_prec if nm == "pos" else a.prec, "shapes": nc, "per_shape_rel_max_median": round(float(np.median(per)), 6),
                                                             "per_shape_rel_max_max": round(float(per.max()), 6),
                                                             "shapes_above_1e-3": int((per > 1e-3).sum())}
        # throughput of the two fp32-grade modes: "fp32" (fp32 MFMA) and "split" (the same plan, contractions as two-term fp16
        # operand splits on the fp16 matrix pipe; <= 4e-6 of the fp32 mode on a forward)
        for mode in ("fp32", "split"):
            p32 = PositionSampler(pc["pointnet_config"], sd_p, B, dev, pc["diffusion_config"], prec=mode, seed=7, use_graph=True)
            f32 = FeatureSampler(fc["pointnet_config"], sd_f, B, dev, fc["standard_diffusion_config"], prec=mode, seed=8, use_graph=True)
            j32 = JointSampler(p32, f32)
            for k_ in (2, a.fp32_steps):
                p32.begin(torch.zeros(B, dtype=torch.int64, device=dev), torch.randn(B, 16, 3, device=dev, generator=gen))
                f32.begin(torch.full((B,), 4, dtype=torch.int64, device=dev), torch.as_tensor(synth_keypoints(B, seed=5), device=dev),
                          torch.randn(B, 16, 51, device=dev, generator=gen))
                j32.synchronize(); torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                j32.advance(k_)
                j32.synchronize()
                d32 = time.perf_counter() - t1
            par["%s_mode_shapes_per_s" % mode] = round(B / (1000.0 * d32 / a.fp32_steps), 2)
            del p32, f32, j32
    return par


def main():
    # multi-process GPU work on this pool needs dmabuf IPC (RCCL / hipIpc* fail with the legacy mode)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256, help="latent shapes per GPU (BASELINE configs[1]/[2]: 256)")
    ap.add_argument("--prec", default="fp16", choices=["fp16", "fp32", "split"],
                    help="MFMA operand type (fp32 accumulate): fp16 = throughput mode; fp32 = exact parity mode (fp32 MFMA); split = fp32 "
                         "storage with the contractions as two-term fp16 operand splits on the fp16 matrix pipe (fp32-grade results)")
    ap.add_argument("--pos-prec", default=None, choices=["fp16", "fp32", "split"],
                    help="arithmetic of the POSITION plan.  Default: split beside --prec fp16 (the fp16 position plan misses north_star's "
                         "1e-3 on single forwards, DESIGN.md section 5; every forward of the default arrangement meets it), else --prec")
    ap.add_argument("--sub-batches", type=int, default=3,
                    help="independent sub-batches of the per-GPU batch replayed concurrently (scheduling only)")
    ap.add_argument("--replay", default=os.environ.get("SLIDE_REPLAY", "eager"), choices=["eager", "threads", "graph"],
                    help="eager: eager launches of all chains from one host thread, round-robin per step (default); "
                         "threads: one host thread per chain; graph: one captured hipGraph per chain and step")
    ap.add_argument("--workload", default="default", choices=["default", "five-cat"],
                    help="default: BASELINE configs[1]+[2] (one position + one feature weight set); five-cat: BASELINE configs[3] -- the "
                         "256 shapes of every GPU are its contiguous shard of a five-category run (labels 0, 2, 3, 4, 6, one weight "
                         "set per category: a chain pair per category segment), latents all-gathered")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true", help="skip the autoencoder-decode leg (BASELINE configs[4]) of the JSON line")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` object (BASELINE configs 2, 3, 4 timed one by one after the headline)")
    ap.add_argument("--no-parity", action="store_true", help="skip the live fp16-vs-fp32 forward error and the fp32-mode timing")
    ap.add_argument("--fp32-steps", type=int, default=10, help="reverse steps of the exact-fp32 mode timed for parity.fp32_mode_shapes_per_s")
    a = ap.parse_args()
    if a.pos_prec is None:
        a.pos_prec = "split" if a.prec == "fp16" else a.prec
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: spawn the ranks ourselves and pass their exit status on
        import subprocess
        sys.exit(subprocess.call(relaunch_argv(a.gpus, sys.argv[1:]), env=dict(os.environ)))
    # The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints its version banner to the C
    # stdout of every rank, flushed at exit, i.e. after anything printed here): fd 1 is pointed at stderr for the whole
    # run and the JSON line goes to the saved descriptor at the very end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from slide_amd import configs, model_spec
    from slide_amd._lib import check, lib
    from slide_amd.diffusion import EagerChainsSampler, FeatureSampler, JointSampler, OwnGraphSampler, PositionSampler, SplitJointSampler, ThreadedEagerSampler
    from slide_amd.generation import POS_CU_SHARE  # share of the CUs the position chain's stream may use beside the feature chains
    from slide_amd.engine import OP_ATTN_TAIL, OP_GEMM
    from slide_amd.synth import synth_keypoints, synth_state_dict

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks" % (a.gpus, world))
    # SLIDE_BENCH_SHARE_GPU=1 (test knob for 1-GPU boxes): every rank uses device 0 and the collectives run on gloo
    share = os.environ.get("SLIDE_BENCH_SHARE_GPU", "0") != "0"
    if share:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # SLIDE_FORCE_DIST=1: also take the RCCL path (init, barrier, all-gather, all-reduce) in a single-rank run
    use_dist = world > 1 or os.environ.get("SLIDE_FORCE_DIST", "0") != "0"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # bound to this rank's GPU
    B = a.batch

    pc, fc = configs.position_ddpm_config(), configs.feature_ddpm_config()
    sd_p = synth_state_dict(model_spec.denoiser_param_spec(pc["pointnet_config"]))
    sd_f = synth_state_dict(model_spec.denoiser_param_spec(fc["pointnet_config"]))

    def build_arrangement(replay):
        """the chains of this rank and the sampler that steps them, for one replay mode (built once; a second time only when the
        host-enqueue calibration below switches an eager run to graph replay)"""
        a.replay = replay
        P = max(1, min(a.sub_batches, B))
        sizes = [B // P + (1 if i < B % P else 0) for i in range(P)]
        if B % 8 == 0 and B // 8 >= P:  # multiples of 8 samples: the row tiles of a launch then spread evenly over the 8 XCDs
            sizes = [8 * ((B // 8) // P + (1 if i < (B // 8) % P else 0)) for i in range(P)]
        if os.environ.get("SLIDE_SUB_SIZES"):  # experiment knob: explicit sub-batch sizes, e.g. 112,144
            sizes = [int(v) for v in os.environ["SLIDE_SUB_SIZES"].split(",")]
            assert sum(sizes) == B
            P = len(sizes)
        # the feature plan runs as P concurrent sub-batches; the position plan (launch-bound at any size) as ONE chain over
        # the whole batch beside them
        if P == 1:  # one chain pair: the two plans as the two branches of one step graph (two lone streams serialise)
            a.replay = "graph"
        eager = a.replay != "graph"
        if a.workload == "default":
            # position chain over pos_mult x B shapes, stepping once per pos_mult rounds of the feature chains (same shapes per unit time,
            # fewer dependent launches): 1 for the fp16 plan (measured neutral), 2 when the position plan runs wide operands beside fp16
            # feature chains (--pos-prec split: 272 -> 315 shapes/s)
            wide_pos = a.pos_prec in ("split", "fp32") and a.prec == "fp16"
            pos_mult = int(os.environ.get("SLIDE_POS_MULT", "2" if (wide_pos and eager and a.replay == "eager") else "1"))
            pos = PositionSampler(pc["pointnet_config"], sd_p, B * pos_mult, dev, pc["diffusion_config"], prec=a.pos_prec, seed=1000 + rank * 16,
                                  use_graph=not eager, cu_share=POS_CU_SHARE if (a.replay == "eager" and P > 1 and wide_pos) else 0.0)
            # position plan: its own step graph on its own stream ("own", default: 1-1.5 % faster) or a parallel branch of the
            # first feature sub-batch's graph ("branch")
            pos_own = os.environ.get("SLIDE_POS_GRAPH", "own") == "own" and P > 1
            subs = []
            for i, b in enumerate(sizes):
                f_ = FeatureSampler(fc["pointnet_config"], sd_f, b, dev, fc["standard_diffusion_config"], prec=a.prec,
                                    seed=2000 + rank * 16 + i, use_graph=not eager)
                subs.append((f_, JointSampler(pos if (i == 0 and not pos_own) else None, f_), synth_keypoints(b, seed=rank * 16 + i)))
            feat, kp = subs[0][0], subs[0][2]  # sub-batch 0 also serves the roofline leg below
            members = [OwnGraphSampler(s_[0]) if eager else s_[1] for s_ in subs]
            if pos_own:
                members.insert(int(os.environ.get("SLIDE_POS_ORDER", "1")), OwnGraphSampler(pos))
            joint = SplitJointSampler(members)  # one hipGraph per member and step, launched round-robin
            if a.replay == "threads":
                joint = ThreadedEagerSampler([pos] + [s_[0] for s_ in subs])
            elif a.replay == "eager":
                order = [s_[0] for s_ in subs[:1]] + [pos] + [s_[0] for s_ in subs[1:]]
                if os.environ.get("SLIDE_BENCH_ONLY"):  # diagnostic: time a subset of the chains ("pos", "feat", "feat1" = one sub-batch)
                    only = os.environ["SLIDE_BENCH_ONLY"]
                    order = {"pos": [pos], "feat": [s_[0] for s_ in subs], "feat1": [subs[0][0]], "feat2": [s_[0] for s_ in subs[:2]],
                             "pos+feat1": [subs[0][0], pos]}[only]
                share_q = os.environ.get("SLIDE_POS_SHARE")  # diagnostic: the position chain on feature sub-batch <k>'s stream (one queue)
                if share_q is not None and pos in order:
                    pos.stream = subs[int(share_q)][0].stream
                joint = EagerChainsSampler(order, every=[pos_mult if s_ is pos else 1 for s_ in order])
        cat_desc = None
        if a.workload == "five-cat":
            # per-category weight sets (synthetic, keyed on the category id); every segment of this rank's shard is a chain pair
            from slide_amd.generation import CategoryChains
            spec_p, spec_f = model_spec.denoiser_param_spec(pc["pointnet_config"]), model_spec.denoiser_param_spec(fc["pointnet_config"])
            cc = CategoryChains(B * world, rank, world, pc, fc,
                                lambda c: (synth_state_dict(spec_p, seed=100 + c), synth_state_dict(spec_f, seed=200 + c)), dev,
                                prec="mixed" if (a.prec == "fp16" and a.pos_prec == "split") else a.prec, seed=rank + 1)
            cat_desc = [(c, hi - lo) for c, lo, hi, _, _ in cc.chains]
            pos_chains = [(ps, torch.full((hi - lo,), c, dtype=torch.int64, device=dev)) for c, lo, hi, ps, _ in cc.chains]
            feat_chains = [(fs, torch.full((hi - lo,), c, dtype=torch.int64, device=dev),
                            torch.as_tensor(synth_keypoints(hi - lo, seed=rank * 16 + k), device=dev))
                           for k, (c, lo, hi, _, fs) in enumerate(cc.chains)]
            order = []
            for (ps, _), (fs, _, _) in zip(pos_chains, feat_chains):
                order += [fs, ps]
            joint = EagerChainsSampler(order)
            a.replay = "eager"
            feat, kp = feat_chains[0][0], feat_chains[0][2].cpu().numpy()
            sizes = [fs.B for fs, _, _ in feat_chains]
            P = len(sizes)
        else:
            pos_chains = [(pos, torch.zeros(pos.B, dtype=torch.int64, device=dev))]
            feat_chains = [(f_, torch.full((b,), 4, dtype=torch.int64, device=dev), torch.as_tensor(k_, device=dev))
                           for (f_, _, k_), b in zip(subs, sizes)]
        return dict(joint=joint, pos_chains=pos_chains, feat_chains=feat_chains, feat=feat, kp=kp, sizes=sizes, P=P,
                    pos_mult=pos_mult if a.workload == "default" else 1, cat_desc=cat_desc)

    def adopt(arr):
        return tuple(arr[k] for k in ("joint", "pos_chains", "feat_chains", "feat", "kp", "sizes", "P", "pos_mult", "cat_desc"))

    joint, pos_chains, feat_chains, feat, kp, sizes, P, pos_mult, cat_desc = adopt(build_arrangement(a.replay))
    rs = np.random.RandomState(rank)
    # chain starts: x_T is drawn ON THE DEVICE (torch's Philox generator; plumbing) and labels / key points are resident,
    # so that a restart inside the timed region costs a few launches, not a host RNG pass + four uploads (round 1's
    # driver-timed 20-step number carried ~4 ms of that)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)

    def reset():
        for p_, l_ in pos_chains:
            p_.begin(l_, torch.randn(p_.B, 16, 3, device=dev, generator=gen))
        for f_, l_, k_ in feat_chains:
            f_.begin(l_, k_, torch.randn(f_.B, 16, 51, device=dev, generator=gen))

    state = {"left": 0, "enq": 0.0}  # reverse steps left in the current chains; host seconds spent enqueueing launches

    def run(n):
        """n reverse steps of each DDPM.  A chain is 1000 steps; it is restarted from fresh device-side noise when it ends
        (inside the timed region when --steps > 1000 crosses a boundary).  The first chain of the timed region is begun
        BEFORE the clock starts: a generation's inputs are resident when its first step is timed."""
        done = 0
        while done < n:
            if state["left"] == 0:
                reset()
                state["left"] = 1000
            k = min(n - done, state["left"])
            t_enq = time.perf_counter()
            joint.advance(k)
            state["enq"] += time.perf_counter() - t_enq
            state["left"] -= k
            done += k

    def sync_all():
        for p_, _ in pos_chains:
            p_.stream.synchronize()
        for f_, _, _ in feat_chains:
            f_.stream.synchronize()
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier() if share else dist.barrier(device_ids=[local])

    # Host-enqueue calibration (round 6): eager replay needs the host to enqueue ~130 k launches/s per rank; with
    # N launcher processes on one host that may not hold.  A few untimed steps measure the share of the wall time the host spends
    # inside the launch calls (MAX over the ranks).  While the GPU paces the run that share is host cost / GPU time (0.45-0.5 on the
    # boxes measured: 3.2 us per launch, 88 launches, 0.65 ms per step); it approaches 1 when the host does.  Above the limit (0.75;
    # SLIDE_BENCH_AUTO_REPLAY_SHARE -- graph replay costs ~5 % while the GPU paces, so the switch is made only close to the edge) the
    # arrangement is rebuilt for graph replay (one hipGraphLaunch per chain and step).  `config.replay` / `config.replay_auto` report
    # what ran.
    replay_auto = None
    if a.replay == "eager" and a.workload == "default" and os.environ.get("SLIDE_BENCH_AUTO_REPLAY", "1") != "0":
        run(3)
        sync_all()
        state["enq"] = 0.0
        t_c = time.perf_counter()
        run(12)
        enq_c = state["enq"]
        sync_all()
        share_c = enq_c / max(time.perf_counter() - t_c, 1e-9)
        if use_dist:
            tt = torch.tensor([share_c], device=torch.device("cpu") if share else dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            share_c = float(tt.item())
        limit = float(os.environ.get("SLIDE_BENCH_AUTO_REPLAY_SHARE", "0.75"))
        replay_auto = {"host_enqueue_share_max_over_ranks": round(share_c, 3), "limit": limit, "switched_to_graph": share_c > limit}
        if share_c > limit:
            joint = pos_chains = feat_chains = feat = None  # (free the eager arrangement's buffers first)
            joint, pos_chains, feat_chains, feat, kp, sizes, P, pos_mult, cat_desc = adopt(build_arrangement("graph"))
        state["left"] = 0
    # Device priming (round 4, now OPT-IN: SLIDE_BENCH_PRIME=<steps>): a fresh process starts on a cold GPU and the first tens of
    # milliseconds of chain replay run ~5 % slower than steady state.  Round 5 (VERDICT r4 item 8 / ADVICE):
    # the bench does exactly the W warm-up steps the command line asks for; `config.prime_steps` records any opt-in priming.
    prime = int(os.environ.get("SLIDE_BENCH_PRIME", "0"))  # (round 5: opt-in -- the driver's --warmup is the warm-up)
    # Parity is the gate, so it is checked FIRST: the live parity object (forwards of the benched arithmetic against the fp32 mode,
    # complete 1000-step chains) is measured before the warm-up and the timed region, not after them (SLIDE_BENCH_PARITY_FIRST=0:
    # the order of rounds 3-4).  It is other work than the timed steps -- its samplers are built, run and freed inside parity_leg.
    parity_first = os.environ.get("SLIDE_BENCH_PARITY_FIRST", "1") != "0"
    parity_obj = None
    if not a.no_parity and parity_first:  # (every rank checks its own GPU -- and enters the timed region in the same state; rank 0 reports)
        parity_obj = parity_leg(dev, B, a, pc, fc, sd_p, sd_f, gen)
        if rank != 0:
            parity_obj = None
    run(max(a.warmup, 1) + max(prime, 0))  # (one replay call: priming steps, then the W warm-up steps)
    gdev = torch.device("cpu") if share else dev
    gathered = [torch.empty(B, 16, 51, device=gdev) for _ in range(world)] if use_dist else None

    def gather_latents():  # the single collective of the path: all ranks' latents (835 KB / rank at B=256)
        for f_, _, _ in feat_chains:
            f_.stream.synchronize()
        dist.all_gather(gathered, torch.cat([f_.engine.x.reshape(-1, 16, 51) for f_, _, _ in feat_chains], 0).to(gdev))

    if use_dist:
        gather_latents()  # untimed, like the warm-up steps: the first call builds RCCL's channels
    if os.environ.get("SLIDE_BENCH_NO_RESET"):  # diagnostic: the timed region continues the warm-up chains
        pass
    else:
        reset()  # begin the timed chains (x_T, labels, key points, per-chain pre-computes) before the clock starts
        state["left"] = 1000
    sync_all()
    if os.environ.get("SLIDE_BENCH_PRESPIN"):  # diagnostic: keep the GPU busy right up to the start of the timed region
        _w = torch.randn(4096, 4096, device=dev)
        for _ in range(int(os.environ["SLIDE_BENCH_PRESPIN"])):
            _w = _w @ _w * 1e-4
        torch.cuda.synchronize(dev)
    state["enq"] = 0.0
    chain_ev = None
    if os.environ.get("SLIDE_BENCH_CHAIN_ENDS"):  # diagnostic: when each chain retires its last step (stderr)
        streams = [p_.stream for p_, _ in pos_chains] + [f_.stream for f_, _, _ in feat_chains]
        chain_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), st_) for st_ in streams]
        for e0, _, st_ in chain_ev:
            e0.record(st_)
    t0 = time.perf_counter()
    run(a.steps)
    host_enq = state["enq"]
    if chain_ev:
        for _, e1, st_ in chain_ev:
            e1.record(st_)
    if use_dist:
        gather_latents()
    sync_all()
    dt = time.perf_counter() - t0
    if chain_ev:
        print("chain ends (ms after its start event; position chain(s) first): %s; wall %.3f ms" %
              (", ".join("%.3f" % e0.elapsed_time(e1) for e0, e1, _ in chain_ev), dt * 1e3), file=sys.stderr)
    dist_obj = None
    if use_dist:
        tt = torch.tensor([dt], device=gdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # what the process group saw (round 6): its size and backend, every rank's device, every rank's host enqueue
        # time, and that slot r of the all-gather holds rank r's latents (checksums travel as Python objects beside the tensors)
        mine = torch.cat([f_.engine.x.reshape(-1, 16, 51) for f_, _, _ in feat_chains], 0)
        props = torch.cuda.get_device_properties(dev)
        info = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "device": int(torch.cuda.current_device()),
                "device_name": props.name, "device_uuid": str(getattr(props, "uuid", "")),
                "host_enqueue_ms_per_step": round(host_enq * 1e3 / a.steps, 4), "latent_sum": float(mine.double().sum().item())}
        infos = [None] * world
        dist.all_gather_object(infos, info)
        if rank == 0:
            order_ok = all(abs(float(gathered[r].double().sum().item()) - infos[r]["latent_sum"]) <= 1e-6 * (1.0 + abs(infos[r]["latent_sum"]))
                           for r in range(world))
            dist_obj = {"backend": dist.get_backend(), "rccl_ranks": dist.get_world_size(), "gather_order_ok": bool(order_ok),
                        "rank_devices": [[i_["rank"], i_["device"], i_["device_uuid"][-12:]] for i_ in infos],
                        "host_enqueue_ms_per_step_max": max(i_["host_enqueue_ms_per_step"] for i_ in infos),
                        "gathered_shapes": int(sum(g_.shape[0] for g_ in gathered))}
    finite = all(bool(torch.isfinite(p_.state()).all().item()) for p_, _ in pos_chains) and \
        all(bool(torch.isfinite(f_.state()).all().item()) for f_, _, _ in feat_chains)
    ms_per_step = dt * 1e3 / a.steps
    value = world * B / (1000.0 * (ms_per_step / 1e3))

    out = {"metric": "latent shapes/sec (pos+feat DDPM, 1000 steps, 16 pts)", "value": round(value, 3),
           "unit": "shapes/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": {"fp16": "f16 (MFMA operands + activation storage; f32 accumulate, norm statistics, softmax)", "fp32": "f32",
                     "split": "f32 storage, contractions as two-term f16 operand splits (3 x f16 MFMA per product, f32 accumulate)"}[a.prec]
                    + ("" if a.pos_prec == a.prec else "; position DDPM: " + {"fp16": "f16", "fp32": "f32", "split": "two-term f16 operand splits "
                       "(fp32-grade)"}[a.pos_prec]),
           "data": "synthetic",
           "config": {"workload": "BASELINE configs[1]+[2]: airplane position DDPM (16x3) + chair feature DDPM (16x51), "
                                  "batch %d per GPU; 1 step = one reverse step of each; shape = 1000+1000 steps" % B,
                      "batch_per_gpu": B, "sub_batches": sizes, "prec": a.prec, "pos_prec": a.pos_prec, "replay": a.replay,
                      "pos_batch_multiple": pos_mult if a.workload == "default" else 1,
                      "pos_stream_cus": (getattr(pos_chains[0][0], "n_cus", 0) or "all") if pos_chains else None,
                      "launches_per_step": sum(p_.n_launches for p_, _ in pos_chains) + sum(f_.n_launches for f_, _, _ in feat_chains),
                      # host seconds inside the launch calls of the timed region, per step: close to ms_per_step = the host
                      # (or a full hardware queue it is blocked on) paces the run, far below = the GPU does
                      "host_enqueue_ms_per_step": round(host_enq * 1e3 / a.steps, 4),
                      "replay_auto": replay_auto,  # host-enqueue calibration: share measured before the run, whether it switched to graph replay
                      "prime_steps": prime,  # untimed device priming before the --warmup steps (set-up; see bench.py)
                      "finite": finite}}
    if dist_obj is not None:
        out["dist"] = dist_obj
    if cat_desc is not None:
        out["config"]["workload"] = ("BASELINE configs[3]: five-category run (labels 0, 2, 3, 4, 6; one position + one feature weight set "
                                     "per category), %d shapes sharded contiguously over %d GPU(s), %d per GPU; 1 step = one reverse step "
                                     "of every category segment's position and feature chain; latents all-gathered; shape = 1000+1000 steps"
                                     % (B * world, world, B))
        out["config"]["segments_rank0"] = [{"label": c, "shapes": n} for c, n in cat_desc]

    if rank == 0 and not a.no_roofline:
        L = lib()
        f = feat
        with torch.cuda.stream(f.stream):
            n = len(f.step_ops)
            ms = (ctypes.c_float * n)()
            tot = np.zeros(n)
            reps = 5
            b0 = sizes[0]
            feat.begin(np.full(b0, 4, np.int64), kp, rs.standard_normal((b0, 16, 51)).astype(np.float32))
            for _ in range(reps):
                check(L.slide_run_ops_timed(f.step_ops, n, ctypes.c_void_p(f.stream.cuda_stream), ms), "run_ops_timed")
                tot += np.array(list(ms))
        tot /= reps
        flops = f.gemm_flops  # per MFMA launch of the step plan, algorithmic (logical channels), whole sub-batch

        def kernel_of(o):
            """rocprofv3 name of the kernel run_gemm / run_attn_tail (engine.hip) launches for this op (default knobs)"""
            b = lambda v: "true" if v else "false"
            if o.kind == OP_ATTN_TAIL:  # (run_attn_tail: the register-X form unless SLIDE_TAIL_RX=0 or the values' chunk count is odd)
                rx = os.environ.get("SLIDE_TAIL_RX", "1") != "0" and (o.i[4] // 32) % 4 == 0
                if rx:  # (second template argument: fragment-major u / mo, f[1] bit 4)
                    return "attn_tail_rx_kernel<%d, %s>" % (o.i[6], b(int(o.f[1]) & 16))
                return "attn_tail_kernel<%d>" % o.i[6]
            if o.kind != OP_GEMM:
                return None
            rows, n_cob, npxl, cbw = o.i[0], o.i[3], o.i[4], o.i[7]
            grid = ((rows + 63) // 64) * ((n_cob + 1) // 2)
            if a.prec == "fp16" and npxl == 4 and grid <= (1024 if o.p[6] else 8192):
                return "gemm_small_kernel<2, %s>" % b(o.p[3])
            if a.prec in ("fp32", "split") or not o.i[8]:
                return "gemm_kernel<%d, %d, %d>" % ({"fp32": 0, "fp16": 1, "split": 2}[a.prec], npxl, cbw)
            if cbw == 2 and not (o.p[3] and o.p[8]):
                return "gemm_glds_occ3_kernel<%d, %s, %s>" % (npxl, b(o.p[3]), b(o.p[8]))  # three workgroups per CU
            return "gemm_glds_kernel<%d, %d, 3, 32, %s, %s>" % (npxl, cbw, b(o.p[3]), b(o.p[8]))

        # dominant kernel = the MFMA kernel with the largest share of the feature denoiser's step time, measured here
        by_k = {}
        for i in range(n):
            k = f.kernel_names.get(i) or kernel_of(f.step_ops[i])  # (round-3 ops name their kernel at plan time)
            if k is not None and i in flops:
                e = by_k.setdefault(k, [0.0, 0.0, 0])
                e[0] += tot[i]; e[1] += flops[i]; e[2] += 1
        kname = max(by_k, key=lambda k: by_k[k][0])
        dms, dflops, dn = by_k[kname]
        ach = dflops / (dms * 1e-3) / 1e12
        mfma_ms = sum(e[0] for e in by_k.values())
        mfma_fl = sum(e[1] for e in by_k.values())
        traffic, tsrc = None, None
        pj = os.path.join(REPO, "profiles", "hbm_pmc_latest.json")  # written by tools/rocprof_summarize.py from PMC passes
        if os.path.exists(pj):
            pm = json.load(open(pj))
            # only a PMC pass collected at the SAME samples per launch as the kernel timed here is comparable
            if kname in pm.get("kernels", {}) and pm.get("samples_per_launch") == sizes[0]:
                traffic = pm["kernels"][kname]["hbm_bytes_per_launch"]
                tsrc = pm["source"]
        out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_TFLOPS[a.prec], "unit": "TFLOP/s",
                           "frac": round(ach / PEAK_TFLOPS[a.prec], 4), "traffic": traffic, "traffic_source": tsrc,
                           "kernel": "%s (%s MFMA; feature denoiser, %d samples/launch: %d launches/step, avg %.1f us, "
                                     "%.1f%% of the step's kernel time)" % (kname, a.prec, sizes[0], dn, 1e3 * dms / dn,
                                                                          100 * dms / tot.sum()),
                           "flops_per_launch_avg": dflops / dn,
                           "step_ms_eager_sum": round(float(tot.sum()), 4),
                           "mfma_kernels": {k: {"launches": e[2], "ms": round(e[0], 4),
                                                "tflops": round(e[1] / (e[0] * 1e-3) / 1e12, 1)}
                                            for k, e in sorted(by_k.items(), key=lambda kv: -kv[1][0])},
                           "mfma_all": {"ms": round(float(mfma_ms), 4),
                                        "tflops": round(float(mfma_fl / (mfma_ms * 1e-3) / 1e12), 1)}}
        # whole-path fraction on the driver's record: the reference's algorithmic conv / linear FLOPs of one joint step of the
        # batch (77 954 048 + 1 057 054 784 per sample) / the timed step / the dense MFMA peak
        step_fl = B * (77954048 + 1057054784)
        out["roofline"]["step"] = {"algorithmic_gflop_per_gpu_step": round(step_fl / 1e9, 1),
                                   "achieved_tflops": round(step_fl / (ms_per_step * 1e-3) / 1e12, 1),
                                   "frac": round(step_fl / (ms_per_step * 1e-3) / 1e12 / PEAK_TFLOPS[a.prec], 4)}
    if rank == 0 and not a.no_parity:
        out["parity"] = parity_obj if parity_obj is not None else parity_leg(dev, B, a, pc, fc, sd_p, sd_f, gen)
        out["parity"]["measured"] = "before the timed region" if parity_obj is not None else "after the timed region"
    if rank == 0 and world == 1 and not a.no_configs and a.workload == "default" and a.prec == "fp16":
        try:
            out["configs"] = configs_leg(dev, B, a, pc, fc, sd_p, sd_f, gen, feat_chains, pos_chains[0][0])
            out["configs"]["config5_decode"] = "see `decode` (BASELINE configs[4]: the CLIs' default arithmetic)"
        except Exception as ex:  # (a reporting leg must not take the headline down with it)
            out["configs"] = {"error": repr(ex)[:300]}
    if rank == 0 and not a.no_decode and a.workload == "default":
        out["decode"] = decode_leg(dev, B)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
