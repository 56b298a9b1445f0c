"""One chain under rocprofv3 --kernel-trace: per-launch durations AND the gaps between consecutive kernels of a step.
on the GPU box:
  cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ct -o ct -- \
      python $GRAFT_REPO_ROOT/tools/ab/chain_trace.py run feat 256
  python tools/ab/chain_trace.py report gpurun_out/ct [n_ops]
"""
import csv, glob, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def short(n):
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:60]


if sys.argv[1] == "run":
    import numpy as np
    import torch
    from slide_amd import configs, model_spec
    from slide_amd.diffusion import FeatureSampler, PositionSampler
    from slide_amd.synth import synth_keypoints, synth_state_dict
    which, B = sys.argv[2], int(sys.argv[3])
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(0)
    if which == "feat":
        c = configs.feature_ddpm_config()
        s = FeatureSampler(c["pointnet_config"], synth_state_dict(model_spec.denoiser_param_spec(c["pointnet_config"])), B, dev,
                           c["standard_diffusion_config"], prec="fp16", use_graph=False)
        s.begin(np.full(B, 4, np.int64), synth_keypoints(B), rs.standard_normal((B, 16, 51)).astype(np.float32))
    else:
        c = configs.position_ddpm_config()
        s = PositionSampler(c["pointnet_config"], synth_state_dict(model_spec.denoiser_param_spec(c["pointnet_config"])), B, dev,
                            c["diffusion_config"], prec="fp16", use_graph=False)
        s.begin(np.zeros(B, np.int64), rs.standard_normal((B, 16, 3)).astype(np.float32))
    s.advance(30)
    s.stream.synchronize(); torch.cuda.synchronize()
    print("n_ops", len(s.step_ops))
else:
    d = sys.argv[2]
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    n_ops = int(sys.argv[3]) if len(sys.argv) > 3 else None
    names = [short(r["Kernel_Name"]) for r in rows]
    if n_ops is None:  # period = distance between the last two update kernels
        idx = [i for i, n in enumerate(names) if n.startswith("update_")]
        n_ops = idx[-1] - idx[-2]
    last = rows[-2 * n_ops:-n_ops]  # the step before the last one
    t0 = int(last[0]["Start_Timestamp"])
    prev_end = None
    tot_k = tot_g = 0.0
    for r in last:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = 0.0 if prev_end is None else (st - prev_end) / 1e3
        dur = (en - st) / 1e3
        tot_k += dur; tot_g += max(gap, 0)
        print("%8.1f  dur %7.1f  gap %6.1f  grid %7s  %s" % ((st - t0) / 1e3, dur, gap, r.get("Grid_Size", "?"), short(r["Kernel_Name"])))
        prev_end = en
    nxt = int(rows[-n_ops]["Start_Timestamp"])
    print("step period %.1f us: kernels %.1f + gaps %.1f (n_ops %d)" % ((nxt - t0) / 1e3, tot_k, tot_g, n_ops))
