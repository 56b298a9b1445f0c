"""Per-op device time of one feature-/position-denoiser step (instrumented eager replay, HIP events).
usage: python tools/profile_ops.py [--prec fp16] [--batch 256] [--which feat|pos]"""
import argparse, ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slide_amd import configs, model_spec
from slide_amd._lib import check, lib
from slide_amd.diffusion import FeatureSampler, PositionSampler
from slide_amd.synth import synth_keypoints, synth_state_dict

ap = argparse.ArgumentParser()
ap.add_argument("--prec", default="fp16"); ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--which", default="feat"); ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0"); B = a.batch
rs = np.random.RandomState(0)
if a.which == "feat":
    c = configs.feature_ddpm_config()
    s = FeatureSampler(c["pointnet_config"], synth_state_dict(model_spec.denoiser_param_spec(c["pointnet_config"])), B, dev,
                       c["standard_diffusion_config"], prec=a.prec)
    s.begin(np.full(B, 4, np.int64), synth_keypoints(B), rs.standard_normal((B, 16, 51)).astype(np.float32))
else:
    c = configs.position_ddpm_config()
    s = PositionSampler(c["pointnet_config"], synth_state_dict(model_spec.denoiser_param_spec(c["pointnet_config"])), B, dev,
                        c["diffusion_config"], prec=a.prec)
    s.begin(np.zeros(B, np.int64), rs.standard_normal((B, 16, 3)).astype(np.float32))
n = len(s.step_ops); ms = (ctypes.c_float * n)(); tot = np.zeros(n)
names = {16: "ATTN_TAIL", 1: "GEMM", 2: "PREP", 3: "ASM_SA", 4: "ASM_FP", 5: "FIN_GN", 6: "ATTN", 7: "COPY", 8: "TEMB", 9: "COND", 10: "UPD_P", 11: "UPD_F", 12: "ADV_T"}
with torch.cuda.stream(s.stream):
    for r in range(a.reps + 1):
        check(lib().slide_run_ops_timed(s.step_ops, n, ctypes.c_void_p(s.stream.cuda_stream), ms), "timed")
        if r: tot += np.array(list(ms))
tot /= a.reps
print("%3s %-7s %8s %7s %6s %6s %6s %8s %8s %8s %8s %8s" % ("#", "kind", "us", "rows", "K", "N", "npx", "GFLOP", "TFLOP/s", "rdMB", "wrMB", "GB/s"))
for i in range(n):
    o = s.step_ops[i]
    if o.kind == 1:
        fl = s.gemm_flops[i]
        rd, wr = s.gemm_bytes[i]
        print("%3d %-7s %8.1f %7d %6d %6d %6d %8.2f %8.1f %8.1f %8.1f %8.0f" % (i, "GEMM", tot[i] * 1e3, o.i[0], o.i[2], o.i[3] * 32, 1 << o.i[4], fl / 1e9, fl / (tot[i] * 1e-3) / 1e12, rd / 1e6, wr / 1e6, (rd + wr) / (tot[i] * 1e-3) / 1e9))
    else:
        kn = getattr(s, "kernel_names", {}).get(i, "")
        print("%3d %-7s %8.1f   %s  i=%s" % (i, names.get(o.kind, "k%d" % o.kind), tot[i] * 1e3, kn, list(o.i)[:6]))
print("total us %.1f  gemm us %.1f" % (tot.sum() * 1e3, sum(tot[i] for i in range(n) if s.step_ops[i].kind == 1) * 1e3))
