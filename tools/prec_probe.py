"""Where does the fp16 mode's single-forward error come from?  

For each denoiser (pos / feat), at batch 64 and t spread over the schedule:
  * relative L2 of the fp16 plan vs the exact-fp32 plan, for the output and for every per-point level (sa0, sa1, fp1, fp0);
  * the same against an fp32 plan that runs on fp16-ROUNDED weights (isolates activation rounding from weight rounding);
  * plan variants selected by environment knobs (given on the command line as NAME=VALUE groups separated by '/').
Usage: python tools/prec_probe.py [--nets pos,feat] [--variants "SLIDE_GX=0/SLIDE_BODY=0 SLIDE_SA_CHAIN=0"]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def levels_of(eng):
    out = {}
    for k, (t, c) in eng.levels.items():
        out[k] = t.float().cpu().numpy().reshape(t.shape[0], -1)[:, :c].copy()
    return out


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b.astype(np.float64)), 1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nets", default="pos,feat")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--variants", default="")
    ap.add_argument("--input", default="normal", help="normal | keypoints (pos: x = synthetic key points) | scaled:<s> (x = s * normal)")
    ap.add_argument("--t", default="", help="fixed timestep for every sample (default: spread over 0..999)")
    a = ap.parse_args()
    import torch
    from slide_amd import configs, model_spec
    from slide_amd.engine import DenoiserEngine
    from slide_amd.synth import synth_keypoints, synth_state_dict
    dev = torch.device("cuda", 0)
    B = a.batch
    rs = np.random.RandomState(0)
    for nm in a.nets.split(","):
        cfg = configs.position_ddpm_config() if nm == "pos" else configs.feature_ddpm_config()
        hp = cfg["pointnet_config"]
        sd = synth_state_dict(model_spec.denoiser_param_spec(hp))
        sd_r = {k: (np.asarray(v, np.float32).astype(np.float16).astype(np.float32) if (k.endswith("weight") and np.asarray(v).ndim >= 2
                                                                                          and "class_emb" not in k and "fc_t" not in k
                                                                                          and not k.endswith(".fc.weight")
                                                                                          and not k.endswith(".fc_condition.weight")) else v)
                for k, v in sd.items()}
        x = rs.standard_normal((B, 16, 3 + hp["in_fea_dim"])).astype(np.float32)
        if nm == "feat" or a.input == "keypoints":
            x[:, :, :3] = synth_keypoints(B, seed=99)
        if a.input.startswith("scaled:"):
            x *= float(a.input.split(":")[1])
            if nm == "feat":
                x[:, :, :3] = synth_keypoints(B, seed=99)
        ts = np.linspace(0, 999, B).astype(np.float32) if not a.t else np.full(B, float(a.t), np.float32)
        print("[%s] input %s t %s: |x| rms %.3f" % (nm, a.input, a.t or "spread", float(np.sqrt((x ** 2).mean()))))
        lab = np.full(B, 4 if nm == "feat" else 0, np.int64)
        e32 = DenoiserEngine(hp, sd, B, dev, prec="fp32")
        y32 = e32.forward(x, ts, lab).double().cpu().numpy()
        l32 = levels_of(e32)
        e32r = DenoiserEngine(hp, sd_r, B, dev, prec="fp32")
        y32r = e32r.forward(x, ts, lab).double().cpu().numpy()
        l32r = levels_of(e32r)
        print("[%s] fp32(rounded weights) vs fp32: out %.3e  " % (nm, rel(y32r, y32)) +
              "  ".join("%s %.3e" % (k, rel(l32r[k], l32[k])) for k in l32))
        variants = [""] + [v for v in a.variants.split("/") if v]
        for var in variants:
            saved = {}
            for kv in var.split():
                k, v = kv.split("=")
                saved[k] = os.environ.get(k)
                os.environ[k] = v
            try:
                e16 = DenoiserEngine(hp, sd, B, dev, prec="fp16")
                y16 = e16.forward(x, ts, lab).double().cpu().numpy()
                l16 = levels_of(e16)
                per_s = np.linalg.norm((y16 - y32).reshape(B, -1), axis=1) / np.linalg.norm(y32.reshape(B, -1), axis=1)
                print("[%s] |y32| rms %.4f; fp16 %-40s vs fp32: out %.3e (per-sample max %.3e, median %.3e; max-norm %.3e)  " % (
                    nm, float(np.sqrt((y32 ** 2).mean())), var or "(default)", rel(y16, y32), per_s.max(), np.median(per_s),
                    np.abs(y16 - y32).max() / np.abs(y32).max()) +
                      "  ".join("%s %.3e" % (k, rel(l16[k], l32[k])) for k in l32))
                print("[%s] fp16 %-40s vs fp32(rounded W): out %.3e  " % (nm, var or "(default)", rel(y16, y32r)) +
                      "  ".join("%s %.3e" % (k, rel(l16[k], l32r[k])) for k in l32))
                n_l = len([o for o in e16.ops if o is not None])
                print("[%s]      launches %d" % (nm, n_l))
            finally:
                for k, v in saved.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v


if __name__ == "__main__":
    main()
