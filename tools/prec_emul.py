"""CPU emulation of operand rounding in the denoiser (numpy oracle with conv1x1 monkeypatched): which roundings matter?
modes: f16 = round X and W of every 1x1 convolution to fp16; w = W only; x = X only;
       xc = X rounded after removing its per-sample per-channel mean over positions (common mode kept exact), W rounded;
       split = both operands as two-term fp16 splits, three products (hi*hi + hi*lo + lo*hi)."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import denoiser_np as D
from slide_amd import configs, model_spec
from slide_amd.synth import synth_keypoints, synth_state_dict
F32 = np.float32
h = lambda a: a.astype(np.float16).astype(F32)
orig = D.conv1x1
MODE = [None]

def conv(x, w, b=None):
    m = MODE[0]
    if m is None:
        return orig(x, w, b)
    w2 = w.reshape(w.shape[0], -1).astype(F32)
    B, C = x.shape[:2]
    X = x.reshape(B, C, -1).astype(F32)
    mm = lambda W_, X_: np.matmul(W_[None].astype(np.float64), X_.astype(np.float64))
    if m == "f16":
        y = mm(h(w2), h(X))
    elif m == "w":
        y = mm(h(w2), X)
    elif m == "x":
        y = mm(w2, h(X))
    elif m == "xc":
        mu = X.mean(axis=2, keepdims=True)
        y = mm(h(w2), h(X - mu)) + mm(w2, mu)
    elif m == "xc2":  # common mode through fp16 weights too, but X's common mode exact
        mu = X.mean(axis=2, keepdims=True)
        y = mm(h(w2), h(X - mu)) + mm(h(w2), mu)
    elif m == "split":
        wh, xh = h(w2), h(X)
        wl, xl = h(w2 - wh), h(X - xh)
        y = mm(wh, xh) + mm(wh, xl) + mm(wl, xh)
    else:
        raise ValueError(m)
    y = y.astype(F32)
    if b is not None:
        y = y + b.astype(F32)[None, :, None]
    return y.reshape((B, w2.shape[0]) + x.shape[2:]).astype(F32)

D.conv1x1 = conv
rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b.astype(np.float64)))
B = 8
rs = np.random.RandomState(0)
for nm in sys.argv[1:] or ["pos"]:
    cfg = configs.position_ddpm_config() if nm == "pos" else configs.feature_ddpm_config()
    hp = cfg["pointnet_config"]
    sd = synth_state_dict(model_spec.denoiser_param_spec(hp))
    for inp in ("normal", "keypoints", "scaled0.3"):
        x = rs.standard_normal((B, 16, 3 + hp["in_fea_dim"])).astype(F32)
        if nm == "feat" or inp == "keypoints":
            x[:, :, :3] = synth_keypoints(B, seed=99)
        if inp == "scaled0.3":
            x *= 0.3
            if nm == "feat":
                x[:, :, :3] = synth_keypoints(B, seed=99)
        ts = np.linspace(0, 999, B).astype(F32)
        lab = np.full(B, 4 if nm == "feat" else 0, np.int64)
        MODE[0] = None
        y0 = D.denoiser_forward(hp, sd, x, ts, lab)
        out = []
        for m in ("f16", "w", "x", "xc", "xc2", "split"):
            MODE[0] = m
            out.append("%s %.2e" % (m, rel(D.denoiser_forward(hp, sd, x, ts, lab), y0)))
        print("[%s] %-10s " % (nm, inp) + "  ".join(out), flush=True)

# ---- selective split: which modules need the wide operands?  (pos net)
if os.environ.get("SELECT"):
    cfg = configs.position_ddpm_config(); hp = cfg["pointnet_config"]
    sd = synth_state_dict(model_spec.denoiser_param_spec(hp))
    name_of = {id(v): k for k, v in sd.items()}
    SEL = [()]
    def conv_sel(x, w, b=None):
        k = name_of.get(id(w), "?")
        MODE[0] = "split" if any(k.startswith(p) or p in k for p in SEL[0]) else "f16"
        return conv(x, w, b)
    D.conv1x1 = conv_sel
    sets = {"none": (), "all": ("",), "FP": ("FP_modules",), "FP+head": ("FP_modules", "fc_lyaer"), "SA": ("SA_modules",),
            "FP0+head": ("FP_modules.0", "fc_lyaer"), "FP1": ("FP_modules.1",), "mlp2+head": (".mlp2.", "fc_lyaer"),
            "attn": ("attention",), "mlp": (".mlps.", ".mlp1.", ".mlp2."), "FPattn": ("FP_modules.0.attention", "FP_modules.1.attention"),
            "FPmlp1": ("FP_modules.0.mlp1", "FP_modules.1.mlp1"), "head": ("fc_lyaer",)}
    for inp in ("keypoints", "scaled0.3"):
        rs = np.random.RandomState(0)
        x = rs.standard_normal((B, 16, 3)).astype(F32)
        if inp == "keypoints":
            x = synth_keypoints(B, seed=99).astype(F32)
        else:
            x *= 0.3
        ts = np.linspace(0, 999, B).astype(F32); lab = np.zeros(B, np.int64)
        D.conv1x1 = orig
        y0 = D.denoiser_forward(hp, sd, x, ts, lab)
        D.conv1x1 = conv_sel
        res = []
        for nm_, s_ in sets.items():
            SEL[0] = s_
            res.append("%s %.2e" % (nm_, rel(D.denoiser_forward(hp, sd, x, ts, lab), y0)))
        print("[pos select] %-10s " % inp + "  ".join(res), flush=True)
