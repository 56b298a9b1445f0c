"""CPU emulation (tools/prec_emul.py's operand-rounding model on the numpy oracle): which layers of the FEATURE denoiser must run wide
operands for the fp16 plan's forward to meet 1e-3 in the MAX-norm on the reference goldens?  (round 6)"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import denoiser_np as D
from slide_amd.synth import synth_state_dict
F32 = np.float32
h = lambda a: a.astype(np.float16).astype(F32)
orig = D.conv1x1
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "golden_denoiser_feat.npz"))
hp = json.loads(str(g["config_json"]))
spec = [(str(n), tuple(int(x) for x in str(s).split(",")) if str(s) else ()) for n, s in zip(g["spec_names"], g["spec_shapes"])]
sd = synth_state_dict(spec)
name_of = {id(v): k for k, v in sd.items()}
SEL = [None]

def conv(x, w, b=None):
    k = name_of.get(id(w), "?")
    if SEL[0] is None:
        return orig(x, w, b)
    wide = any(p in k for p in SEL[0])
    w2 = w.reshape(w.shape[0], -1).astype(F32)
    B, C = x.shape[:2]
    X = x.reshape(B, C, -1).astype(F32)
    mm = lambda W_, X_: np.matmul(W_[None].astype(np.float64), X_.astype(np.float64))
    y = mm(w2, X) if wide else mm(h(w2), h(X))
    y = y.astype(F32)
    if b is not None:
        y = y + b.astype(F32)[None, :, None]
    return y.reshape((B, w2.shape[0]) + x.shape[2:]).astype(F32)

D.conv1x1 = conv
sets = {"FP0mlp2+head": ("FP_modules.0.mlp2.", "fc_lyaer"), "FP1mlp2": ("FP_modules.1.mlp2.",), "FP0mlp2+head+FPattn": ("FP_modules.0.mlp2.", "fc_lyaer", "FP_modules.0.attention", "FP_modules.1.attention"), "FP0mlp2+head+FPmlp1": ("FP_modules.0.mlp2.", "fc_lyaer", "FP_modules.0.mlp1", "FP_modules.1.mlp1"), "FP0all+head": ("FP_modules.0", "fc_lyaer"), "none": (), "all": ("",), "head": ("fc_lyaer",), "mlp2+head": (".mlp2.", "fc_lyaer"), "FP0": ("FP_modules.0",), "FP0+head": ("FP_modules.0", "fc_lyaer"),
        "FP+head": ("FP_modules", "fc_lyaer"), "SA": ("SA_modules",), "attn": ("attention",), "w5": ("weight_conv.5",), "w5+mlp2+head": ("weight_conv.5", ".mlp2.", "fc_lyaer"),
        "feat_out": ("feat_out_conv",), "res": ("res_connect",), "res+head": ("res_connect", "fc_lyaer"), "first+res": ("first_mlp", "res_connect", "grouped_feat_conv", "feat_conv")}
keys = [k for k in sys.argv[1:] if not k.startswith("only=")] or ["t0", "t500", "mixed"]
only = [k[5:].split(",") for k in sys.argv[1:] if k.startswith("only=")]
if only:
    sets = {k: v for k, v in sets.items() if k in only[0]}
for k in keys:
    x, ts, lab, ref = g["x_" + k], g["ts_" + k], g["label_" + k], g["eps_" + k]
    res = []
    for nm, s_ in sets.items():
        SEL[0] = s_
        y = D.denoiser_forward(hp, sd, x, ts, lab)
        l2 = float(np.linalg.norm(y - ref) / np.linalg.norm(ref)); mx = float(np.abs(y - ref).max() / np.abs(ref).max())
        res.append("%s L2 %.2e max %.2e" % (nm, l2, mx))
        print("[%s] %s" % (k, res[-1]), flush=True)
