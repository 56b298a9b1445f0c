"""throughput of PointAutoencoder.decode (HIP module path) on synthetic latents: B shapes -> (B,2048,6)"""
import json, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "pointnet2"))
import torch
from models.autoencoder import PointAutoencoder
from slide_amd.synth import synth_keypoints, synth_state_dict
g = np.load(os.path.join(REPO, "tests", "golden", "golden_decode.npz"))
decs = json.loads(str(g["decoder_configs_json"]))
spec = [(str(n), tuple(int(x) for x in str(s).split(","))) for n, s in zip(g["spec_names"], g["spec_shapes"])]
vals = synth_state_dict([("ae." + n, s) for n, s in spec])
dev = torch.device("cuda:0")
ae = PointAutoencoder(None, decs, True)
ae.load_state_dict({n: torch.from_numpy(vals["ae." + n]) for n, _ in spec}); ae = ae.to(dev).eval()
B = int(os.environ.get("B", 256))
kp = torch.from_numpy(synth_keypoints(B)).to(dev); feat = 0.5 * torch.randn(B, 16, 48, device=dev); lab = torch.zeros(B, dtype=torch.long, device=dev)
for _ in range(2):
    out = ae.decode(kp, feat, label=lab)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 3
for _ in range(n):
    out = ae.decode(kp, feat, label=lab)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print("decode B=%d: %.1f ms  -> %.0f shapes/s  out %s finite %s" % (B, dt * 1e3, B / dt, tuple(out.shape), bool(torch.isfinite(out).all())))
