"""is PointAutoencoder.decode of a sub-batch bit-identical to the same rows of a full-batch decode (single stream)?  And is a two-stream
decode deterministic?  (authoring tool)"""
import json, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
B = 256
kp = torch.from_numpy(synth_keypoints(B)).to(dev); feat = 0.5 * torch.randn(B, 16, 48, device=dev); lab = torch.zeros(B, dtype=torch.long, device=dev)
start = torch.zeros(B, dtype=torch.int32, device=dev)
ref = ae.decode(kp, feat, label=lab, fps_start_idx=start)
ref2 = ae.decode(kp, feat, label=lab, fps_start_idx=start)
print("full batch twice equal:", bool(torch.equal(ref, ref2)))
for n in (128, 86, 64, 32):
    o = ae.decode(kp[:n], feat[:n], label=lab[:n], fps_start_idx=start[:n])
    d = (o - ref[:n]).abs().max().item()
    bad = (o != ref[:n]).flatten(1).any(1).sum().item()
    print("first %d alone (one stream) == rows of the full batch: %s  max |diff| %.3e  shapes differing %d" % (n, bool(torch.equal(o, ref[:n])), d, bad))
o = ae.decode(kp[128:], feat[128:], label=lab[128:], fps_start_idx=start[128:])
print("second 128 alone == rows:", bool(torch.equal(o, ref[128:])), (o - ref[128:]).abs().max().item())
