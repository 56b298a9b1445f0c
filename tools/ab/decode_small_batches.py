"""PointAutoencoder.decode of 7 shapes at once against the same shapes as batches of 3, 3, 1 (the CLI test's decomposition), per
module precision: bit-identical?  (authoring tool: SLIDE_HIP_LIB selects the library build)"""
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
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    torch.manual_seed(seed)
    B = 7
    kp = torch.from_numpy(synth_keypoints(B, seed=seed) if "seed" in synth_keypoints.__code__.co_varnames else synth_keypoints(B)).to(dev)
    kp = kp + 0.01 * torch.randn_like(kp)
    feat = 0.5 * torch.randn(B, 16, 48, device=dev); lab = torch.zeros(B, dtype=torch.long, device=dev)
    start = torch.zeros(B, dtype=torch.int32, device=dev)
    ref = ae.decode(kp, feat, label=lab, fps_start_idx=start)
    parts = torch.cat([ae.decode(kp[a:b], feat[a:b], label=lab[a:b], fps_start_idx=start[a:b]) for a, b in ((0, 3), (3, 6), (6, 7))])
    d = (parts - ref).abs().flatten(1).max(1).values.cpu().numpy()
    print("seed %d: equal=%s per-shape max |diff| %s" % (seed, bool(torch.equal(parts, ref)), np.array2string(d, precision=2)))
