"""host-side profile of STEADY-STATE decode passes (cProfile over three passes after two warm-up passes): where the Python program
spends its time between launches.  usage: python tools/ab/decode_hostprof.py"""
import cProfile, io, json, os, pstats, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "pointnet2"))
os.environ.setdefault("SLIDE_MODULE_PREC", "fp16")
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
for _ in range(2):
    out = ae.decode(kp, feat, label=lab)
torch.cuda.synchronize()
pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
for _ in range(3):
    out = ae.decode(kp, feat, label=lab)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize(); pr.disable(); dt = time.perf_counter() - t0
print("3 passes: host enqueue %.1f ms, wall %.1f ms (%.1f ms per pass)" % (t_enq * 1e3, dt * 1e3, dt / 3 * 1e3))
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30)
print(s.getvalue()[:7000])
