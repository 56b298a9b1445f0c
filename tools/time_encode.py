"""throughput of PointAutoencoder.encode (HIP module path): B clouds (B,2048,6) + key points -> (B,16,48) latents"""
import json, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "pointnet2"))
import torch
from models.autoencoder import PointAutoencoder
from slide_amd import _ext as E
from slide_amd.synth import synth_state_dict
g = np.load(os.path.join(REPO, "tests", "golden", "golden_encode.npz"))
enc, decs = json.loads(str(g["encoder_config_json"])), json.loads(str(g["decoder_configs_json"]))
spec = [(str(n), tuple(int(x) for x in str(s).split(","))) for n, s in zip(g["spec_names"], g["spec_shapes"])]
vals = synth_state_dict([("ae." + n, s) for n, s in spec])
dev = torch.device("cuda:0")
ae = PointAutoencoder(enc, decs, True)
ae.load_state_dict({n: torch.from_numpy(vals["ae." + n]) for n, _ in spec}, strict=False); ae = ae.to(dev).eval()
B = int(os.environ.get("B", 256))
u = torch.randn(B, 2048, 3, device=dev); u = u / u.norm(dim=2, keepdim=True)
pc = torch.cat([u * 0.7, u], dim=2).contiguous()
kidx = E.furthest_point_sampling(pc[:, :, :3].contiguous(), 16).long()
kp = torch.gather(pc[:, :, :3], 1, kidx.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
lab = torch.zeros(B, dtype=torch.long, device=dev)
for _ in range(2):
    out = ae.encode(pc, kp, label=lab, sample_posterior=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 3
for _ in range(n):
    out = ae.encode(pc, kp, label=lab, sample_posterior=False)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print("encode B=%d: %.1f ms  -> %.0f shapes/s  out %s finite %s" % (B, dt * 1e3, B / dt, tuple(out.shape), bool(torch.isfinite(out).all())))
