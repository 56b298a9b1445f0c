"""GPU debugging aid for the LDS-resident kernel: runs the op program truncated after k ops for k = 1 .. n on the GPU (every
workgroup dumps its LDS arena) and in the numpy emulator (tests/resident_emu.py), and reports the first op whose arena
differs.  usage: python tools/ab/debug_resident.py [first_k]"""
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import golden_spec, load_golden  # noqa: E402
from resident_emu import Emu  # noqa: E402
from slide_amd.experiments import resident as R  # noqa: E402
from slide_amd.synth import synth_state_dict  # noqa: E402

g = load_golden("golden_denoiser_pos.npz")
hp = json.loads(str(g["config_json"]))
sd = synth_state_dict(golden_spec(g))
dev = torch.device("cuda:0")
B = 3
den = R.ResidentDenoiser(hp, sd, B, dev)
plan, e = den.plan, den.engine
emu = Emu(plan)
key = "mixed"
x, ts, label, ref = g["x_" + key], g["ts_" + key], g["label_" + key], g["eps_" + key]
eps, _ = den.forward(x, ts, label, dbg=True)
torch.cuda.synchronize()
print("full forward rel err", float(np.abs(eps.cpu().numpy() - ref).max() / np.abs(ref).max()))
tv, cv = e.tvec.cpu().numpy(), e.cvec.cpu().numpy()
names = {1: "PREP", 2: "ASSEMBLE", 3: "GEMM", 4: "FINALIZE", 5: "AFFINE", 6: "TAIL", 7: "ZFILL"}
bad = 0
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1, len(plan.ops) + 1):
    d = torch.zeros(B, plan.lds_bytes, dtype=torch.uint8, device=dev)
    a = plan.args(1, e.x, eps_out=den.eps, per_sample_t=True, dbg=d)
    a.n_ops = k
    R._run(a, torch.cuda.current_stream())
    torch.cuda.synchronize()
    got = d.cpu().numpy()
    worst = 0.0
    for b in range(B):
        emu.run(x[b], tv[b], cv[b], n_ops=k)
        want = emu.lds
        w16 = want.view(np.uint16)
        valid = w16 != 0x7E00
        gh, wh = got[b].view(np.float16).astype(np.float32), want.view(np.float16).astype(np.float32)
        # fp32 areas (everything below the first feature table) are compared as floats
        lo = plan.names["feat0"].off
        gf, wf = got[b][:lo].view(np.float32), want[:lo].view(np.float32)
        vf = ~np.isnan(wf)
        vf[plan.knn // 4:plan.knn // 4 + 64] = False
        ef = np.abs(gf[vf] - wf[vf]) / (1e-3 + np.abs(wf[vf]))
        ih = np.arange(len(wh)) >= lo // 2
        m = valid & ih
        eh = np.abs(gh[m] - wh[m]) / (0.05 + np.abs(wh[m]))
        knn_ok = np.array_equal(got[b][plan.knn:plan.knn + 256], want[plan.knn:plan.knn + 256])
        worst = max(worst, float(ef.max()) if ef.size else 0, float(eh.max()) if eh.size else 0, 0 if knn_ok else 99)
        if eh.size and eh.max() > 0.05 and bad < 3:
            idx = np.nonzero(m)[0][np.argmax(eh)]
            owner = [n for n, bf in plan.names.items() if bf.off <= 2 * idx < bf.off + bf.nbytes]
            print("   sample", b, "worst fp16 at byte", 2 * idx, owner, "got", gh[idx], "want", wh[idx])
        if ef.size and ef.max() > 0.01 and bad < 3:
            vi = np.nonzero(vf)[0][np.argmax(ef)]
            print("   sample", b, "worst fp32 at byte", 4 * vi, "got", gf[vi], "want", wf[vi])
    op = plan.ops[k - 1]
    print("op %2d %-8s rows_log2 %d strips %d parts %d  worst rel diff %.4f" % (k, names[op.type], op.rows_log2, op.n_strips, op.parts, worst))
    if worst > 0.05:
        bad += 1
        if bad >= 3:
            break
