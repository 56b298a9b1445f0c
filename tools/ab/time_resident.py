"""ms per reverse step of the LDS-resident position sampler (one launch of n steps) vs the engine plan, batch 256"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from slide_amd import configs, model_spec
from slide_amd.diffusion import PositionSampler
from slide_amd.experiments.resident import ResidentPositionSampler
from slide_amd.synth import synth_state_dict
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
pc = configs.position_ddpm_config()
sd = synth_state_dict(model_spec.denoiser_param_spec(pc["pointnet_config"]))
rs = np.random.RandomState(0)
xT = rs.standard_normal((B, 16, 3)).astype(np.float32)
lab = np.zeros(B, np.int64)
r = ResidentPositionSampler(pc["pointnet_config"], sd, B, dev, pc["diffusion_config"], seed=5)
print("lds bytes", r.plan.lds_bytes, "ops", len(r.plan.ops), "weights KB", r.plan.weight_bytes / 1024)
for rep in range(3):
    r.begin(lab, xT)
    r.stream.synchronize()
    t0 = time.perf_counter()
    r.advance(n)
    r.stream.synchronize()
    dt = time.perf_counter() - t0
    print("resident: %d steps  %.4f ms/step" % (n, dt * 1e3 / n), "finite", bool(torch.isfinite(r.state()).all()))
p = PositionSampler(pc["pointnet_config"], sd, B, dev, pc["diffusion_config"], prec="fp16", seed=5, use_graph=True)
for rep in range(2):
    p.begin(lab, xT)
    p.stream.synchronize()
    t0 = time.perf_counter()
    p.advance(n)
    p.stream.synchronize()
    dt = time.perf_counter() - t0
    print("engine plan (graph): %d steps  %.4f ms/step" % (n, dt * 1e3 / n))
xr, xp = r.state().cpu().numpy(), p.state().cpu().numpy()
print("final-state rel diff resident vs plan (same seeds, 1000 steps, fp16 both):", float(np.abs(xr - xp).max() / np.abs(xp).max()))
# per-op timeline of workgroup 0 (shader clock), first step of a 3-step launch
tl = torch.zeros(len(r.plan.ops) + 2, dtype=torch.int64, device=dev)
r.begin(lab, xT)
with torch.cuda.stream(r.stream):
    a = r.plan.args(3, r.engine.x, t_dev=r.engine.t_dev, tabs=r.tabs, seed=5, timeline=tl)
    from slide_amd.experiments.resident import _run
    _run(a, r.stream)
r.stream.synchronize()
t = tl.cpu().numpy()
names = {1: "PREP", 2: "ASSEMBLE", 3: "GEMM", 4: "FINALIZE", 5: "AFFINE", 6: "TAIL", 7: "ZFILL"}
tot = t[len(r.plan.ops)] - t[0]
print("timeline: %d ticks per step (100 MHz s_memtime -> %.1f us)" % (tot, tot / 100.0))
for i, op in enumerate(r.plan.ops):
    d = t[i + 1] - t[i]
    print("  op %2d %-8s rows 2^%d strips %d parts %d nks %d+%d | %d+%d : %6.2f us" % (i, names[op.type], op.rows_log2, op.n_strips, op.parts,
          op.a.nks_gat, op.a.nks_x, op.b.nks_gat, op.b.nks_x, d / 100.0))
