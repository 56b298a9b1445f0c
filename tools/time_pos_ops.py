"""event-timed launches of the POSITION step plan (slide_run_ops_timed), fp16 and fp32 modes, batch B (default 256)"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slide_amd import configs, model_spec
import slide_amd.engine as E
from slide_amd._lib import check, lib
from slide_amd.diffusion import PositionSampler
from slide_amd.synth import synth_state_dict
names = {getattr(E, n): n for n in dir(E) if n.startswith("OP_")}
dev = torch.device("cuda:0"); B = int(os.environ.get("B", 256))
pc = configs.position_ddpm_config()
sd = synth_state_dict(model_spec.denoiser_param_spec(pc["pointnet_config"]))
rs = np.random.RandomState(0)
for prec in sys.argv[1:] or ["fp16", "fp32"]:
    s = PositionSampler(pc["pointnet_config"], sd, B, dev, pc["diffusion_config"], prec=prec, use_graph=False)
    s.begin(np.zeros(B, np.int64), rs.standard_normal((B, 16, 3)).astype(np.float32))
    n = len(s.step_ops)
    ms = (ctypes.c_float * n)()
    tot = np.zeros(n)
    with torch.cuda.stream(s.stream):
        for r in range(6):
            check(lib().slide_run_ops_timed(s.step_ops, n, ctypes.c_void_p(s.stream.cuda_stream), ms), "timed")
            if r:
                tot += np.array(list(ms))
    tot /= 5
    s.stream.synchronize()
    s.begin(np.zeros(B, np.int64), rs.standard_normal((B, 16, 3)).astype(np.float32))
    s.advance(50); s.stream.synchronize()
    t0 = time.perf_counter(); s.advance(300); s.stream.synchronize(); dt = (time.perf_counter() - t0) / 300
    print("== %s: %d launches, sum of event-timed launches %.1f us, chain alone %.1f us/step" % (prec, n, tot.sum() * 1e3, dt * 1e6))
    for i in range(n):
        o = s.step_ops[i]
        print("  %2d %-18s %-34s %7.1f us  i=%s" % (i, names.get(o.kind, o.kind), s.engine.kernel_names.get(i, ""), tot[i] * 1e3, list(o.i)[:10]))
