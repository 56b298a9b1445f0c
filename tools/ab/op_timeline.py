"""Per-workgroup timeline of single launches of the feature / position step plan (100 MHz stamps written by instrumented
kernels: SlideOp.p[5] of SLIDE_OP_GEMM, p[12] of SLIDE_OP_GEMM_GX).
  python tools/ab/op_timeline.py --build                 (where hipcc is: build_tmp/libT.so, -DSLIDE_TIMELINE)
  python tools/ab/op_timeline.py feat 256 13 14 ...      (on the GPU box: op indices as tools/profile_ops.py prints them)
stamps: 0 start | 7 tables staged (GX) | 1 ring primed / lane set-up done | 2 K loop done | 3 partial statistics published |
        4 barrier passed | 5 stores issued | 6 stores retired"""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
LIBT = os.path.join(ROOT, "build_tmp", "libT.so")
if "--build" in sys.argv:
    from slide_amd import build as B
    os.makedirs(os.path.dirname(LIBT), exist_ok=True)
    objs = []
    for src, extra in B.SOURCES:
        o = os.path.join(ROOT, "build_tmp", "T_" + src.replace(".hip", ".o"))
        subprocess.check_call([B.HIPCC] + B.COMMON + extra + ["-DSLIDE_TIMELINE", "-c", os.path.join(B.CSRC, src), "-o", o],
                              stderr=subprocess.DEVNULL)
        objs.append(o)
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIBT] + objs)
    sys.exit(0)
os.environ["SLIDE_HIP_LIB"] = LIBT
import torch
from slide_amd import configs, model_spec
from slide_amd._lib import check, lib
from slide_amd.engine import SlideOp
from slide_amd.diffusion import FeatureSampler, PositionSampler
from slide_amd.synth import synth_keypoints, synth_state_dict
which, B = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda:0")
rs = np.random.RandomState(0)
if which == "feat":
    c = configs.feature_ddpm_config()
    s = FeatureSampler(c["pointnet_config"], synth_state_dict(model_spec.denoiser_param_spec(c["pointnet_config"])), B, dev,
                       c["standard_diffusion_config"], prec="fp16")
    s.begin(np.full(B, 4, np.int64), synth_keypoints(B), rs.standard_normal((B, 16, 51)).astype(np.float32))
else:
    c = configs.position_ddpm_config()
    s = PositionSampler(c["pointnet_config"], synth_state_dict(model_spec.denoiser_param_spec(c["pointnet_config"])), B, dev,
                        c["diffusion_config"], prec=os.environ.get("TL_PREC", "fp16"))
    s.begin(np.zeros(B, np.int64), rs.standard_normal((B, 16, 3)).astype(np.float32))
n = len(s.step_ops)
st = ctypes.c_void_p(s.stream.cuda_stream)
with torch.cuda.stream(s.stream):
    check(lib().slide_run_ops(s.step_ops, n, st), "warm")
    want = sys.argv[3:]
    if want and want[0].startswith("kind"):  # "kind31" / "kind1": every launch of that op kind
        want = [i for i in range(n) if s.step_ops[i].kind == int(want[0][4:])]
    for idx in (int(v) for v in want):
        op = SlideOp.from_buffer_copy(bytes(s.step_ops[idx]))
        nwg = 16384
        dbg = torch.zeros(nwg * 16, dtype=torch.int64, device=dev)
        slot = {1: 5, 17: 12, 19: 12, 16: 8, 30: 1, 31: 13}.get(op.kind)
        if op.kind == 17 and int(op.f[0]) == 3:  # split generated-X GEMM (gemm_gxs.hip): p[12] / p[13] carry a chained layer
            slot = None if op.p[12] else 13
        if slot is None:
            print("op %d: kind %d carries no stamps" % (idx, op.kind)); continue
        op.p[slot] = dbg.data_ptr()
        ops = (SlideOp * 1)(op)
        for _ in range(3):
            check(lib().slide_run_ops(ops, 1, st), "run")
        s.stream.synchronize()
        t = dbg.cpu().numpy().reshape(nwg, 16).astype(np.float64)
        t = t[t[:, 0] > 0]
        if not len(t):
            print("op %d kind %d: no stamps (kernel variant without instrumentation)" % (idx, op.kind)); continue
        t0 = t[:, 0].min()
        t = (t - t0) / 100.0
        seq = [0, 7, 1, 2, 3, 4, 5, 6] if op.kind == 17 else [0, 1, 2, 3, 4, 5, 6]
        names = {0: "start", 7: "tables", 1: "primed", 2: "kloop", 3: "stats", 4: "barrier", 5: "stored", 6: "retired"}
        if op.kind == 16:  # eight-wave attention tail: 1 first DMA issued | 2 landed | 3 scores K loop | 4 values K loop | 5 statistics | 6 softmax + stores | 7 retired
            seq = [0, 1, 2, 3, 4, 5, 6, 7]
            names = {1: "prologue", 2: "first_stage", 3: "kloop_s", 4: "kloop_v", 5: "stats", 6: "softmax", 7: "retired"}
            if os.environ.get("SLIDE_TAIL8", "0") == "0":  # register-X tail: 1 pipeline primed | 2 first RXD chunks | 3 values K loop | 4 scores K loop | 5 drained + barrier | 6 epilogue | 7 retired
                names = {1: "primed", 2: "first4", 3: "kloop_v", 4: "kloop_s", 5: "drain", 6: "epilogue", 7: "retired"}
        if op.kind == 30:  # block body: 1 prologue | 2 h2 | 3 mo | 4 u | 5 tail | 6 retired
            seq = [0, 1, 2, 3, 4, 5, 6]
            names = {1: "prologue", 2: "h2", 3: "mo", 4: "u", 5: "tail", 6: "retired"}
        if op.kind == 19:  # fused SA chain: 14 prologue | 1 stage-1 K loop | 2 h2 in registers | 3 / 6 slab K loop | 4 / 7 statistics | 5 / 8 stored | 15 retired
            seq = [0, 14, 1, 2, 3, 4, 5] + ([6, 7, 8] if op.i[4] > 256 else []) + [15]
            names = {14: "prologue", 1: "kloop1", 2: "epi1", 3: "kloop2a", 4: "stats2a", 5: "store2a", 6: "kloop2b", 7: "stats2b", 8: "store2b", 15: "retired"}
            t[:, 6] = t[:, 15] if op.i[4] <= 256 else t[:, 6]
        last = seq[-1]
        print("op %d kind %d rows %d k %d n %d: %d workgroups, span %.1f us, mean start %.1f" % (
            idx, op.kind, op.i[0], op.i[2], op.i[3] * 32, len(t), t[:, last].max(), t[:, 0].mean()))
        print("   " + "  ".join("%s %.2f" % (names[b_], (t[:, b_] - t[:, a_]).mean()) for a_, b_ in zip(seq[:-1], seq[1:])) +
              "  | total %.2f" % (t[:, last] - t[:, 0]).mean())
