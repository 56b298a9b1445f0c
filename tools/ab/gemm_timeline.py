"""Per-workgroup timeline of one GEMM launch (100 MHz stamps written by the kernel when SlideOp.p[5] is set).
usage: python tools/ab/gemm_timeline.py rows,npxl,K,N,mode[,extras]   (same case syntax as tools/ab/gemm_micro.py)
stamps: 0 start | 1 tables staged + ring primed | 2 K loop done | 3 partial statistics published | 4 barrier passed |
        5 stores issued | 6 stores retired"""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
LIBT = os.path.join(ROOT, "build_tmp", "libT.so")
if "--build" in sys.argv:  # instrumented copy of the library (-DSLIDE_TIMELINE); run this part where hipcc is
    sys.argv.remove("--build")
    os.makedirs(os.path.dirname(LIBT), exist_ok=True)
    F = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-DSLIDE_TIMELINE"]
    subprocess.check_call(F + ["-ffp-contract=off", "-c", ROOT + "/slide_amd/csrc/point_ops.hip", "-o", ROOT + "/build_tmp/pT.o"], stderr=subprocess.DEVNULL)
    subprocess.check_call(F + ["-mllvm", "-pragma-unroll-threshold=100000", "-c", ROOT + "/slide_amd/csrc/engine.hip", "-o", ROOT + "/build_tmp/eT.o"], stderr=subprocess.DEVNULL)
    subprocess.check_call(F + ["-mllvm", "-pragma-unroll-threshold=100000", "-c", ROOT + "/slide_amd/csrc/gemm_xs.hip", "-o", ROOT + "/build_tmp/xT.o"], stderr=subprocess.DEVNULL)
    subprocess.check_call(F + ["-mllvm", "-pragma-unroll-threshold=100000", "-c", ROOT + "/slide_amd/csrc/resident.hip", "-o", ROOT + "/build_tmp/rT.o"], stderr=subprocess.DEVNULL)
    subprocess.check_call(F + ["-c", ROOT + "/slide_amd/csrc/rows_ops.hip", "-o", ROOT + "/build_tmp/oT.o"], stderr=subprocess.DEVNULL)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIBT, ROOT + "/build_tmp/pT.o", ROOT + "/build_tmp/eT.o",
                           ROOT + "/build_tmp/xT.o", ROOT + "/build_tmp/rT.o", ROOT + "/build_tmp/oT.o"])
    if len(sys.argv) == 1:
        sys.exit(0)
os.environ["SLIDE_HIP_LIB"] = LIBT
import torch
from slide_amd import engine as E
from slide_amd._lib import check, lib
import gemm_micro as G

for spec in sys.argv[1:]:
    f = spec.split(",")
    rows, npxl, K, N, mode = (int(x) for x in f[:5])
    extras = tuple(f[5:])
    B = rows >> npxl
    m = G.Mini(B, "fp16")
    cm = "cm" in extras and npxl >= 7  # chunk-major input / output / weights, as in the fp16 plans
    if cm:
        m.use_cm, m._cm = True, set()
    rs = np.random.RandomState(0)
    X = m.A.put(rs.standard_normal((rows, K)).astype(np.float32), m.adt)
    if cm:
        m._cm.add(X.data_ptr())
    out = m._buf(rows, N, cm=cm)
    seg = dict(w=rs.standard_normal((N, K)).astype(np.float32) / np.sqrt(K), bias=rs.standard_normal(N).astype(np.float32),
               mode=mode, out=out)
    if mode == E.EPI_NORM:
        seg.update(flags=E.F_POST_RELU, layout=E.gn_layout(N), gn=(np.ones(N, np.float32), np.zeros(N, np.float32)))
    if mode == E.EPI_STATS:
        seg.update(flags=E.F_PRE_RELU, stats=(m.A.zeros(B, E.ru(N)), m.A.zeros(B, E.ru(N)), 0, 1.0))
    if "res" in extras:
        seg["residual"] = m._buf(rows, N)
    if "addvec" in extras:
        seg["addvec"] = (m.A.zeros(B, E.ru(N)), 0, E.ru(N), None, 0)
    m._gemm(X, npxl, [seg])
    op = m.ops[0]
    nwg = 8192
    dbg = torch.zeros(nwg * 16, dtype=torch.int64, device=G.dev)
    op.p[5] = dbg.data_ptr()
    ops = (E.SlideOp * 1)(op)
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        check(lib().slide_run_ops(ops, 1, s), "run")
    torch.cuda.synchronize()
    t = dbg.cpu().numpy().reshape(nwg, 16).astype(np.float64)
    t = t[t[:, 0] > 0]
    if op.p[10]:  # X-stationary kernel: 0 start | 1 DMA issued, tables staged | 2 X landed | 6 / 7 first tile: K loop, epilogue done | 13 / 14 last tile
        t0 = t[:, 0].min()
        tt = (t - t0) / 100.0
        m = tt.mean(axis=0)
        print("case %s (X-stationary): %d workgroups, span %.1f us, start spread %.1f us" % (spec, len(t), tt[:, 14].max(), tt[:, 0].max()))
        print("   issue %.2f  landed %.2f  | first tile: kloop %.2f  epilogue %.2f (stats %.2f, barrier->stores %.2f) | last tile ends %.2f: kloop done @%.2f, epilogue %.2f" % (
            m[1] - m[0], m[2] - m[1], m[6] - m[2], m[7] - m[6], 0, 0, m[14], m[13], m[14] - m[13]))
        continue
    t0 = t[:, 0].min()
    t = (t - t0) / 100.0  # us
    print("case %s: %d workgroups, launch span %.1f us" % (spec, len(t), t[:, 6].max()))
    names = ["start", "primed", "kloop", "stats", "barrier", "stored", "retired"]
    order = np.argsort(t[:, 0])
    first = t[order[: len(t) // 4]]
    for lab, sel in (("all", t), ("first quarter (by start)", first)):
        d = np.diff(sel[:, :7], axis=1)
        print("  %-26s start@%.1f  " % (lab, sel[:, 0].mean()) +
              "  ".join("%s %.2f" % (names[i + 1], d[:, i].mean()) for i in range(6)) + "  | total %.2f" % (sel[:, 6] - sel[:, 0]).mean())
    # concurrency: how many WGs alive over time
    ev = np.concatenate([np.stack([t[:, 0], np.ones(len(t))], 1), np.stack([t[:, 6], -np.ones(len(t))], 1)])
    ev = ev[np.argsort(ev[:, 0])]
    alive = np.cumsum(ev[:, 1])
    print("  max concurrent workgroups %d" % alive.max())
    if t[:, 8].max() > 0:  # finer stamps inside phase 2 of channel block 0
        print("  phase 2, block 0: descriptors %.2f  totals+normalise %.2f  convert+store %.2f  (us, mean)" % (
            (t[:, 8] - t[:, 4]).mean(), (t[:, 9] - t[:, 8]).mean(), (t[:, 10] - t[:, 9]).mean()))
        if t[:, 11].max() > 0:
            print("     early loads + totals from LDS %.2f  mean/rstd %.2f  scale+fma %.2f" % (
                (t[:, 11] - t[:, 8]).mean(), (t[:, 12] - t[:, 11]).mean(), (t[:, 9] - t[:, 12]).mean()))
