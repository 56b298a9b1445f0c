"""micro-benchmark of the engine GEMM op in isolation (one op, many reps, HIP events)"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from slide_amd import engine as E
from slide_amd._lib import check, lib

dev = torch.device("cuda:0")


class Mini(E.DenoiserEngine):
    def __init__(self, B, prec):
        self.B, self.device, self.prec = B, dev, E.PREC[prec]
        self.adt = torch.float16 if self.prec == 1 else torch.float32
        self.A = E._Arena(dev); self.ops = []; self.flops = 0; self.gemm_flops = {}
        self.per_sample_t = True
        self.two_lanes = False; self._lane = 0; self.gemm_bytes = {}
        self.persistent = os.environ.get('SLIDE_PERSISTENT', '0') != '0'
        self.use_glds = os.environ.get('SLIDE_GLDS', '1') != '0'
        self.glds_nst = int(os.environ.get('SLIDE_GLDS_WIDE', '0'))


def bench(rows, npxl, K, N, mode, prec="fp16", extras=(), reps=20):
    B = rows >> npxl
    m = Mini(B, prec)
    rs = np.random.RandomState(0)
    X = m.A.put(rs.standard_normal((rows, K)).astype(np.float32), m.adt)
    out = m._buf(rows, N)
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
    ops = (E.SlideOp * 1)(*m.ops)
    L = lib(); s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        check(L.slide_run_ops(ops, 1, s), "run")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        check(L.slide_run_ops(ops, 1, s), "run")
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    fl = 2.0 * rows * K * N
    print("rows %6d npx %3d K %4d N %4d mode %d %-12s %s: %8.1f us  %7.1f TF" % (rows, 1 << npxl, K, N, mode, ",".join(extras), prec, us, fl / us / 1e6))


if __name__ == "__main__":
    if len(sys.argv) > 1:  # rows,npxl,K,N,mode[,extras...]   e.g. 65536,8,64,448,1
        for spec in sys.argv[1:]:
            f = spec.split(",")
            bench(int(f[0]), int(f[1]), int(f[2]), int(f[3]), int(f[4]), "fp16", tuple(f[5:]), reps=10)
        sys.exit(0)
    for prec in ["fp16"]:
        for mode in (E.EPI_RAW, E.EPI_STATS, E.EPI_NORM):
            bench(4096, 4, 128, 128, mode, prec)
        bench(4096, 4, 128, 128, E.EPI_NORM, prec, ("res", "addvec"))
        bench(4096, 4, 128, 32, E.EPI_NORM, prec)
        bench(4096, 4, 512, 128, E.EPI_RAW, prec)
        bench(4096, 4, 2048, 128, E.EPI_RAW, prec)
        for mode in (E.EPI_RAW, E.EPI_NORM):
            bench(65536, 8, 256, 256, mode, prec)
            bench(65536, 8, 512, 512, mode, prec)
            bench(65536, 8, 2048, 512, mode, prec)
        bench(65536, 8, 64, 512, E.EPI_RAW, prec)
        bench(65536 * 4, 8, 512, 512, E.EPI_RAW, prec)
