"""HBM roofline of the stand-alone pointnet2_ops kernels (SURVEY.md section 8(d) byte formulas), through the C-ABI.
usage: python tools/bench_ops.py [--json out.json]
Algorithmic bytes (fp32 / int32, per call):
  group_points(B,c,n,np,ns):    read 4 c n B + 4 np ns B,  write 4 c np ns B
  gather_points(B,c,n,m):       read 4 c m B + 4 m B,      write 4 c m B
  three_interpolate(B,c,m,n):   read 4 c m B + 24 n B,     write 4 c n B
  three_nn(B,n,m):              read 12 (n + m) B,         write 24 n B      (compute-bound: n*m distance evaluations)
  ball_query(B,n,m,ns):         read 12 (n + m) B,         write 4 m (ns + 1) B
  knn_points(B,n1,n2,K):        read 12 (n1 + n2) B,       write 12 n1 K B
  furthest_point_sampling(B,n,m): read 12 n B, write 4 m B  (latency-bound: m dependent block-wide arg-max steps)"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slide_amd import _ext as E

PEAK = 8000.0  # GB/s, MI355X HBM3E (MI355X_MICROARCH.md)
dev = torch.device("cuda:0")
ap = argparse.ArgumentParser(); ap.add_argument("--json"); a = ap.parse_args()
g = torch.Generator(device=dev); g.manual_seed(0)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps  # us


rows = []


def report(name, shape, us, nbytes, note=""):
    gbs = nbytes / us / 1e3
    rows.append(dict(op=name, shape=shape, us=round(us, 1), algorithmic_MB=round(nbytes / 1e6, 2), GBps=round(gbs, 1),
                     frac_of_hbm_peak=round(gbs / PEAK, 4), note=note))
    print("%-24s %-34s %9.1f us %9.1f MB %8.1f GB/s  %5.1f%% of %d GB/s  %s" % (name, shape, us, nbytes / 1e6, gbs, 100 * gbs / PEAK, PEAK, note))


B = 256
for (c, n, npo, ns) in [(128, 2048, 512, 32), (256, 1024, 256, 32), (64, 1024, 1024, 16)]:
    pts = torch.randn(B, c, n, device=dev, generator=g)
    idx = torch.randint(0, n, (B, npo, ns), device=dev, generator=g, dtype=torch.int32)
    # zero-filled output allocation (the reference wrapper's torch.zeros) is part of the op as shipped; time both
    us = timeit(lambda: E.group_points(pts, idx))
    report("group_points", "B%d c%d n%d np%d ns%d" % (B, c, n, npo, ns), us, 4 * B * (c * n + npo * ns + c * npo * ns), "incl. zeros alloc")
for (c, n, m) in [(256, 2048, 1024), (128, 8192, 2048)]:
    pts = torch.randn(B, c, n, device=dev, generator=g)
    idx = torch.randint(0, n, (B, m), device=dev, generator=g, dtype=torch.int32)
    us = timeit(lambda: E.gather_points(pts, idx))
    report("gather_points", "B%d c%d n%d m%d" % (B, c, n, m), us, 4 * B * (c * m + m + c * m), "incl. zeros alloc")
for (c, m, n) in [(256, 256, 2048), (128, 1024, 4096)]:
    pts = torch.randn(B, c, m, device=dev, generator=g)
    idx = torch.randint(0, m, (B, n, 3), device=dev, generator=g, dtype=torch.int32)
    w = torch.rand(B, n, 3, device=dev, generator=g)
    us = timeit(lambda: E.three_interpolate(pts, idx, w))
    report("three_interpolate", "B%d c%d m%d n%d" % (B, c, m, n), us, 4 * B * (c * m + 6 * n + c * n), "incl. zeros alloc")
for (n, m) in [(2048, 512), (4096, 1024)]:
    u = torch.randn(B, n, 3, device=dev, generator=g); k = torch.randn(B, m, 3, device=dev, generator=g)
    us = timeit(lambda: E.three_nn(u, k))
    report("three_nn", "B%d n%d m%d" % (B, n, m), us, B * (12 * (n + m) + 24 * n), "%.1f G dist/s" % (B * n * m / us / 1e3))
for (n, m, ns) in [(2048, 512, 32), (1024, 1024, 16)]:
    x = torch.rand(B, n, 3, device=dev, generator=g); q = x[:, :m].contiguous()
    us = timeit(lambda: E.ball_query(q, x, 0.2, ns))
    report("ball_query", "B%d n%d m%d ns%d" % (B, n, m, ns), us, B * (12 * (n + m) + 4 * m * (ns + 1)), "%.1f G dist/s" % (B * n * m / us / 1e3))
for (n1, n2, K) in [(2048, 2048, 16), (1024, 256, 16), (16, 16, 16)]:
    p1 = torch.randn(B, n1, 3, device=dev, generator=g); p2 = torch.randn(B, n2, 3, device=dev, generator=g)
    us = timeit(lambda: E.knn_points(p1, p2, K))
    report("knn_points", "B%d n1 %d n2 %d K%d" % (B, n1, n2, K), us, B * (12 * (n1 + n2) + 12 * n1 * K), "%.1f G dist/s" % (B * n1 * n2 / us / 1e3))
for (n, m) in [(2048, 512), (1024, 256), (8192, 2048)]:
    x = torch.randn(B, n, 3, device=dev, generator=g)
    us = timeit(lambda: E.furthest_point_sampling(x, m), reps=5)
    report("furthest_point_sampling", "B%d n%d m%d" % (B, n, m), us, B * (12 * n + 4 * m), "%.2f us per selection step" % (us / m))
if a.json:
    json.dump(dict(peak_GBps=PEAK, rows=rows), open(a.json, "w"), indent=1)
