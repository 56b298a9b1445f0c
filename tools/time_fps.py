"""FPS (furthest_point_sampling + the plain decode FPS) time per selection on the decode / encode / 8(d) cloud sizes"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slide_amd import _ext as E
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
for B in (256, 2048):
    for n, m in ((1024, 256), (2048, 1024), (4096, 2048), (8192, 2048)):
        xyz = (torch.rand(B, n, 3, device=dev, generator=g) * 2 - 1).contiguous()
        for name, fn in (("furthest_point_sampling", lambda: E.furthest_point_sampling(xyz, m)),
                         ("sample_farthest_points", lambda: E.sample_farthest_points(xyz, K=m))):
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            print("B %4d n %5d m %5d %-24s %8.1f us  %.3f us / selection" % (B, n, m, name, dt * 1e6, dt * 1e6 / m), flush=True)
