"""Stand-alone pointnet2_ops kernels on SURVEY.md section 8(d)'s synthetic sizes, to be run UNDER rocprofv3 (kernel trace or
PMC pass): every case is `reps` calls through the product API (slide_amd/_ext.py -> C-ABI), cases are separated by a marker
kernel (torch.arange) so that the report can attribute dispatches.  Writes the case manifest (algorithmic bytes per call,
SURVEY 8(d) formulas) to --manifest.
usage: rocprofv3 --kernel-trace --output-format csv -d OUT -o t -- python tools/ops_roofline.py --manifest OUT/manifest.json"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slide_amd import _ext as E

ap = argparse.ArgumentParser()
ap.add_argument("--manifest", required=True); ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--max-gb", type=float, default=40.0)  # (B 2048 x N 8192 group_points writes 34 GB; the GPU holds 288)
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
SIZES = [(16, 16, 16, 51), (16, 16, 16, 256), (256, 128, 32, 64), (1024, 256, 32, 128), (2048, 1024, 32, 32),
         (8192, 2048, 32, 64)]  # (N, npoint, nsample, C): every row of SURVEY.md section 8(d)
cases = []


def marker():
    torch.arange(7, device=dev)


def run(op, shape, nbytes, fn, extra=None):
    if nbytes["alloc"] > a.max_gb * 1e9:
        return
    marker()
    for _ in range(a.reps):
        fn()
    torch.cuda.synchronize()
    cases.append(dict(op=op, shape=shape, reps=a.reps, read_B=nbytes["r"], write_B=nbytes["w"], **(extra or {})))


for B in (256, 2048):
    for (N, M, NS, C) in SIZES:
        tag = "B%d N%d np%d ns%d C%d" % (B, N, M, NS, C)
        xyz = (torch.rand(B, N, 3, device=dev, generator=g) * 2 - 1).contiguous()  # U(-1, 1), seed 0
        new_xyz = xyz[:, :M].contiguous()
        feats = None
        if 4.0 * B * C * N <= a.max_gb * 1e9:
            feats = torch.randn(B, C, N, device=dev, generator=g)
        idx_g = torch.randint(0, N, (B, M, NS), device=dev, generator=g, dtype=torch.int32)
        idx_m = torch.randint(0, N, (B, M), device=dev, generator=g, dtype=torch.int32)
        if feats is not None:
            run("group_points", tag, dict(r=4 * B * (C * N + M * NS), w=4 * B * C * M * NS, alloc=4.0 * B * C * M * NS),
                lambda: E.group_points(feats, idx_g))
            run("gather_points", tag, dict(r=4 * B * (C * M + M), w=4 * B * C * M, alloc=4.0 * B * C * M),
                lambda: E.gather_points(feats, idx_m), dict(physical_read_B=4 * B * C * N))
            # the same gather on a ROW-MAJOR (B, N, C) table (slide_gather_rows): moves what it gathers
            feats_r = feats.transpose(1, 2).contiguous()
            run("gather_rows", tag, dict(r=4 * B * (C * M + M), w=4 * B * C * M, alloc=4.0 * B * C * M),
                lambda: E.gather_rows(feats_r, idx_m))
            del feats_r
        known = torch.randn(B, C, M, device=dev, generator=g)
        idx3 = torch.randint(0, M, (B, N, 3), device=dev, generator=g, dtype=torch.int32)
        w3 = torch.rand(B, N, 3, device=dev, generator=g)
        if 4.0 * B * C * N <= a.max_gb * 1e9:
            run("three_interpolate", tag, dict(r=4 * B * C * M + 24 * B * N, w=4 * B * C * N, alloc=4.0 * B * C * N),
                lambda: E.three_interpolate(known, idx3, w3))
        run("three_nn", tag, dict(r=12 * B * (N + M), w=24 * B * N, alloc=0), lambda: E.three_nn(xyz, new_xyz),
            dict(dist_evals=B * N * M))
        run("ball_query", tag, dict(r=12 * B * (N + M), w=4 * B * M * (NS + 1), alloc=0), lambda: E.ball_query(new_xyz, xyz, 0.2, NS),
            dict(dist_evals=B * N * M))
        run("knn_points", tag, dict(r=12 * B * (N + M), w=12 * B * M * NS, alloc=0), lambda: E.knn_points(new_xyz, xyz, NS),
            dict(dist_evals=B * N * M))
        run("furthest_point_sampling", tag, dict(r=12 * B * N, w=4 * B * M, alloc=0), lambda: E.furthest_point_sampling(xyz, M),
            dict(selections=M))
        del feats, xyz, known, idx_g, idx3, w3
        torch.cuda.empty_cache()
marker()
torch.cuda.synchronize()
json.dump(dict(cases=cases), open(a.manifest, "w"), indent=1)
print("cases", len(cases))
