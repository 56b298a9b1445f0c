"""tools/ref_shims.py -- AUTHORING-CONTAINER ONLY (never imported by tests -m gpu / bench / smoke).

Makes the reference's *Python* importable on CPU so that tools/gen_golden.py can run it and
record golden input/output vectors:
  1. `pointnet2_ops._ext`      -> oracle/ops.py (plain-C CPU restatement), wrapped for torch tensors
  2. `pytorch3d.ops.knn` etc.  -> oracle knn_points / knn_gather (pytorch3d is not installed)
  3. Tensor.cuda / Module.cuda -> identity
  4. empty stubs for torchvision / plyfile
Nothing from /root/reference is copied; it is imported where it lies.
"""
import collections
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _n(t):
    return t.detach().cpu().numpy()


def install():
    sys.dont_write_bytecode = True
    if REPO not in sys.path:
        sys.path.append(REPO)
    from oracle import ops as O

    ext = types.ModuleType("pointnet2_ops._ext")
    ext.gather_points = lambda p, i: torch.from_numpy(O.gather_points(_n(p), _n(i)))
    ext.gather_points_grad = lambda g, i, n: torch.from_numpy(O.gather_points_grad(_n(g), _n(i), n))
    ext.furthest_point_sampling = lambda p, m: torch.from_numpy(O.furthest_point_sampling(_n(p), m))
    ext.three_nn = lambda u, k: [torch.from_numpy(a) for a in O.three_nn(_n(u), _n(k))]
    ext.three_interpolate = lambda p, i, w: torch.from_numpy(O.three_interpolate(_n(p), _n(i), _n(w)))
    ext.three_interpolate_grad = lambda g, i, w, m: torch.from_numpy(
        O.three_interpolate_grad(_n(g), _n(i), _n(w), m))
    ext.ball_query = lambda q, x, r, ns: tuple(torch.from_numpy(a) for a in O.ball_query(_n(q), _n(x), r, ns))
    ext.group_points = lambda p, i: torch.from_numpy(O.group_points(_n(p), _n(i)))
    ext.group_points_grad = lambda g, i, n: torch.from_numpy(O.group_points_grad(_n(g), _n(i), n))
    sys.modules["pointnet2_ops._ext"] = ext

    KNN = collections.namedtuple("KNN", "dists idx knn")

    def knn_points(p1, p2, lengths1=None, lengths2=None, norm=2, K=1, version=-1, return_nn=False,
                   return_sorted=True):
        l2 = None if lengths2 is None else _n(lengths2).astype(np.int64)
        d, i = O.knn_points(_n(p1), _n(p2), K, l2)
        nn = None
        if return_nn:
            nn = torch.from_numpy(O.knn_gather(_n(p2), i))
        return KNN(torch.from_numpy(d), torch.from_numpy(i), nn)

    def knn_gather(x, idx, lengths=None):
        # pytorch3d's own formulation (a torch gather: differentiable in x, which the gradient fixtures need; same values as
        # oracle.ops.knn_gather)
        B, N1, K = idx.shape
        U = x.shape[2]
        return x[:, :, None].expand(-1, -1, K, -1).gather(1, idx[:, :, :, None].expand(-1, -1, -1, U).long())

    def sample_farthest_points(points, lengths=None, K=50, random_start_point=False):
        # plain fp32 FPS from the oracle; start index 0 (the reference's random start makes its result
        # non-reproducible: "parity unpinned", compared distributionally)
        out, idx = O.sample_farthest_points(_n(points), K, None)
        return torch.from_numpy(out), torch.from_numpy(idx)

    def masked_gather(points, idx):
        return torch.from_numpy(np.take_along_axis(_n(points), _n(idx)[..., None].astype(np.int64), axis=1))

    p3d = types.ModuleType("pytorch3d")
    p3d_ops = types.ModuleType("pytorch3d.ops")
    p3d_knn = types.ModuleType("pytorch3d.ops.knn")
    p3d_utils = types.ModuleType("pytorch3d.ops.utils")
    p3d_struct = types.ModuleType("pytorch3d.structures")
    p3d_pc = types.ModuleType("pytorch3d.structures.pointclouds")
    p3d_knn.knn_points = knn_points; p3d_knn.knn_gather = knn_gather
    p3d_ops.knn = p3d_knn; p3d_ops.knn_points = knn_points; p3d_ops.knn_gather = knn_gather
    p3d_ops.sample_farthest_points = sample_farthest_points
    p3d_utils.masked_gather = masked_gather
    p3d_ops.utils = p3d_utils
    p3d_pc.Pointclouds = type("Pointclouds", (), {})
    p3d_struct.pointclouds = p3d_pc; p3d_struct.Pointclouds = p3d_pc.Pointclouds
    p3d.ops = p3d_ops; p3d.structures = p3d_struct
    for name, mod in [("pytorch3d", p3d), ("pytorch3d.ops", p3d_ops), ("pytorch3d.ops.knn", p3d_knn),
                      ("pytorch3d.ops.utils", p3d_utils), ("pytorch3d.structures", p3d_struct),
                      ("pytorch3d.structures.pointclouds", p3d_pc)]:
        sys.modules[name] = mod

    for name in ["torchvision", "torchvision.transforms", "plyfile"]:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]

    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self

    # the REFERENCE packages must win over this repo's same-named drop-in package
    for p in [os.path.join(REF, "pointnet2"), os.path.join(REF, "pointnet2_ops_lib")]:
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)
    for k in [k for k in sys.modules if k == "pointnet2_ops" or
              (k.startswith("pointnet2_ops.") and k != "pointnet2_ops._ext")]:
        del sys.modules[k]
