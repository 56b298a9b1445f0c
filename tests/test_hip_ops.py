"""GPU parity: the HIP `_ext` operators (through the C-ABI) vs the CPU oracle -- bit-exact indices,
exact gathers; golden fixtures + larger seeded cases + size-independent properties."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import ops as O

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def N(t):
    return t.cpu().numpy()


def test_device_is_gfx950(gpu_device):
    from slide_amd._lib import lib
    assert lib().slide_hip_device_ok() == 1


def test_fps_golden_and_random(gpu_device):
    from slide_amd import _ext
    g = load_golden("golden_ops.npz")
    for ci in range(int(g["n_fps"])):
        p = g["fps%d_in" % ci]
        idx = _ext.furthest_point_sampling(T(p, gpu_device), g["fps%d_idx" % ci].shape[1])
        assert np.array_equal(N(idx), g["fps%d_idx" % ci]), ci
    assert np.array_equal(N(_ext.furthest_point_sampling(T(g["fps_grid_in"], gpu_device), 100)), g["fps_grid_idx"])
    rs = np.random.RandomState(5)
    for (B, n, m) in [(4, 2048, 512), (2, 4096, 300), (2, 8192, 128), (1, 9000, 64), (3, 513, 513), (2, 1, 1),
                      (64, 256, 128)]:
        p = rs.uniform(-1, 1, (B, n, 3)).astype(np.float32)
        p[:, n // 2] = p[:, 0]          # duplicate -> exact ties
        if n > 8:
            p[:, 3] = 0.001             # inside the origin ball
        ref = O.furthest_point_sampling(p, m)
        got = N(_ext.furthest_point_sampling(T(p, gpu_device), m))
        assert np.array_equal(got, ref), (B, n, m)
    # property: FPS of a set is a permutation prefix -> indices unique while distinct points remain
    p = rs.uniform(-1, 1, (8, 1024, 3)).astype(np.float32) + 2.0
    got = N(_ext.furthest_point_sampling(T(p, gpu_device), 1024))
    assert all(len(set(r.tolist())) == 1024 for r in got)


def test_ball_query(gpu_device):
    from slide_amd import _ext
    g = load_golden("golden_ops.npz")
    for ci in range(int(g["n_bq"])):
        idx, cnt = _ext.ball_query(T(g["bq%d_new" % ci], gpu_device), T(g["bq%d_xyz" % ci], gpu_device),
                                   float(g["bq%d_r" % ci]), int(g["bq%d_ns" % ci]))
        assert np.array_equal(N(idx), g["bq%d_idx" % ci]) and np.array_equal(N(cnt), g["bq%d_cnt" % ci]), ci
    rs = np.random.RandomState(6)
    xyz = rs.uniform(-1, 1, (3, 5000, 3)).astype(np.float32)
    q = rs.uniform(-1.2, 1.2, (3, 700, 3)).astype(np.float32)
    ri, rc = O.ball_query(q, xyz, 0.1, 32)
    idx, cnt = _ext.ball_query(T(q, gpu_device), T(xyz, gpu_device), 0.1, 32)
    assert np.array_equal(N(idx), ri) and np.array_equal(N(cnt), rc)


def test_gather_group_interpolate(gpu_device):
    from slide_amd import _ext
    rs = np.random.RandomState(7)
    for (B, C, n, m) in [(2, 3, 256, 128), (4, 51, 16, 16), (2, 256, 1024, 256), (1, 1, 5, 9)]:
        pts = rs.standard_normal((B, C, n)).astype(np.float32)
        idx = rs.randint(0, n, (B, m)).astype(np.int32)
        assert np.array_equal(N(_ext.gather_points(T(pts, gpu_device), T(idx, gpu_device))), O.gather_points(pts, idx))
        G = rs.standard_normal((B, C, m)).astype(np.float32)
        assert np.allclose(N(_ext.gather_points_grad(T(G, gpu_device), T(idx, gpu_device), n)),
                           O.gather_points_grad(G, idx, n), atol=1e-4)
    # (the last two: long rows -> 8 / 16 index quads per thread in the row-in-LDS kernel)
    for (B, C, n, npnt, ns) in [(4, 51, 16, 16, 16), (2, 256, 16, 16, 16), (2, 64, 256, 128, 32), (1, 128, 1024, 256, 32),
                                (2, 3, 7, 5, 3), (1, 5, 4096, 256, 32), (2, 3, 8192, 512, 32)]:
        pts = rs.standard_normal((B, C, n)).astype(np.float32)
        idx = rs.randint(0, n, (B, npnt, ns)).astype(np.int32)
        assert np.array_equal(N(_ext.group_points(T(pts, gpu_device), T(idx, gpu_device))), O.group_points(pts, idx))
        G = rs.standard_normal((B, C, npnt, ns)).astype(np.float32)
        assert np.allclose(N(_ext.group_points_grad(T(G, gpu_device), T(idx, gpu_device), n)),
                           O.group_points_grad(G, idx, n), atol=1e-3)
    # (n = 256 / 512 / 1024: the rows-in-LDS kernel, 4 / 2 / 1 channel rows per pass, ragged channel counts and row lengths)
    for (B, C, m, n) in [(2, 64, 100, 300), (1, 7, 3, 5), (2, 128, 256, 1024), (3, 37, 128, 256), (2, 10, 101, 512),
                         (2, 3, 64, 1024),
                         # staged rows beyond 64 KB of LDS (n = 256: m > 4092; n = 512: m > 8188) take the generic kernels
                         (1, 6, 4096, 256), (1, 5, 8192, 256), (1, 3, 4092, 256), (1, 3, 8192, 512), (1, 3, 8188, 512)]:
        pts = rs.standard_normal((B, C, m)).astype(np.float32)
        idx = rs.randint(0, m, (B, n, 3)).astype(np.int32)
        w = rs.uniform(0, 1, (B, n, 3)).astype(np.float32)
        assert np.array_equal(N(_ext.three_interpolate(T(pts, gpu_device), T(idx, gpu_device), T(w, gpu_device))),
                              O.three_interpolate(pts, idx, w))
        G = rs.standard_normal((B, C, n)).astype(np.float32)
        assert np.allclose(N(_ext.three_interpolate_grad(T(G, gpu_device), T(idx, gpu_device), T(w, gpu_device), m)),
                           O.three_interpolate_grad(G, idx, w, m), atol=1e-3)


def test_gather_rows(gpu_device):
    """row-layout gather (build addition, include/slide_hip.h): (B,N,C) rows by (B,M) indices == numpy take_along_axis, and
    == gather_points on the transposed layout; 16-byte pieces, unaligned widths, repeated and boundary indices, empty output"""
    from slide_amd import _ext
    rs = np.random.RandomState(21)
    for (B, N, M, C) in [(3, 100, 37, 64), (2, 1024, 256, 128), (4, 16, 16, 51), (1, 7, 9, 3), (2, 2048, 1024, 32), (1, 5, 1, 4)]:
        pts = rs.standard_normal((B, N, C)).astype(np.float32)
        idx = rs.randint(0, N, (B, M)).astype(np.int32)
        idx[:, 0] = N - 1
        idx[:, -1] = 0
        got = _ext.gather_rows(T(pts, gpu_device), T(idx, gpu_device)).cpu().numpy()
        want = np.take_along_axis(pts, idx[:, :, None].astype(np.int64), axis=1)
        assert got.shape == (B, M, C) and np.array_equal(got, want) and np.array_equal(O.gather_rows(pts, idx), want)
        ref_api = _ext.gather_points(T(np.ascontiguousarray(pts.transpose(0, 2, 1)), gpu_device), T(idx, gpu_device)).cpu().numpy()
        assert np.array_equal(got, ref_api.transpose(0, 2, 1))
    assert _ext.gather_rows(T(pts, gpu_device), T(np.zeros((1, 0), np.int32), gpu_device)).shape == (1, 0, 4)


def test_three_nn(gpu_device):
    from slide_amd import _ext
    g = load_golden("golden_ops.npz")
    for ci in range(int(g["n_tn"])):
        d, i = _ext.three_nn(T(g["tn%d_u" % ci], gpu_device), T(g["tn%d_k" % ci], gpu_device))
        assert np.array_equal(N(i), g["tn%d_i" % ci]) and np.array_equal(N(d), g["tn%d_d" % ci]), ci
    rs = np.random.RandomState(8)
    u = rs.uniform(-1, 1, (2, 3000, 3)).astype(np.float32); k = rs.uniform(-1, 1, (2, 2500, 3)).astype(np.float32)
    rd, ri = O.three_nn(u, k)
    d, i = _ext.three_nn(T(u, gpu_device), T(k, gpu_device))
    assert np.array_equal(N(i), ri) and np.array_equal(N(d), rd)


def test_knn(gpu_device):
    from slide_amd import _ext
    g = load_golden("golden_ops.npz")
    for ci in range(int(g["n_knn"])):
        d, i = _ext.knn_points(T(g["knn%d_p1" % ci], gpu_device), T(g["knn%d_p2" % ci], gpu_device), int(g["knn%d_K" % ci]))
        assert np.array_equal(N(i), g["knn%d_i" % ci]) and np.array_equal(N(d), g["knn%d_d" % ci]), ci
    rs = np.random.RandomState(9)
    p1 = rs.uniform(-1, 1, (2, 1000, 3)).astype(np.float32); p2 = rs.uniform(-1, 1, (2, 4096, 3)).astype(np.float32)
    for K in (1, 8, 32, 64):
        rd, ri = O.knn_points(p1, p2, K)
        d, i = _ext.knn_points(T(p1, gpu_device), T(p2, gpu_device), K)
        assert np.array_equal(N(i), ri) and np.array_equal(N(d), rd), K
    l2 = np.array([5, 100], np.int64)
    rd, ri = O.knn_points(p1, p2, 8, l2)
    d, i = _ext.knn_points(T(p1, gpu_device), T(p2, gpu_device), 8, torch.from_numpy(l2))
    assert np.array_equal(N(i), ri) and np.array_equal(N(d), rd)
    x = rs.standard_normal((2, 4096, 11)).astype(np.float32)
    assert np.array_equal(N(_ext.knn_gather(T(x, gpu_device), i)), O.knn_gather(x, ri))


def test_error_behaviour(gpu_device):
    from slide_amd import _ext
    with pytest.raises(RuntimeError, match="contiguous"):
        _ext.gather_points(torch.zeros(1, 8, 3, device=gpu_device).transpose(1, 2),
                           torch.zeros(1, 2, dtype=torch.int32, device=gpu_device))
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        _ext.gather_points(torch.zeros(1, 3, 8, device=gpu_device), torch.zeros(1, 2, dtype=torch.int32))
    with pytest.raises(RuntimeError, match="float tensor"):
        _ext.three_nn(torch.zeros(1, 3, 3, device=gpu_device).double(), torch.zeros(1, 3, 3, device=gpu_device))


def test_empty_and_degenerate_inputs(gpu_device):
    """empty batches / zero queries return empty tensors without a launch; nsample > n, K > n and a single point follow
    the oracle (ball_query pads with the first hit, kNN pads missing neighbours with index 0 / distance 0)"""
    from slide_amd import _ext
    dev = gpu_device
    z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=dev)
    assert tuple(_ext.gather_points(z(0, 3, 8), z(0, 4, dt=torch.int32)).shape) == (0, 3, 4)
    assert tuple(_ext.gather_points(z(2, 3, 8), z(2, 0, dt=torch.int32)).shape) == (2, 3, 0)
    assert tuple(_ext.group_points(z(2, 3, 8), z(2, 0, 4, dt=torch.int32)).shape) == (2, 3, 0, 4)
    assert tuple(_ext.furthest_point_sampling(z(0, 16, 3), 4).shape) == (0, 4)
    idx, cnt = _ext.ball_query(z(2, 0, 3), z(2, 5, 3), 0.5, 4)
    assert tuple(idx.shape) == (2, 0, 4) and tuple(cnt.shape) == (2, 0)
    d, i = _ext.three_nn(z(1, 0, 3), z(1, 4, 3))
    assert tuple(d.shape) == (1, 0, 3) and tuple(i.shape) == (1, 0, 3)
    rs = np.random.RandomState(11)
    xyz = rs.uniform(-1, 1, (2, 5, 3)).astype(np.float32)
    q = rs.uniform(-1, 1, (2, 7, 3)).astype(np.float32)
    ri, rc = O.ball_query(q, xyz, 0.8, 16)  # nsample > n
    idx, cnt = _ext.ball_query(T(q, dev), T(xyz, dev), 0.8, 16)
    assert np.array_equal(N(idx), ri) and np.array_equal(N(cnt), rc)
    ri, rc = O.ball_query(q, xyz, 1e-6, 4)  # empty balls
    idx, cnt = _ext.ball_query(T(q, dev), T(xyz, dev), 1e-6, 4)
    assert np.array_equal(N(idx), ri) and np.array_equal(N(cnt), rc) and int(N(cnt).max()) == 0
    rd, ri = O.knn_points(q, xyz, 8)  # K > n2
    d, i = _ext.knn_points(T(q, dev), T(xyz, dev), 8)
    assert np.array_equal(N(i), ri) and np.array_equal(N(d), rd)
    one = rs.uniform(-1, 1, (3, 1, 3)).astype(np.float32)
    assert np.array_equal(N(_ext.furthest_point_sampling(T(one, dev), 1)), O.furthest_point_sampling(one, 1))
    rd, ri = O.three_nn(q, xyz[:, :3])  # exactly three known points
    d, i = _ext.three_nn(T(q, dev), T(xyz[:, :3].copy(), dev))
    assert np.array_equal(N(i), ri) and np.array_equal(N(d), rd)


@pytest.mark.parametrize("shape", [(256, 1024, 256, 32, 128), (64, 8192, 2048, 32, 64)])
def test_ops_full_size_properties(gpu_device, shape):
    """SURVEY.md section 8(d) sizes, where the CPU oracle would take minutes: size-independent properties of every op,
    checked with torch tensor arithmetic on the device (B, N, npoint, nsample, C; coordinates U(-1,1), features N(0,1))"""
    from slide_amd import _ext
    B, n, m, ns, C = shape
    g = torch.Generator(device="cpu").manual_seed(0)
    xyz = (torch.rand(B, n, 3, generator=g) * 2 - 1).to(gpu_device)
    feats = torch.randn(B, C, n, generator=g).to(gpu_device)
    bi = torch.arange(B, device=gpu_device)[:, None]
    # FPS: starts at point 0, never repeats a point, and every pick is the farthest point from the picks before it
    fidx = _ext.furthest_point_sampling(xyz, m).long()
    assert fidx.shape == (B, m) and bool((fidx[:, 0] == 0).all())
    assert bool((torch.sort(fidx, dim=1)[0][:, 1:] != torch.sort(fidx, dim=1)[0][:, :-1]).all())
    new_xyz = xyz[bi, fidx]
    sub = slice(0, 4)  # the greedy criterion on a few clouds (m x n distance matrices)
    d = ((new_xyz[sub, :, None, :] - xyz[sub, None, :, :]) ** 2).sum(-1)              # (4, m, n)
    run_min = torch.cummin(d, dim=1)[0]                                              # distance to the set after each pick
    for j in (1, 2, m // 2, m - 1):
        far = run_min[:, j - 1].max(dim=1)[0]
        picked = run_min[:, j - 1].gather(1, fidx[sub, j:j + 1])[:, 0]
        assert torch.allclose(picked, far, rtol=1e-6, atol=0), j
    # gather / group: exactly the indexed elements
    assert torch.equal(_ext.gather_points(feats, fidx.int()), feats.gather(2, fidx[:, None, :].expand(-1, C, -1)))
    # kNN: sorted, true distances of the returned indices, and nothing outside the list is closer than its last entry
    d2, kidx = _ext.knn_points(new_xyz, xyz, ns, None)
    assert bool((d2[:, :, 1:] >= d2[:, :, :-1]).all()) and int(kidx.min()) >= 0 and int(kidx.max()) < n
    nb = xyz[bi[:, :, None], kidx]                                                   # (B, m, ns, 3)
    assert torch.allclose(((nb - new_xyz[:, :, None, :]) ** 2).sum(-1), d2, rtol=1e-5, atol=1e-7)
    full = ((new_xyz[sub, :64, None, :] - xyz[sub, None, :, :]) ** 2).sum(-1)          # (4, 64, n)
    kth = torch.topk(full, ns, dim=2, largest=False)[0]
    assert torch.allclose(kth, d2[sub, :64], rtol=1e-5, atol=1e-7)
    grouped = _ext.group_points(feats, kidx.int())
    assert grouped.shape == (B, C, m, ns)
    assert torch.equal(grouped[:, :, :, 0], feats.gather(2, kidx[:, None, :, 0].expand(-1, C, -1)))
    assert torch.equal(grouped[:, :, :, ns - 1], feats.gather(2, kidx[:, None, :, ns - 1].expand(-1, C, -1)))
    # ball query: counts within [0, ns], the first `count` slots lie inside the ball, the rest repeat the first hit
    r = 0.12
    bidx, cnt = _ext.ball_query(new_xyz, xyz, r, ns)
    assert int(cnt.min()) >= 1 and int(cnt.max()) <= ns                              # a centre is inside its own ball
    bd = ((xyz[bi[:, :, None], bidx.long()] - new_xyz[:, :, None, :]) ** 2).sum(-1)
    assert bool((bd < r * r).all())
    slot = torch.arange(ns, device=gpu_device)[None, None, :]
    assert bool(((bidx == bidx[:, :, :1]) | (slot < cnt[:, :, None])).all())
    # three_nn / three_interpolate: the three smallest distances, weights that sum to one reproduce constants, linearity
    dist, i3 = _ext.three_nn(xyz, new_xyz)                                           # unknown = all points, known = centres
    assert bool((dist[:, :, 1:] >= dist[:, :, :-1]).all())
    full3 = ((xyz[sub, :64, None, :] - new_xyz[sub, None, :, :]) ** 2).sum(-1)
    assert torch.allclose(torch.topk(full3, 3, dim=2, largest=False)[0], dist[sub, :64], rtol=1e-5, atol=1e-7)
    w = 1.0 / (dist + 1e-8)
    w = w / w.sum(dim=2, keepdim=True)
    kf = torch.randn(B, C, m, generator=g).to(gpu_device)
    kf2 = torch.randn(B, C, m, generator=g).to(gpu_device)
    ones = _ext.three_interpolate(torch.ones(B, 2, m, device=gpu_device), i3, w)
    assert torch.allclose(ones, torch.ones_like(ones), atol=1e-6)
    a, b_, ab = (_ext.three_interpolate(t, i3, w) for t in (kf, kf2, kf + 2 * kf2))
    assert torch.allclose(ab, a + 2 * b_, rtol=1e-5, atol=1e-5)
