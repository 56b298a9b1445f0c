"""GPU: the differentiable row-major layers of the training step (slide_amd/train/functions.py: forward = the module path's HIP
kernels, backward = csrc/train_ops.hip + the forward GEMM on transposed weights) against plain PyTorch fp32 restatements of the same
layers under torch.autograd -- values and every gradient."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.detach() - b.detach()).abs().max() / b.detach().abs().max().clamp_min(1e-12))


def _pad(x, ld):
    return torch.nn.functional.pad(x, (0, ld - x.shape[1])).contiguous()


@pytest.mark.parametrize("rows,I,O,bias", [(37 * 16, 45, 70, True), (5 * 256, 131, 128, True), (3 * 128, 32, 51, False), (16, 515, 256, True)])
def test_conv_rows(gpu_device, rows, I, O, bias):
    from slide_amd.train import functions as F
    g = torch.Generator(device=gpu_device); g.manual_seed(rows + I)
    x = torch.randn(rows, I, device=gpu_device, generator=g)
    w = (torch.randn(O, I, 1, 1, device=gpu_device, generator=g) / np.sqrt(I)).requires_grad_()
    b = torch.randn(O, device=gpu_device, generator=g).requires_grad_() if bias else None
    xr = x.clone().requires_grad_()
    ref = xr @ w.reshape(O, I).t() + (b if bias else 0.0)
    dy = torch.randn(rows, O, device=gpu_device, generator=g)
    gref = torch.autograd.grad(ref, [xr, w] + ([b] if bias else []), dy)
    xp = _pad(x, F.ru(I)).requires_grad_()
    y = F.conv_rows(xp, w, b)
    assert y.shape == (rows, F.ru(O)) and _rel(y[:, :O], ref) <= 2e-5 and float(y[:, O:].abs().max() if y.shape[1] > O else 0) == 0.0
    got = torch.autograd.grad(y, [xp, w] + ([b] if bias else []), _pad(dy, F.ru(O)))
    assert _rel(got[0][:, :I], gref[0]) <= 2e-5 and float(got[0][:, I:].abs().max() if got[0].shape[1] > I else 0) == 0.0
    assert _rel(got[1], gref[1]) <= 1e-4
    if bias:
        assert _rel(got[2], gref[2]) <= 1e-5


@pytest.mark.parametrize("B,S,C,pre,post", [(3, 48, 70, False, True), (2, 256, 128, False, True), (4, 16, 64, True, False), (2, 128, 111, True, True),
                                            (3, 16, 51, False, False), (2, 40, 20, False, True), (2, 256, 515, True, True), (3, 100, 1024, False, True),
                                            (33, 128, 64, False, True)])
def test_group_norm_rows(gpu_device, B, S, C, pre, post):
    """MyGroupNorm(min(32, C), C) (pointnet2_modules.py:24-42): the first C - C % G channels in G groups, the rest pass through"""
    from slide_amd.train import functions as F
    g = torch.Generator(device=gpu_device); g.manual_seed(B * S + C)
    G = min(32, C)
    n_norm = C - C % G
    x = torch.randn(B * S, C, device=gpu_device, generator=g) * 1.5 + 0.3
    gam = (1 + 0.2 * torch.randn(n_norm, device=gpu_device, generator=g)).requires_grad_()
    bet = (0.2 * torch.randn(n_norm, device=gpu_device, generator=g)).requires_grad_()
    xr = x.clone().requires_grad_()
    z = torch.relu(xr) if pre else xr
    zn = z[:, :n_norm].reshape(B, S, n_norm).permute(0, 2, 1)                      # (B, C, S)
    yn = torch.nn.functional.group_norm(zn, G, gam, bet, eps=1e-5).permute(0, 2, 1).reshape(B * S, n_norm)
    ref = torch.cat([yn, z[:, n_norm:]], dim=1)
    ref = torch.relu(ref) if post else ref
    dy = torch.randn(B * S, C, device=gpu_device, generator=g)
    gref = torch.autograd.grad(ref, [xr, gam, bet], dy)
    ld = F.ru(C)
    xp = _pad(x, ld).requires_grad_()
    y = F.gn_rows(xp, gam, bet, B, S, G, pre, post)
    assert _rel(y[:, :C], ref) <= 2e-5 and float(y[:, C:].abs().max() if ld > C else 0) == 0.0
    got = torch.autograd.grad(y, [xp, gam, bet], _pad(dy, ld))
    assert _rel(got[0][:, :C], gref[0]) <= 1e-4, _rel(got[0][:, :C], gref[0])
    assert _rel(got[1], gref[1]) <= 1e-4 and _rel(got[2], gref[2]) <= 1e-4
    # ReLUs only (a layer without GroupNorm)
    y2 = F.gn_rows(xp, None, None, B, S, 0, pre, post)
    r2 = torch.relu(xr) if (pre or post) else xr
    assert _rel(y2[:, :C], r2) <= 1e-6
    d2 = torch.autograd.grad(y2, [xp], _pad(dy, ld))[0]
    assert _rel(d2[:, :C], torch.autograd.grad(r2, [xr], dy)[0]) <= 1e-6


@pytest.mark.parametrize("rows,ld", [(5, 32), (4096, 128), (1000, 544), (70000, 64), (300, 1024)])
def test_col_sums(gpu_device, rows, ld):
    from slide_amd.train import functions as F
    g = torch.Generator(device=gpu_device); g.manual_seed(rows)
    x = torch.randn(rows, ld, device=gpu_device, generator=g)
    ref = x.double().sum(0)
    assert float((F.col_sums(x).double() - ref).abs().max()) <= 1e-5 * float(x.abs().sum(0).max())


@pytest.mark.parametrize("fp", [False, True])
def test_group_rows(gpu_device, fp):
    """QueryAndGroup 'nn' (pointnet2_utils.py:383-430: [feat | rel | abs | centre]) and group_knn (:497-524: [feat | d2 | w | abs | rel
    | centre]): forward channels and the feature gradient (a scatter-add over the neighbour table)"""
    from slide_amd import _ext
    from slide_amd.train import functions as F
    from slide_amd.rows import GROUP_ABS, GROUP_CENTER, GROUP_FP
    g = torch.Generator(device=gpu_device); g.manual_seed(7 + fp)
    B, N, C, K = 5, 16, 51 if fp else 37, 8 if fp else 16
    xyz = torch.randn(B, N, 3, device=gpu_device, generator=g)
    feat = torch.randn(B * N, C, device=gpu_device, generator=g)
    d2, idx = _ext.knn_points(xyz, xyz, K, None)
    fr = feat.clone().requires_grad_()
    nb = torch.gather(fr.reshape(B, N, C), 1, idx.reshape(B, N * K, 1).expand(-1, -1, C)).reshape(B * N * K, C)
    q = torch.gather(xyz, 1, idx.reshape(B, N * K, 1).expand(-1, -1, 3)).reshape(B, N, K, 3)
    ctr = xyz[:, :, None, :].expand(-1, -1, K, -1)
    if fp:
        w = 1.0 / (d2 + 1e-8)
        w = w / w.sum(dim=2, keepdim=True)
        coords = torch.cat([d2[..., None], w[..., None], q, q - ctr, ctr], dim=3)
        flags = GROUP_FP
    else:
        coords = torch.cat([q - ctr, q, ctr], dim=3)
        flags = GROUP_ABS | GROUP_CENTER
    ref = torch.cat([nb, coords.reshape(B * N * K, -1)], dim=1)
    fpad = _pad(feat, F.ru(C)).requires_grad_()
    out = F.group_rows(fpad, xyz, xyz, idx, d2 if fp else None, flags, C)
    W = ref.shape[1]
    assert out.shape == (B * N * K, F.ru(W)) and _rel(out[:, :W], ref) <= 1e-6 and float(out[:, W:].abs().max()) == 0.0
    dy = torch.randn(B * N * K, W, device=gpu_device, generator=g)
    gref = torch.autograd.grad(ref, [fr], dy)[0]
    got = torch.autograd.grad(out, [fpad], _pad(dy, F.ru(W)))[0]
    assert _rel(got[:, :C], gref) <= 1e-5 and float(got[:, C:].abs().max() if got.shape[1] > C else 0) == 0.0


@pytest.mark.parametrize("pts,K,C1,C2", [(5 * 16, 16, 51, 60), (3 * 16, 8, 256, 139), (7, 16, 3, 12)])
def test_concat_qk_and_attend(gpu_device, pts, K, C1, C2):
    from slide_amd.train import functions as F
    g = torch.Generator(device=gpu_device); g.manual_seed(pts + K)
    q = torch.randn(pts, C1, device=gpu_device, generator=g)
    k = torch.randn(pts * K, C2, device=gpu_device, generator=g)
    qr, kr = q.clone().requires_grad_(), k.clone().requires_grad_()
    ref = torch.relu(torch.cat([qr[:, None, :].expand(-1, K, -1).reshape(pts * K, C1), kr], dim=1))
    qp, kp = _pad(q, F.ru(C1)).requires_grad_(), _pad(k, F.ru(C2)).requires_grad_()
    out = F.concat_qk(qp, kp, K, C1, C2)
    W = C1 + C2
    assert _rel(out[:, :W], ref) <= 1e-6 and float(out[:, W:].abs().max() if out.shape[1] > W else 0) == 0.0
    dy = torch.randn(pts * K, W, device=gpu_device, generator=g)
    gref = torch.autograd.grad(ref, [qr, kr], dy)
    got = torch.autograd.grad(out, [qp, kp], _pad(dy, F.ru(W)))
    assert _rel(got[0][:, :C1], gref[0]) <= 1e-5 and _rel(got[1][:, :C2], gref[1]) <= 1e-6
    # softmax over the K neighbours + weighted sum
    C = C2
    s = torch.randn(pts * K, C, device=gpu_device, generator=g) * 2
    v = torch.randn(pts * K, C, device=gpu_device, generator=g)
    sr, vr = s.clone().requires_grad_(), v.clone().requires_grad_()
    wgt = torch.softmax(sr.reshape(pts, K, C), dim=1)
    ref = (wgt * vr.reshape(pts, K, C)).sum(dim=1)
    sp, vp = _pad(s, F.ru(C)).requires_grad_(), _pad(v, F.ru(C)).requires_grad_()
    out = F.attend_rows(sp, vp, K, C)
    assert _rel(out[:, :C], ref) <= 1e-5 and float(out[:, C:].abs().max() if out.shape[1] > C else 0) == 0.0
    do = torch.randn(pts, C, device=gpu_device, generator=g)
    gref = torch.autograd.grad(ref, [sr, vr], do)
    got = torch.autograd.grad(out, [sp, vp], _pad(do, F.ru(C)))
    assert _rel(got[0][:, :C], gref[0]) <= 1e-4 and _rel(got[1][:, :C], gref[1]) <= 1e-5
