"""GPU: the module-level drop-in API (`pointnet2_ops.*`, `PointNet2CloudCondition`) on HIP kernels vs golden vectors
produced by the REFERENCE modules."""
import json
import os
import sys

import numpy as np
import pytest
import torch

from conftest import REPO, golden_spec, load_golden
from slide_amd.synth import synth_state_dict

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _load(mod, g, prefix, tag, dev):
    spec = golden_spec(g, prefix)
    vals = synth_state_dict([(tag + n, s) for n, s in spec])
    mod.load_state_dict({n: torch.from_numpy(vals[tag + n]) for n, _ in spec})
    return mod.to(dev).eval()


def test_blocks_match_reference_modules(gpu_device):
    from pointnet2_ops import pointnet2_modules as PM
    from pointnet2_ops import pointnet2_utils as PU
    from pointnet2_ops.attention import AttentionModule
    g = load_golden("golden_blocks.npz")
    d = gpu_device
    xyz, feats = T(g["xyz"], d), T(g["feats"], d)
    fidx = PU.furthest_point_sample(xyz, g["fps_idx"].shape[1])
    assert np.array_equal(fidx.cpu().numpy(), g["fps_idx"])
    new_xyz = PU.gather_operation(xyz.transpose(1, 2).contiguous(), fidx).transpose(1, 2).contiguous()
    assert np.array_equal(new_xyz.cpu().numpy(), g["new_xyz"])
    o, c = PU.QueryAndGroup(0, 8, True, True, True, "nn")(xyz, new_xyz, feats, subset=True, return_counts=True)
    assert np.array_equal(o.cpu().numpy(), g["qg_nn"]) and np.array_equal(c.cpu().numpy(), g["qg_nn_counts"])
    qr = PU.QueryAndGroup(0.6, 8, True, True, False, "radius")
    o, c = qr(xyz, new_xyz, feats, subset=True, return_counts=True)
    assert np.array_equal(o.cpu().numpy(), g["qg_radius"]) and np.array_equal(c.cpu().numpy(), g["qg_radius_counts"])
    o, c = qr(xyz, T(g["q2"], d), feats, subset=False, return_counts=True)
    assert np.allclose(o.cpu().numpy(), g["qg_radius_nosubset"], atol=1e-7)
    assert np.allclose(PU.group_knn(new_xyz, xyz, feats, 6, transpose=True).cpu().numpy(), g["group_knn"], rtol=1e-5, atol=1e-6)
    dist, i3 = PU.three_nn(xyz, new_xyz)
    assert np.array_equal(i3.cpu().numpy(), g["three_nn_idx"]) and np.allclose(dist.cpu().numpy(), g["three_nn_dist"], atol=1e-6)
    C = feats.shape[1]
    fp = _load(PM.PointnetFPModule(mlp=[C + 7, 16, 16], bn=True, include_t=False, bias=True, res_connect=True), g, "fp_spec", "fpmod.", d)
    out = fp(xyz, new_xyz, T(g["fp_unknown_feats"], d), T(g["fp_known_feats"], d))
    assert np.allclose(out.cpu().numpy(), g["fp_out"], atol=5e-5)
    att = {"use_attention_module": True, "attention_bn": True, "transform_grouped_feat_out": True, "last_activation": True}
    sa = _load(PM.PointnetSAModule(mlp=[C, 16, 16, 32], npoint=12, radius=0, nsample=8, bn=True, use_xyz=True, t_dim=24,
                                   include_t=True, include_abs_coordinate=True, include_center_coordinate=True, bias=True,
                                   res_connect=True, include_condition=True, condition_dim=10, neighbor_def="nn",
                                   attention_setting=att), g, "sa_spec", "samod.", d)
    nx, nf = sa(xyz, feats, t_emb=T(g["sa_t_emb"], d), condition_emb=T(g["sa_cond_emb"], d))
    assert np.array_equal(nx.cpu().numpy(), g["sa_new_xyz"]) and np.allclose(nf.cpu().numpy(), g["sa_new_features"], atol=5e-5)
    am = _load(AttentionModule(C, C + 6, C, C + 6, 32, True, True, True), g, "att_spec", "attmod.", d)
    a = am(T(g["att_query"], d), T(g["qg_radius"], d), T(g["att_grouped_feat_out"], d), T(g["qg_radius_counts"], d))
    assert np.allclose(a.cpu().numpy(), g["att_out"], atol=5e-5)


@pytest.mark.parametrize("name", ["pos", "feat"])
def test_denoiser_module_path_and_state_dict(gpu_device, name):
    sys.path.insert(0, os.path.join(REPO, "pointnet2"))
    from models.pointnet2_with_pcld_condition import PointNet2CloudCondition
    g = load_golden("golden_denoiser_%s.npz" % name)
    hp = json.loads(str(g["config_json"]))
    spec = golden_spec(g)
    net = PointNet2CloudCondition(hp)
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == dict(spec)  # checkpoint compatibility
    sd = synth_state_dict(spec)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net = net.to(gpu_device).eval()
    for k in ["t0", "t999", "mixed"]:
        x, ts, lab = T(g["x_" + k], gpu_device), T(g["ts_" + k], gpu_device), T(g["label_" + k], gpu_device)
        y = net(x, ts=ts, label=lab).cpu().numpy()
        ref = g["eps_" + k]
        assert np.abs(y - ref).max() <= 2e-4 * np.abs(ref).max(), k
        yf = net(x, ts=ts, label=lab, fused=True).cpu().numpy()
        assert np.abs(yf - ref).max() <= 2e-4 * np.abs(ref).max(), k


def test_autoencoder_decode_matches_reference(gpu_device):
    """config 5: latents -> 256 -> 1024 -> 2048 x 6 on the HIP module path vs the reference's decode (FPS start index 0)."""
    sys.path.insert(0, os.path.join(REPO, "pointnet2"))
    from models.autoencoder import PointAutoencoder
    from oracle import denoiser_np as D
    g = load_golden("golden_decode.npz")
    decs = json.loads(str(g["decoder_configs_json"]))
    spec = golden_spec(g)
    ae = PointAutoencoder(None, decs, apply_kl_regularization=True)
    assert {k: tuple(v.shape) for k, v in ae.state_dict().items()} == dict(spec)  # decode-side checkpoint compatibility
    vals = synth_state_dict([("ae." + n, s) for n, s in spec])
    ae.load_state_dict({n: torch.from_numpy(vals["ae." + n]) for n, _ in spec})
    ae = ae.to(gpu_device).eval()
    d = gpu_device
    kp, feat, lab = T(g["keypoint"], d), T(g["feature"], d), T(g["label"], d)
    B = kp.shape[0]
    start = torch.zeros(B, dtype=torch.int32, device=d)
    l1 = ae.keypoint_encoder.upsample_points(feat, kp, start)
    assert np.abs(l1.cpu().numpy() - g["level1"]).max() <= 1e-5
    # level by level, each fed with the reference's previous level (FPS selection order is perturbation-fragile)
    f2, l2 = ae.decoder.decoders[0](kp, feat, T(g["level1"], d), label=lab, fps_start_idx=start)
    f3, l3 = ae.decoder.decoders[1](T(g["level1"], d)[:, :, :3].contiguous(), f2, T(g["level2"], d), label=lab, fps_start_idx=start)
    for b in range(B):
        err, bij = D.match_point_sets(l2[b].cpu().numpy(), g["level2"][b])
        assert bij and err <= 1e-4, ("level2", err)
        err, bij = D.match_point_sets(l3[b].cpu().numpy(), g["level3"][b])
        assert bij and err <= 1e-4, ("level3", err)
    # end to end (own previous levels, own FPS order): Chamfer vs the reference cloud
    full = ae.decode(kp, feat, label=lab, fps_start_idx=start).cpu().numpy()
    assert full.shape == (B, 2048, 6)
    cd = max(D.chamfer(full[b], g["level3"][b]) for b in range(B))
    print("decode end-to-end Chamfer (sum of both directions, squared) vs reference: %.3e" % cd)
    assert cd <= 1e-5
    # random-start variant runs and stays on the same surface
    full_r = ae.decode(kp, feat, label=lab).cpu().numpy()
    assert max(D.chamfer(full_r[b], g["level3"][b]) for b in range(B)) <= 1e-3


def test_autoencoder_encode_matches_reference(gpu_device):
    """2048 x 6 cloud -> PointNet2Encoder (FPS 1024/256/64/32, kNN-32 SA stack) -> key-point encoder ->
    (B,16,48) latent features, HIP module path vs the reference's `PointAutoencoder.encode` (posterior mode).  FPS runs on
    the INPUT coordinates only, so the selected points are bit-exact and the features compare element-wise."""
    sys.path.insert(0, os.path.join(REPO, "pointnet2"))
    from models.autoencoder import PointAutoencoder
    g = load_golden("golden_encode.npz")
    enc, decs = json.loads(str(g["encoder_config_json"])), json.loads(str(g["decoder_configs_json"]))
    spec = golden_spec(g)
    ae = PointAutoencoder(enc, decs, apply_kl_regularization=True)
    have = {k: tuple(v.shape) for k, v in ae.state_dict().items()}
    assert {k: v for k, v in have.items() if k.startswith(("encoder.", "keypoint_encoder."))} == dict(spec)  # checkpoint keys
    vals = synth_state_dict([("ae." + n, s) for n, s in spec])
    ae.load_state_dict({n: torch.from_numpy(vals["ae." + n]) for n, _ in spec}, strict=False)
    ae = ae.to(gpu_device).eval()
    d = gpu_device
    pc, kp, lab = T(g["pointcloud"], d), T(g["keypoint"], d), T(g["label"], d)
    out, l_xyz, _ = ae.encoder(pc, ts=None, label=lab)
    assert np.array_equal(l_xyz[-1].cpu().numpy(), g["encoder_xyz_last"])  # four FPS levels, bit-exact selections
    ref = g["encoder_out"]
    assert np.abs(out.cpu().numpy() - ref).max() <= 2e-4 * np.abs(ref).max()
    feat = ae.encode(pc, kp, ts=None, label=lab, sample_posterior=False).cpu().numpy()
    ref = g["feature_at_keypoint"]
    assert feat.shape == ref.shape == (pc.shape[0], 16, 48)
    assert np.abs(feat - ref).max() <= 2e-4 * np.abs(ref).max(), np.abs(feat - ref).max()
    # the sampled posterior differs from the mode by std * N(0,1) only
    fs = ae.encode(pc, kp, ts=None, label=lab, sample_posterior=True).cpu().numpy()
    assert np.isfinite(fs).all() and fs.shape == ref.shape
    # encode -> decode round trip runs end to end on the module path
    rec = ae.decode(kp, T(feat, d), label=lab, fps_start_idx=torch.zeros(pc.shape[0], dtype=torch.int32, device=d))
    assert rec.shape == (pc.shape[0], 2048, 6) and bool(torch.isfinite(rec).all())


def test_module_path_fp16_operands(gpu_device, monkeypatch):
    """SLIDE_MODULE_PREC=fp16 (throughput mode of the module-level path: fp16 GEMM operands, fp32 accumulate / outputs):
    encode within 5e-3 relative L2 of the reference, FPS selections unchanged (they depend on the input coordinates only)."""
    monkeypatch.setenv("SLIDE_MODULE_PREC", "fp16")
    sys.path.insert(0, os.path.join(REPO, "pointnet2"))
    from models.autoencoder import PointAutoencoder
    g = load_golden("golden_encode.npz")
    enc, decs = json.loads(str(g["encoder_config_json"])), json.loads(str(g["decoder_configs_json"]))
    spec = golden_spec(g)
    ae = PointAutoencoder(enc, decs, apply_kl_regularization=True)
    vals = synth_state_dict([("ae." + n, s) for n, s in spec])
    ae.load_state_dict({n: torch.from_numpy(vals["ae." + n]) for n, _ in spec}, strict=False)
    ae = ae.to(gpu_device).eval()
    d = gpu_device
    pc, kp, lab = T(g["pointcloud"], d), T(g["keypoint"], d), T(g["label"], d)
    out, l_xyz, _ = ae.encoder(pc, ts=None, label=lab)
    assert np.array_equal(l_xyz[-1].cpu().numpy(), g["encoder_xyz_last"])
    feat = ae.encode(pc, kp, ts=None, label=lab, sample_posterior=False).cpu().numpy()
    ref = g["feature_at_keypoint"]
    rel = float(np.linalg.norm(feat - ref) / np.linalg.norm(ref))
    assert rel <= 5e-3, rel


def test_sample_farthest_points(gpu_device):
    from oracle import ops as O
    from slide_amd import _ext
    rs = np.random.RandomState(4)
    for (B, n, K) in [(2, 512, 256), (3, 2048, 1024), (1, 4096, 2048), (2, 100, 100)]:
        p = rs.uniform(-1, 1, (B, n, 6)).astype(np.float32)
        p[:, 5] = p[:, 1]
        start = rs.randint(0, n, B).astype(np.int32)
        ro, ri = O.sample_farthest_points(p, K, start)
        go, gi = _ext.sample_farthest_points(T(p, gpu_device), K=K, start_idx=torch.from_numpy(start))
        assert np.array_equal(gi.cpu().numpy(), ri) and np.array_equal(go.cpu().numpy(), ro)


def _ab(monkeypatch, fn):
    """run fn() on the row-major fast path and on the general NCHW program"""
    monkeypatch.setenv("SLIDE_MODULE_ROWS", "1")
    a = fn()
    monkeypatch.setenv("SLIDE_MODULE_ROWS", "0")
    b = fn()
    monkeypatch.delenv("SLIDE_MODULE_ROWS")
    return a, b


@pytest.mark.parametrize("prec", ["fp32", "fp16"])
def test_row_major_path_equals_general_program(gpu_device, monkeypatch, prec):
    """The row-major fast path (slide_amd.rows) and the general NCHW program are two implementations of the same
    reference modules: same outputs on random weights -- multi-scale grouping, fewer points than npoint (no FPS), odd
    channel counts (pad columns, un-normalised GroupNorm tail channels), K larger than a wave of rows, no attention bn."""
    from pointnet2_ops import pointnet2_modules as PM
    monkeypatch.setenv("SLIDE_MODULE_PREC", prec)
    tol = dict(atol=2e-4, rtol=1e-4) if prec == "fp32" else dict(atol=3e-2, rtol=3e-2)
    d = gpu_device
    gen = torch.Generator().manual_seed(5)
    B, N, C = 3, 200, 13
    xyz = torch.rand(B, N, 3, generator=gen).to(d)
    feats = torch.randn(B, C, N, generator=gen).to(d)
    temb, cemb = torch.randn(B, 24, generator=gen).to(d), torch.randn(B, 10, generator=gen).to(d)

    def randomise(m):
        g2 = torch.Generator().manual_seed(11)
        with torch.no_grad():
            for n, p in m.named_parameters():
                p.copy_(torch.randn(p.shape, generator=g2) * (0.3 if p.dim() > 1 else 0.5) + (1.0 if n.endswith("group_norm.weight") else 0.0))
        return m.to(d).eval()

    for att_bn, last_act in ((True, True), (False, False)):
        att = {"use_attention_module": True, "attention_bn": att_bn, "transform_grouped_feat_out": True, "last_activation": last_act}
        sa = randomise(PM.PointnetSAModuleMSG(npoint=50, radii=[0, 0], nsamples=[5, 48], mlps=[[C, 20, 20, 45], [C, 16, 24, 40, 33]],
                                              bn=True, use_xyz=True, t_dim=24, include_t=True, include_abs_coordinate=True,
                                              include_center_coordinate=False, bias=True, res_connect=True, include_condition=True,
                                              condition_dim=10, neighbor_def="nn", attention_setting=att))
        (ax, af), (bx, bf) = _ab(monkeypatch, lambda: sa(xyz, feats, t_emb=temb, condition_emb=cemb))
        assert torch.equal(ax, bx) and af.shape == (B, 45 + 33, 50)
        assert torch.allclose(af, bf, **tol), float((af - bf).abs().max())
    # fewer points than npoint: every point is a centre, the query features are the input features
    sa = randomise(PM.PointnetSAModule(mlp=[C, 16, 16, 32], npoint=512, radius=0, nsample=16, bn=True, use_xyz=True,
                                       include_t=False, include_abs_coordinate=True, include_center_coordinate=True, bias=False,
                                       res_connect=False, neighbor_def="nn", attention_setting=att))
    (ax, af), (bx, bf) = _ab(monkeypatch, lambda: sa(xyz, feats))
    assert ax.shape == (B, N, 3) and torch.allclose(af, bf, **tol), float((af - bf).abs().max())
    # kNN feature propagation: 40 known points -> 200 unknown points, skip features, t / class embeddings
    att = {"use_attention_module": True, "attention_bn": True, "transform_grouped_feat_out": True, "last_activation": True}
    known, kf = xyz[:, :40].contiguous(), torch.randn(B, 21, 40, generator=gen).to(d)
    fp = randomise(PM.PointnetKnnFPModule(mlp1=[21, 24, 24, 30], mlp2=[30 + C, 36, 36], K=7, bn=True, t_dim=24, include_t=True,
                                          bias=True, res_connect=True, include_condition=True, condition_dim=10,
                                          include_second_condition=True, second_condition_dim=10, attention_setting=att))
    a, b = _ab(monkeypatch, lambda: fp(xyz, known, feats, kf, t_emb=temb, condition_emb=cemb, second_condition_emb=cemb))
    assert a.shape == (B, 36, N) and torch.allclose(a, b, **tol), float((a - b).abs().max())
    # feature mapper of the autoencoder: features of 200 points mapped onto 16 key points
    fm = randomise(PM.FeatureMapModule([C, 32, 32, 48], 0, 12, use_xyz=True, include_abs_coordinate=True, include_center_coordinate=True,
                                       bn=True, bn_first=False, bias=True, res_connect=True, neighbor_def="nn",
                                       attention_setting=att, query_feature_dim=19))
    keypts, qf = xyz[:, 100:116].contiguous(), torch.randn(B, 19, 16, generator=gen).to(d)
    a, b = _ab(monkeypatch, lambda: fm(xyz, feats, keypts, subset=False, record_neighbor_stats=True, features_at_new_xyz=qf))
    assert a.shape == (B, 48, 16) and torch.allclose(a, b, **tol), float((a - b).abs().max())
