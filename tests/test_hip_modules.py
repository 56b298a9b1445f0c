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
