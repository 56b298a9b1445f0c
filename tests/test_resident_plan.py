"""CPU: the host side of the LDS-resident denoiser kernel (slide_amd/resident.py) -- op program, arena aliasing, MFMA-fragment
weight packing, column maps -- executed by the numpy interpreter of the kernel's op set (tests/resident_emu.py) and checked
against the reference-generated denoiser golden.  fp16 activation storage: tolerance 1e-2 of max|ref|."""
import json

import numpy as np
import pytest
import torch

from conftest import golden_spec, load_golden
from oracle import denoiser_np as D
from resident_emu import Emu
from slide_amd.engine import DenoiserEngine
from slide_amd.experiments.resident import LDS_LIMIT, R_GEMM, R_TAIL, ResidentPlan
from slide_amd.synth import synth_state_dict

pytestmark = pytest.mark.exp  # (the op program of the experiments build's LDS-resident kernel)


def _vectors(hp, sd, e, ts, label):
    temb = D.calc_t_emb(ts, hp["t_dim"])
    temb = D.swish(D.linear(temb, sd["fc_t1.weight"], sd["fc_t1.bias"]))
    temb = D.swish(D.linear(temb, sd["fc_t2.weight"], sd["fc_t2.bias"]))
    tv = np.concatenate([D.linear(temb, sd[n + ".weight"], sd[n + ".bias"]) for n, _ in e._tvec], axis=1)
    ce = sd["class_emb.weight"][label]
    cv = np.concatenate([D.linear(ce, sd[n + ".weight"], sd[n + ".bias"]) for n, _ in e._cvec], axis=1)
    return tv, cv


def test_resident_plan_emulated_matches_reference():
    g = load_golden("golden_denoiser_pos.npz")
    hp = json.loads(str(g["config_json"]))
    sd = synth_state_dict(golden_spec(g))
    B = 3
    e = DenoiserEngine(hp, sd, B, torch.device("cpu"), prec="fp16", per_sample_t=True)
    plan = ResidentPlan(e)
    assert plan.lds_bytes <= LDS_LIMIT
    # every weight element is used by exactly one wave: the packed pool is the network's conv weights (+ K / channel pads)
    n_conv = sum(v.size for k, v in sd.items() if v.ndim > 1 and not k.startswith(("fc_t", "class_emb")) and ".fc" not in k)
    assert n_conv <= 512 * len(plan.frags) <= 1.6 * n_conv
    for op in plan.ops:
        if op.type in (R_GEMM, R_TAIL):
            assert op.a.nks_gat + op.a.nks_x + op.b.nks_gat + op.b.nks_x <= 16
            assert op.parts in (1, 2, 4) and (op.rows_log2 != 4 or op.parts == 1)
    emu = Emu(plan)
    for k in ["t0", "t1", "t500", "t999", "mixed"]:
        x, ts, label, ref = g["x_" + k], g["ts_" + k], g["label_" + k], g["eps_" + k]
        tv, cv = _vectors(hp, sd, e, ts, label)
        out = np.stack([emu.run(x[b], tv[b], cv[b]) for b in range(B)])
        assert np.isfinite(out).all(), k  # a NaN = the program read a column / row nobody wrote
        err = np.abs(out - ref).max() / np.abs(ref).max()
        assert err <= 1e-2, (k, err)
