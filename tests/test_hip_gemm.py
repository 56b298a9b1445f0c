"""Unit parity of the fused GEMM op (SLIDE_OP_GEMM, every kernel variant the plans use) against a plain numpy fp32
restatement of conv1x1 -> [GroupNorm] -> [ReLU] -> [+vec] -> [+residual] (pointnet2_ops/pointnet2_modules.py:24-176).
Tolerance: fp16 storage of inputs / outputs (2^-11 relative per element) -> 6e-3 * max|ref| absolute."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _Mini:
    """just enough of DenoiserEngine to emit one GEMM op"""

    def __new__(cls, B, prec, device):
        import torch
        from slide_amd import engine as E

        class M(E.DenoiserEngine):
            def __init__(self):
                self.B, self.device, self.prec = B, device, E.PREC[prec]
                self.adt = torch.float16 if self.prec == 1 else torch.float32
                self.A = E._Arena(device); self.ops = []; self.flops = 0; self.gemm_flops = {}; self.gemm_bytes = {}
                self.per_sample_t = True; self.two_lanes = False; self._lane = 0
                self.use_glds = True; self.glds_nst = 0; self.persistent = False
        return M()


def _ref(X, w, bias, npx, mode, layout, gamma, beta, relu, addvec, resid):
    from slide_amd import engine as E
    y = X @ w.T + bias  # [rows][N] logical
    rows, N = y.shape
    if mode == E.EPI_NORM:
        G = min(32, N); n_norm = N - N % G; gs = n_norm // G
        yb = y.reshape(rows // npx, npx, N)
        part = yb[:, :, :n_norm].reshape(rows // npx, npx, G, gs)
        mean = part.mean(axis=(1, 3), keepdims=True)
        var = part.var(axis=(1, 3), keepdims=True)
        part = (part - mean) / np.sqrt(var + 1e-5)
        yb = yb.copy()
        yb[:, :, :n_norm] = part.reshape(rows // npx, npx, n_norm) * gamma[:n_norm] + beta[:n_norm]
        y = yb.reshape(rows, N)
    if relu:
        y = np.maximum(y, 0)
    if addvec is not None:
        y = y + np.repeat(addvec, npx, axis=0)
    if resid is not None:
        y = y + resid
    return y


CASES = [  # rows_per_sample log2, batch, K, N, mode, extras
    (4, 37, 64, 64, 1, ()), (4, 128, 256, 256, 1, ("res", "addvec")), (4, 128, 544, 512, 1, ()), (4, 16, 96, 160, 0, ()),
    (4, 64, 128, 111, 1, ("addvec",)), (7, 64, 160, 288, 1, ("addvec",)), (7, 33, 64, 64, 1, ("res",)),
    (8, 32, 288, 512, 1, ("res", "addvec")), (8, 16, 64, 448, 1, ()), (8, 40, 512, 256, 0, ()), (8, 24, 96, 111, 1, ()),
]


def _to_cm(a):
    """[rows][ld] -> the chunk-major image [ld / 32][rows][32], returned with the logical shape"""
    rows, ld = a.shape
    return np.ascontiguousarray(a.reshape(rows, ld // 32, 32).transpose(1, 0, 2)).reshape(rows, ld)


def _from_cm(a):
    rows, ld = a.shape
    return np.ascontiguousarray(a.reshape(ld // 32, rows, 32).transpose(1, 0, 2)).reshape(rows, ld)


@pytest.mark.parametrize("cm", [False, True])
@pytest.mark.parametrize("npxl,B,K,N,mode,extras", CASES)
def test_gemm_op_matches_numpy(gpu_device, npxl, B, K, N, mode, extras, cm):
    """one GEMM launch with its fused epilogue vs numpy; cm: input, residual and output in chunk-major storage
    ([k / 32][rows][32], the layout of the K-expanded activations of the fp16 plans) -- 128- / 256-row samples only"""
    import torch
    from slide_amd import engine as E
    from slide_amd._lib import check, lib
    if cm and npxl < 7:
        pytest.skip("16-row launches run the small-launch kernel on row-major operands")
    m = _Mini(B, "fp16", gpu_device)
    if cm:
        m.use_cm, m._cm = True, set()
    rs = np.random.RandomState(npxl * 1000 + K + N)
    npx = 1 << npxl
    rows = B * npx
    Xl = rs.standard_normal((rows, K)).astype(np.float32)
    ld = E.ru(K)
    Xp = np.zeros((rows, ld), np.float32); Xp[:, :K] = Xl
    X = m.A.put(_to_cm(Xp) if cm else Xp, m.adt)
    if cm:
        m._cm.add(X.data_ptr())
    w = (rs.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = rs.standard_normal(N).astype(np.float32)
    gamma = (1 + 0.1 * rs.standard_normal(N)).astype(np.float32); beta = (0.1 * rs.standard_normal(N)).astype(np.float32)
    lay = E.gn_layout(N) if mode == E.EPI_NORM else None
    Np = E.ru(lay[1]) if lay else E.ru(N)
    out = m._buf(rows, Np, cm=cm)
    seg = dict(w=w, bias=bias, mode=mode, out=out)
    relu = mode == E.EPI_NORM
    if mode == E.EPI_NORM:
        G = min(32, N); n_norm = N - N % G
        seg.update(flags=E.F_POST_RELU, layout=lay, gn=(gamma[:n_norm], beta[:n_norm]))
    addvec = resid = None
    oidx = lay[0] if lay else np.arange(N)
    if "res" in extras:
        resid = rs.standard_normal((rows, N)).astype(np.float32)
        rp = np.zeros((rows, Np), np.float32); rp[:, oidx] = resid
        seg["residual"] = m.A.put(_to_cm(rp) if cm else rp, m.adt)
        if cm:
            m._cm.add(seg["residual"].data_ptr())
        resid = torch.from_numpy(rp).to(m.adt).float().numpy()[:, oidx]  # what the kernel reads (fp16-rounded)
    if "addvec" in extras:
        addvec = rs.standard_normal((B, N)).astype(np.float32)
        ap = np.zeros((B, Np), np.float32); ap[:, oidx] = addvec
        seg["addvec"] = (m.A.put(ap), 0, Np, None, 0)
    m._gemm(X, npxl, [seg], in_cols=None)
    ops = (E.SlideOp * 1)(*m.ops)
    # the product library carries the 256 x 64 ring tile for 256-row samples (module path) and the small-launch kernel; the
    # 128-row form on stored inputs is a round-2-plan kernel of the experiments build (same tile body and epilogue)
    import contextlib
    from slide_amd import _lib
    if npxl == 7 and not _lib.have_experiments():
        pytest.skip("libslide_hip_exp.so is not built")
    with (_lib.experiments() if npxl == 7 else contextlib.nullcontext()):
        check(lib().slide_run_ops(ops, 1, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "run")
    torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    got = (_from_cm(got) if cm else got)[:, oidx]
    Xr = torch.from_numpy(Xp).to(m.adt).float().numpy()[:, :K]
    wr = torch.from_numpy(w).to(torch.float16).float().numpy()
    ref = _ref(Xr, wr, bias, npx, mode, lay, gamma, beta, relu, addvec, resid)
    assert np.isfinite(got).all()
    tol = 6e-3 * np.abs(ref).max()
    assert np.abs(got - ref).max() <= tol, (np.abs(got - ref).max(), tol)
