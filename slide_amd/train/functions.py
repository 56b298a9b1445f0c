"""Differentiable row-major layers: forward = the module path's HIP kernels, backward = csrc/train_ops.hip + the forward GEMM on the
transposed weights.  Activations are fp32 CUDA tensors [rows, ld], ld a multiple of 32, pad columns ZERO (every Function keeps
them zero).  The GEMMs run in the split mode (fp32 storage, contractions as two-term fp16 operand splits: fp32-grade results on
the fp16 matrix pipe, include/slide_engine.h SLIDE_PREC_SPLIT)."""
import ctypes
import weakref

import numpy as np
import torch

from .._lib import check, lib
from ..engine import EPI_RAW, OP_GEMM, SlideEpi, make_op, ru
from ..rows import (GN_POST_RELU, GN_PRE_RELU, GROUP_ABS, GROUP_CENTER, GROUP_FP, OP_ROWS_ATTN, OP_ROWS_CONCAT_QK, OP_ROWS_GN,
                    OP_ROWS_GROUP, _rop, _run)

PREC_SPLIT = 2
_c = ctypes.c_void_p


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else _c(t.data_ptr())


def _chk(x):
    if not x.is_cuda:
        raise RuntimeError("CPU not supported: the training layers launch HIP kernels")
    assert x.dim() == 2 and x.dtype == torch.float32 and x.shape[1] % 32 == 0 and x.is_contiguous(), (x.shape, x.dtype)


def pad_cols(x, ld=None):
    """[rows, C] -> [rows, ld] (zero pad columns); differentiable (torch ops)"""
    ld = ru(x.shape[1]) if ld is None else ld
    if x.shape[1] == ld:
        return x.contiguous()
    return torch.nn.functional.pad(x, (0, ld - x.shape[1])).contiguous()


_EPI_CACHE = {}   # (bias-vector address, blocks, out_ld) -> (device epilogue-table template, mask of its `out` words): a step's table is
                  # template + mask * output address, one small device add -- an upload would be a blocking host -> device copy
_PACK_CACHE = {}  # id(parameter) -> persistent packed buffers (refilled in place every call: the weights change every step)


def _packs(weight, kp, op_):
    # (the shape is part of the key: a new parameter can inherit the id of a collected one, and the buffers' pad regions must stay zero)
    key = (id(weight), tuple(weight.shape), kp, op_, weight.device.index)
    c = _PACK_CACHE.get(key)
    if c is None:
        if len(_PACK_CACHE) > 4096:
            # (never evict a LIVE parameter's buffers: their addresses are baked into captured step graphs -- ADVICE r4; a model has a
            #  few hundred entries, and entries die with their parameter, below)
            raise RuntimeError("slide_amd.train: more than 4096 packed weight buffers alive -- parameters are being re-created every step")
        dev = weight.device
        c = {"W": torch.zeros(op_, kp, device=dev), "Wt": torch.zeros(kp, op_, device=dev), "vec": torch.zeros(op_, device=dev),
             "zero": torch.zeros(kp, device=dev)}
        _PACK_CACHE[key] = c
        # the entry lives exactly as long as its parameter: a collected parameter's buffers and epilogue tables are released,
        # and a recycled id() can never pick up buffers whose addresses an older captured graph holds
        weakref.finalize(weight, _drop_pack, key)
    return c


def _drop_pack(key):
    c = _PACK_CACHE.pop(key, None)
    if c is not None:
        addr = c["vec"].data_ptr()
        for k in [k for k in _EPI_CACHE if k[0] == addr]:
            _EPI_CACHE.pop(k, None)


def _gemm(x, w_packed, bias_vec, n_out_pad):
    """y [rows, n_out_pad] = x [rows, kp] @ w_packed[n_out_pad, kp]^T + bias_vec (split-precision MFMA GEMM, RAW epilogue)"""
    rows, kp = x.shape
    out = torch.empty(rows, n_out_pad, device=x.device, dtype=torch.float32)
    if rows == 0:
        return out
    n_cob = n_out_pad // 32
    key = (bias_vec.data_ptr(), n_cob, n_out_pad, x.device.index)
    c = _EPI_CACHE.get(key)
    if c is None:
        if len(_EPI_CACHE) > 8192:
            raise RuntimeError("slide_amd.train: more than 8192 epilogue tables alive (their addresses are baked into captured graphs)")
        tab = (SlideEpi * n_cob)()
        for j in range(n_cob):
            t = tab[j]
            t.mode = EPI_RAW
            t.out_ld = n_out_pad
            t.bias = bias_vec.data_ptr() + 4 * 32 * j
            t.out = 4 * 32 * j
        words = ctypes.sizeof(SlideEpi) // 8
        mask = np.zeros((n_cob, words), np.int64)
        mask[:, SlideEpi.out.offset // 8] = 1
        c = (torch.from_numpy(np.frombuffer(bytes(tab), dtype=np.int64).reshape(n_cob, words).copy()).to(x.device),
             torch.from_numpy(mask).to(x.device))
        _EPI_CACHE[key] = c
    epi = torch.add(c[0], c[1], alpha=out.data_ptr())
    # small launches (fewer than 256 of the 256-row tiles): the 64-row-tile kernel (the RAW epilogue does not look at samples, so the
    # "16 rows per sample" form that selects it is only a tile-shape request)
    npx = 4 if ((rows + 255) // 256) * ((n_cob + 1) // 2) < 256 else 8  # (measured: 0 / 256 / 1024 / always -- 256 is best at batch 32 and 256)
    _run(make_op(OP_GEMM, i=(rows, kp, kp, n_cob, npx, 0, PREC_SPLIT, 2, 0, 0),
                 p=(x.data_ptr(), w_packed.data_ptr(), epi.data_ptr())))
    return out


def col_sums(x):
    """sum over the rows of x [rows, ld] -> [ld] (csrc/train_ops.hip col_sums_kernel: coalesced row chunks, then the same kernel over
    the partial rows).  torch's dim-0 reduction of a 65536 x 128 matrix runs at 4 % of the HBM rate, and its multi-block reductions
    (semaphore + staging buffer) returned garbage inside replayed HIP graphs (tools/debug_train_nan4.py)."""
    rows, ld = x.shape
    buf = torch.empty(1025 if rows >= 128 else 1, ld, device=x.device, dtype=torch.float32)
    check(lib().slide_col_sums(ctypes.c_longlong(rows), ld, _p(x), _p(buf), _c(buf.data_ptr() + 4 * ld) if rows >= 128 else None, _stream()),
          "slide_col_sums")
    return buf[0]


def _weight_grad(dy, x):
    """dW (padded) [op_, kp] = dy^T x over the rows: a library GEMM (hipBLASLt through torch) whose contraction is the LONG dimension
    (up to 65536 rows against 32..512 outputs), so it is split into row slabs -- a batched GEMM that fills the chip -- and the
    slab results are added."""
    rows = dy.shape[0]
    slabs = 1
    while slabs < 64 and rows % (slabs * 2) == 0 and rows // (slabs * 2) >= 256:
        slabs *= 2
    if slabs == 1:
        return dy.t() @ x
    return torch.bmm(dy.view(slabs, rows // slabs, -1).transpose(1, 2), x.view(slabs, rows // slabs, -1)).sum(dim=0)


class ConvRows(torch.autograd.Function):
    """nn.Conv2d(1x1) / nn.Conv1d(1) / nn.Linear on rows: y[:, :O] = x[:, :I] @ W^T + b, pad columns zero"""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _chk(x)
        O, I = weight.shape[0], int(np.prod(weight.shape[1:]))
        kp, op_ = x.shape[1], ru(O)
        assert kp == ru(I), (kp, I)
        c = _packs(weight, kp, op_)
        c["W"][:O, :I].copy_(weight.detach().reshape(O, I))
        if bias is not None:
            c["vec"][:O].copy_(bias.detach())
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return _gemm(x, c["W"], c["vec"], op_)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        O, I = weight.shape[0], int(np.prod(weight.shape[1:]))
        kp, op_ = x.shape[1], dy.shape[1]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:  # dx = dy @ W: the forward GEMM on the transposed weights
            c = _packs(weight, kp, op_)
            c["Wt"][:I, :O].copy_(weight.detach().reshape(O, I).t())
            dx = _gemm(dy, c["Wt"], c["zero"], kp)
        if ctx.needs_input_grad[1]:
            dw = _weight_grad(dy, x)[:O, :I].reshape(weight.shape)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = col_sums(dy)[:O]
        return dx, dw, db


class GroupNormRows(torch.autograd.Function):
    """post_relu?(MyGroupNorm(pre_relu?(x))) over the S rows of each of the B samples: G groups over the first len(gamma) channels,
    the rest pass through (pointnet2_modules.py:24-42).  gamma None: ReLUs only."""

    @staticmethod
    def forward(ctx, x, gamma, beta, B, S, G, pre_relu, post_relu):
        _chk(x)
        assert x.shape[0] == B * S
        ld = x.shape[1]
        n_norm = 0 if gamma is None else gamma.shape[0]
        G = G if n_norm else 0
        flags = (GN_PRE_RELU if pre_relu else 0) | (GN_POST_RELU if post_relu else 0)
        out = torch.empty_like(x)
        gam = bet = part = mr = None
        if n_norm:
            gam, bet = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
            part = torch.empty(B * 64 * ld * 2 + B * 2 * ld, device=x.device, dtype=torch.float32)
            mr = torch.empty(B, 64, 2, device=x.device, dtype=torch.float32)  # the statistics, for the backward
        if x.shape[0]:
            _run(_rop(OP_ROWS_GN, False, (B, S, ld, G, n_norm, flags, 0, 0, 0), (x, gam, bet, None, None, part, out, None, None, None, mr)))
        ctx.save_for_backward(x, gam, bet, mr)
        ctx.cfg = (B, S, G, n_norm, flags)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, gam, bet, mr = ctx.saved_tensors
        B, S, G, n_norm, flags = ctx.cfg
        ld = x.shape[1]
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dg = db = scratch = None
        if n_norm:
            dg, db = torch.empty(B, ld, device=x.device), torch.empty(B, ld, device=x.device)
            scratch = torch.empty(B * (64 * ld * 2 + 128), device=x.device, dtype=torch.float32)
        if x.shape[0]:
            check(lib().slide_gn_rows_bwd(B, S, ld, G, n_norm, flags, _p(x), _p(gam), _p(bet), _p(mr), _p(dy), _p(dx), _p(dg), _p(db),
                                          _p(scratch), _stream()), "slide_gn_rows_bwd")
        elif n_norm:
            dg.zero_(), db.zero_()
        return (dx, None if dg is None else col_sums(dg)[:n_norm], None if db is None else col_sums(db)[:n_norm], None, None, None, None, None)


class GroupRows(torch.autograd.Function):
    """grouped input of an SA block (QueryAndGroup 'nn': [feat | rel | abs | centre]) or of a kNN feature-propagation block
    (group_knn: [feat | d2 | w | abs | rel | centre]) as rows [(b, p, k)]; differentiable in the FEATURES only (the coordinates are
    network inputs, pointnet2_utils.py:383-430, :497-524)"""

    @staticmethod
    def forward(ctx, feat, xyz, new_xyz, idx, d2, flags, C):
        _chk(feat)
        B, N = xyz.shape[:2]
        npnt, K = idx.shape[1:]
        ncoord = 11 if flags & GROUP_FP else 3 + (3 if flags & GROUP_ABS else 0) + (3 if flags & GROUP_CENTER else 0)
        out = torch.empty(B * npnt * K, ru(C + ncoord), device=feat.device, dtype=torch.float32)
        idx = idx.contiguous()
        assert idx.dtype == torch.int64 and feat.shape[0] == B * N
        _run(_rop(OP_ROWS_GROUP, False, (B, N, npnt, K, C, feat.shape[1], out.shape[1], flags),
                  (xyz.contiguous().float(), new_xyz.contiguous().float(), feat, idx, None if d2 is None else d2.contiguous(), out, None)))
        ctx.save_for_backward(idx)
        ctx.cfg = (B, N, npnt, K, C, feat.shape[1], out.shape[1])
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        B, N, npnt, K, C, ldf, ldg = ctx.cfg
        dout = dout.contiguous()
        dfeat = torch.zeros(B * N, ldf, device=dout.device, dtype=torch.float32)
        check(lib().slide_group_rows_bwd(B, N, npnt, K, C, ldf, ldg, _p(idx), None, _p(dout), _p(dfeat), _stream()), "slide_group_rows_bwd")
        return dfeat, None, None, None, None, None, None


class ConcatQK(torch.autograd.Function):
    """relu([q(point) broadcast over the K neighbours | k(point, neighbour)])  (attention.py:78-88)"""

    @staticmethod
    def forward(ctx, q, k, K, C1, C2):
        _chk(q), _chk(k)
        assert k.shape[0] == q.shape[0] * K
        out = torch.empty(k.shape[0], ru(C1 + C2), device=k.device, dtype=torch.float32)
        if k.shape[0]:
            _run(_rop(OP_ROWS_CONCAT_QK, False, (k.shape[0], K, C1, q.shape[1], C2, k.shape[1], out.shape[1]), (q, k, out)))
        ctx.save_for_backward(out)
        ctx.cfg = (K, C1, q.shape[1], C2, k.shape[1])
        return out

    @staticmethod
    def backward(ctx, dout):
        (out,) = ctx.saved_tensors
        K, C1, ldq, C2, ldk = ctx.cfg
        dout = dout.contiguous()
        pts = out.shape[0] // K
        dq = torch.zeros(pts, ldq, device=out.device, dtype=torch.float32)
        dk = torch.zeros(out.shape[0], ldk, device=out.device, dtype=torch.float32)
        check(lib().slide_concat_qk_bwd(ctypes.c_longlong(pts), K, C1, ldq, C2, ldk, out.shape[1], _p(out), _p(dout), _p(dq), _p(dk), _stream()),
              "slide_concat_qk_bwd")
        return dq, dk, None, None, None


class AttendRows(torch.autograd.Function):
    """softmax over the K neighbour rows of each point, weighted sum of the values (attention.py:89-95; 'nn' grouping: all K count)"""

    @staticmethod
    def forward(ctx, scores, values, K, C):
        _chk(scores), _chk(values)
        assert scores.shape[0] == values.shape[0]
        pts = scores.shape[0] // K
        out = torch.empty(pts, ru(C), device=scores.device, dtype=torch.float32)
        if pts:
            _run(_rop(OP_ROWS_ATTN, False, (pts, K, C, scores.shape[1], values.shape[1], out.shape[1], 1, 0), (scores, values, out, None, None)))
        ctx.save_for_backward(scores, values)
        ctx.cfg = (K, C, out.shape[1])
        return out

    @staticmethod
    def backward(ctx, dout):
        scores, values = ctx.saved_tensors
        K, C, ldo = ctx.cfg
        dout = dout.contiguous()
        pts = scores.shape[0] // K
        ds, dv = torch.zeros_like(scores), torch.zeros_like(values)
        check(lib().slide_attn_rows_bwd(ctypes.c_longlong(pts), K, C, scores.shape[1], values.shape[1], ldo, _p(scores), _p(values), None,
                                        _p(dout), _p(ds), _p(dv), _stream()), "slide_attn_rows_bwd")
        return ds, dv, None, None


def conv_rows(x, weight, bias=None):
    return ConvRows.apply(x, weight, bias)


def gn_rows(x, gamma, beta, B, S, G, pre_relu=False, post_relu=False):
    return GroupNormRows.apply(x, gamma, beta, B, S, G, pre_relu, post_relu)


def group_rows(feat, xyz, new_xyz, idx, d2, flags, C):
    return GroupRows.apply(feat, xyz, new_xyz, idx, d2, flags, C)


def concat_qk(q, k, K, C1, C2):
    return ConcatQK.apply(q, k, K, C1, C2)


def attend_rows(scores, values, K, C):
    return AttendRows.apply(scores, values, K, C)
