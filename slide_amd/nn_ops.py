"""HIP-backed building blocks of the module-level compatibility path (`pointnet2_ops.pointnet2_modules`):
1x1 convolutions / linears on the MFMA GEMM and GroupNorm, on reference-layout (NCHW) tensors.  Default: exact fp32 MFMA
(the parity mode every module-level test uses).  `SLIDE_MODULE_PREC=fp16` switches the GEMM operands (weights, the
transposed activation copy) to fp16 with fp32 accumulation and fp32 outputs -- the throughput mode of the encode / decode
paths.  Inference only.  The fused, layout-optimised path for the DDPM configs is slide_amd.engine.DenoiserEngine."""
import ctypes
import os

import numpy as np
import torch
import torch.nn as nn

from ._lib import check, lib
from .engine import EPI_RAW, F_OUT_F32, OP_GEMM, SlideEpi, SlideOp, make_op, ru

OP_GROUPNORM_NCHW = 13
OP_TRANSPOSE = 15


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _run(op):
    arr = (SlideOp * 1)(op)
    check(lib().slide_run_ops(arr, 1, _stream()), "slide_run_ops")


class _GemmPlan:
    """packed weight + RAW epilogue table for y[rows, O] = x[rows, I] @ W^T + b (fp32 MFMA)"""

    def __init__(self, weight, bias, rows, device):
        O, I = weight.shape[0], int(np.prod(weight.shape[1:]))
        self.O, self.I, self.rows = O, I, rows
        self.kp, self.op_ = ru(I), ru(O)
        self.half = os.environ.get("SLIDE_MODULE_PREC", "fp32") == "fp16"
        adt = torch.float16 if self.half else torch.float32
        W = torch.zeros(self.op_, self.kp, device=device, dtype=torch.float32)
        W[:O, :I] = weight.detach().reshape(O, I).float()
        self.W = W.to(adt)
        vec = torch.zeros(self.op_, device=device, dtype=torch.float32)
        if bias is not None:
            vec[:O] = bias.detach().float()
        self.vec = vec
        self.x = torch.zeros(rows, self.kp, device=device, dtype=adt)
        self.y = torch.zeros(rows, self.op_, device=device, dtype=torch.float32)
        n_cob = self.op_ // 32
        epis = (SlideEpi * n_cob)()
        for j in range(n_cob):
            e = epis[j]
            e.mode = EPI_RAW
            e.flags = F_OUT_F32 if self.half else 0
            e.out_ld = self.op_
            e.bias = vec.data_ptr() + 4 * 32 * j
            e.out = self.y.data_ptr() + 4 * 32 * j
        self.epi = torch.from_numpy(np.frombuffer(bytes(epis), dtype=np.uint8).copy()).to(device)
        ntr = (rows + 255) // 256
        cbw = 4 if (self.half and n_cob >= 4 and ntr * ((n_cob + 3) // 4) >= 256 and os.environ.get("SLIDE_MODULE_CBW4", "0") != "0") else 2
        self.op = make_op(OP_GEMM, i=(rows, self.kp, self.kp, n_cob, 8, 0, int(self.half), cbw, int(self.half), 0),
                          p=(self.x.data_ptr(), self.W.data_ptr(), self.epi.data_ptr(), None, None))

    def __call__(self, x2d):
        self.x[:, : self.I].copy_(x2d)
        _run(self.op)
        return self.y[:, : self.O]


def _bias_key(bias):
    """plan-cache key part: a packed plan snapshots the bias too (an in-place bias update alone must re-pack)"""
    return None if bias is None else (bias._version, bias.data_ptr())


class HipConv1x1(nn.Module):
    """nn.Conv2d / nn.Conv1d with kernel size 1 (same parameter names and shapes), forward on the HIP GEMM."""

    def __init__(self, in_channels, out_channels, bias=True, ndim=2):
        super().__init__()
        self.in_channels, self.out_channels, self.ndim = in_channels, out_channels, ndim
        self.weight = nn.Parameter(torch.empty((out_channels, in_channels) + (1,) * ndim))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        self._plans = {}

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("CPU not supported")
        B, C = x.shape[:2]
        sp = x.shape[2:]
        rows = B * int(np.prod(sp)) if len(sp) else B
        key = (rows, self.weight._version, self.weight.data_ptr(), _bias_key(self.bias), x.device.index)
        if key not in self._plans:
            self._plans = {key: _GemmPlan(self.weight, self.bias, rows, x.device)}
        plan = self._plans[key]
        P = rows // B
        if P > 1 and x.dtype == torch.float32 and x.is_contiguous() and B * C * P < 2 ** 31 and rows * plan.op_ < 2 ** 31:
            # NCHW -> [pixel][channel] and back with the LDS-tiled transpose kernel (torch's strided copies took
            # half of the decode path's time)
            _run(make_op(OP_TRANSPOSE, i=(B, C, P, P, plan.kp, C * P, P * plan.kp, int(plan.half)), p=(x.data_ptr(), plan.x.data_ptr())))
            _run(plan.op)
            out = torch.empty((B, self.out_channels) + tuple(sp), device=x.device, dtype=torch.float32)
            _run(make_op(OP_TRANSPOSE, i=(B, P, self.out_channels, plan.op_, P, P * plan.op_, self.out_channels * P),
                         p=(plan.y.data_ptr(), out.data_ptr())))
            return out
        x2 = x.reshape(B, C, -1).permute(0, 2, 1).reshape(rows, C)
        y = plan(x2)
        # .clone(): when the permute is a no-op view (one pixel per sample), .contiguous() would alias the plan's persistent
        # output buffer, which the next call with the same row count overwrites
        return y.reshape(B, -1, self.out_channels).permute(0, 2, 1).reshape((B, self.out_channels) + tuple(sp)).clone(memory_format=torch.contiguous_format)


class HipLinear(nn.Module):
    def __init__(self, in_features, out_features):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        self.bias = nn.Parameter(torch.zeros(out_features))
        self._plans = {}

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("CPU not supported")
        key = (x.shape[0], self.weight._version, self.weight.data_ptr(), _bias_key(self.bias), x.device.index)
        if key not in self._plans:
            self._plans = {key: _GemmPlan(self.weight, self.bias, x.shape[0], x.device)}
        return self._plans[key](x).clone()


class HipGroupNorm(nn.Module):
    """nn.GroupNorm(num_groups, num_channels) parameter-compatible; `total_channels` > num_channels passes the tail
    channels through (MyGroupNorm); optional fused ReLU."""

    def __init__(self, num_groups, num_channels, eps=1e-5):
        super().__init__()
        assert abs(eps - 1e-5) < 1e-12
        self.num_groups, self.num_channels = num_groups, num_channels
        self.weight = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))

    def forward(self, x, relu=False):
        if not x.is_cuda:
            raise RuntimeError("CPU not supported")
        x = x.contiguous().float()
        B, C = x.shape[:2]
        HW = int(np.prod(x.shape[2:])) if x.dim() > 2 else 1
        y = torch.empty_like(x)
        _run(make_op(OP_GROUPNORM_NCHW, i=(B, C, HW, self.num_groups, self.num_channels, int(relu)),
                     p=(x.data_ptr(), self.weight.data_ptr(), self.bias.data_ptr(), y.data_ptr())))
        return y
