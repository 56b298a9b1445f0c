"""Row-major activations of the module-level path (`pointnet2_ops.pointnet2_modules`, `pointnet2_ops.attention`).

The reference moves (B, C, npoint, K) tensors through nn.Conv2d(1x1) / GroupNorm / torch.cat / softmax
(pointnet2_ops_lib/pointnet2_ops/pointnet2_modules.py:71-176, attention.py:35-96).  On MI355X those layers are HBM-bound
and the NCHW layout forces a transpose on either side of every GEMM, so the HIP path keeps a grouped activation as ONE
matrix `Rows.data` [B * S][ld] (S = npoint * K rows per sample, ld = channels rounded up to 32, pad columns zero; fp32,
or fp16 with SLIDE_MODULE_PREC=fp16): the grouping kernel writes it, the MFMA GEMM reads and writes it, GroupNorm / ReLU /
the t- and class-embedding adds / the residual run in place in one pass, and the attention kernel reduces it over K.
Reference-layout tensors exist only at module boundaries (`from_ncx` / `to_ncx`).

Everything here launches kernels of libslide_hip.so (rows_ops.hip + the engine GEMM); there is no CPU fallback."""
import ctypes
import os

import numpy as np
import torch

from ._lib import check, lib
from .engine import EPI_RAW, EPI_STATS, F_OUT_F32, F_PRE_RELU, OP_GEMM, SlideEpi, SlideOp, make_op, ru

OP_COPY_COLS = 7
OP_ROWS_FROM_NCX, OP_ROWS_TO_NCX, OP_ROWS_GROUP, OP_ROWS_GN, OP_ROWS_CONCAT_QK, OP_ROWS_ATTN, OP_ROWS_POOL = 20, 21, 22, 23, 24, 25, 26
OP_ROWS_GN_JOINT = 27
OP_GEMM_ATTEND = 38
OP_ROWS_PAIR_EXPAND = 39
GROUP_FP, GROUP_ABS, GROUP_CENTER, GROUP_NO_XYZ, GROUP_IDX32 = 1, 2, 4, 8, 16
POOL_MAX, POOL_AVG, POOL_MAX_AVG = 0, 1, 2
GN_PRE_RELU, GN_POST_RELU, GN_STATS_ONLY, GN_APPLY_ONLY = 1, 2, 4, 8


def half_mode():
    return os.environ.get("SLIDE_MODULE_PREC", "fp32") == "fp16"


def _run(op):
    arr = (SlideOp * 1)(op)
    check(lib().slide_run_ops(arr, 1, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "slide_run_ops")


def _rop(kind, half, i, p):
    i = tuple(i) + (0,) * (9 - len(i)) + (int(half),)
    return make_op(kind, i=i, p=tuple(None if v is None else (v if isinstance(v, int) else v.data_ptr()) for v in p))


class Rows:
    """[B * S][ld] activation, `C` valid channels"""
    __slots__ = ("data", "B", "S", "C", "stats", "pending")

    def __init__(self, data, B, S, C, stats=None):
        assert data.dim() == 2 and data.shape[0] == B * S and data.shape[1] % 32 == 0 and data.is_contiguous()
        self.data, self.B, self.S, self.C = data, B, S, C
        # (per-tile channel sums, sums of squares, of-relu?) published by the GEMM that produced `data`, or None
        self.stats = stats
        # DEFERRED normalisation: `data` is still the raw convolution output; (scale / shift [B][2][ld] fp32, relu, addvec)
        # stand for relu?(data * scale + shift) + addvec.  A following conv() applies it while it loads (no pass over the
        # tensor); every other consumer materialises it first (`materialise`).
        self.pending = None

    @property
    def ld(self):
        return self.data.shape[1]

    @property
    def half(self):
        return self.data.dtype == torch.float16

    @property
    def rows(self):
        return self.data.shape[0]


class LazyGroup(Rows):
    """The grouped input of an SA / feature-map / kNN-FP block that is NOT built until somebody needs its rows (round 6).  Its only
    consumers in the module programs are 1 x 1 convolutions (first_mlp / res_connect / grouped_feat_conv), and a convolution over
    [features of the neighbour | coordinate channels] separates into a per-SOURCE-point table and coordinate terms (`_pair_conv`): the
    grouped matrix -- the widest tensor of a level, B * npoint * K rows -- is then never written or read.  Any other use (`.data`)
    materialises it with the grouping kernel."""
    __slots__ = ("_buf", "spec", "_ld", "_half")

    def __init__(self, spec, B, S, C, ld, half):
        self._buf, self.spec, self._ld, self._half = None, spec, ld, half
        self.B, self.S, self.C = B, S, C
        self.stats = None
        self.pending = None

    @property
    def data(self):
        if self._buf is None:
            self._buf = _group_now(*self.spec)
        return self._buf

    @property
    def ld(self):
        return self._ld

    @property
    def half(self):
        return self._half

    @property
    def rows(self):
        return self.B * self.S


def _empty(rows, ld, half, device):
    return torch.empty(rows, ld, device=device, dtype=torch.float16 if half else torch.float32)


def from_ncx(x, half=None):
    """(B, C, *spatial) fp32 -> Rows with S = prod(spatial); half: storage type (default: SLIDE_MODULE_PREC)"""
    if not x.is_cuda:
        raise RuntimeError("CPU not supported")
    x = x.contiguous().float()
    B, C = x.shape[:2]
    P = int(np.prod(x.shape[2:])) if x.dim() > 2 else 1
    half = half_mode() if half is None else half
    out = _empty(B * P, ru(C), half, x.device)
    _run(_rop(OP_ROWS_FROM_NCX, half, (B, C, P, out.shape[1]), (x, out)))
    return Rows(out, B, P, C)


def to_ncx(r, spatial=None):
    """Rows -> (B, C, *spatial) fp32"""
    materialise(r)
    out = torch.empty((r.B, r.C) + tuple(spatial if spatial is not None else (r.S,)), device=r.data.device, dtype=torch.float32)
    _run(_rop(OP_ROWS_TO_NCX, r.half, (r.B, r.C, r.S, r.ld), (r.data, out)))
    return out


def from_points(x, half=None):
    """(B, N, C) fp32 point-major tensor -> Rows (dtype conversion + column padding only)"""
    B, N, C = x.shape
    x = x.contiguous().float()
    half = half_mode() if half is None else half
    out = torch.zeros(B * N, ru(C), device=x.device, dtype=torch.float16 if half else torch.float32)
    _run(make_op(OP_COPY_COLS, i=(B * N, C, C, out.shape[1], 0, int(half)), p=(x.data_ptr(), out.data_ptr())))
    return Rows(out, B, N, C)


def to_points(r):
    """Rows -> (B, S, C) fp32"""
    materialise(r)
    out = torch.empty(r.B, r.S, r.C, device=r.data.device, dtype=torch.float32)
    _run(make_op(OP_COPY_COLS, i=(r.rows, r.C, r.ld, r.C, int(r.half), 0), p=(r.data.data_ptr(), out.data_ptr())))
    return out


def concat_cols(parts):
    """channel concatenation of Rows with equal (B, S); a part may also be a (B, S, c) fp32 tensor (coordinates)"""
    for p in parts:
        if isinstance(p, Rows):
            materialise(p)
    first = next(p for p in parts if isinstance(p, Rows))
    B, S, half, dev = first.B, first.S, first.half, first.data.device
    widths = [p.C if isinstance(p, Rows) else p.shape[-1] for p in parts]
    C = sum(widths)
    out = torch.zeros(B * S, ru(C), device=dev, dtype=torch.float16 if half else torch.float32)
    esz = out.element_size()
    c0 = 0
    for p, w in zip(parts, widths):
        if isinstance(p, Rows):
            src, sld, s16 = p.data, p.ld, int(p.half)
        else:
            src, sld, s16 = p.contiguous().float(), w, 0
        _run(make_op(OP_COPY_COLS, i=(B * S, w, sld, out.shape[1], s16, int(half)), p=(src.data_ptr(), out.data_ptr() + c0 * esz)))
        c0 += w
    return Rows(out, B, S, C)


# ----------------------------------------------------------------------------------------------------------- GEMM
class _PinnedUploader:
    """Descriptor tables carry the addresses of a call's output / statistics / per-point tensors, and the caching allocator hands a
    decode pass different addresses often enough that two thirds of the GEMM calls build a fresh table (measured: 489 of 750 in
    tools/time_decode.py).  `torch.from_numpy(...).to(device)` of pageable memory is a SYNCHRONOUS copy: the host stalled once per such
    call and the launch queue drained behind it.  Tables now go through a ring of pinned slots and an asynchronous stream-ordered copy;
    a slot is reused only after the event of its previous copy (SLIDE_PINNED_TABLES=0: the synchronous upload)."""
    SLOT, NSLOT = 64 * 1024, 256

    def __init__(self):
        self.buf = None
        self.events = [None] * self.NSLOT
        self.next = 0

    def upload(self, arr, device):
        n = arr.nbytes
        if n > self.SLOT or os.environ.get("SLIDE_PINNED_TABLES", "1") == "0":
            return torch.from_numpy(arr.copy()).to(device)
        if self.buf is None:
            self.buf = torch.empty(self.SLOT * self.NSLOT, dtype=torch.uint8).pin_memory()
            self.np = self.buf.numpy()
        i = self.next
        self.next = (i + 1) % self.NSLOT
        ev = self.events[i]
        if ev is not None:
            ev.synchronize()  # (256 tables ago: long since done)
        self.np[i * self.SLOT:i * self.SLOT + n] = arr
        dev = torch.empty(n, dtype=torch.uint8, device=device)
        dev.copy_(self.buf[i * self.SLOT:i * self.SLOT + n], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        self.events[i] = ev
        return dev


_uploader = _PinnedUploader()


class _ConvPlan:
    """packed weight + bias of one 1x1 convolution / linear for the row-major GEMM: y[rows, O] = x[rows, I] @ W^T + b"""

    def __init__(self, weight, bias, half, device):
        O, I = weight.shape[0], int(np.prod(weight.shape[1:]))
        self.O, self.I, self.kp, self.op_ = O, I, ru(I), ru(O)
        self.half = half
        W = torch.zeros(self.op_, self.kp, device=device, dtype=torch.float32)
        W[:O, :I] = weight.detach().reshape(O, I).float()
        self.W = W.to(torch.float16 if half else torch.float32)
        self.vec = torch.zeros(self.op_, device=device, dtype=torch.float32)
        if bias is not None:
            self.vec[:O] = bias.detach().float()
        self.epis = {}  # output pointer -> device epilogue table (the caching allocator recycles a handful of addresses)
        # PACKED VECTORS (SLIDE_EPI_PACKED_VECS, include/slide_engine.h): the blocks' [bias | gamma | beta] values behind the descriptors,
        # bit 0 of the pointer set -- descriptors and vectors are then staged by LDS-DMA, without the per-workgroup pointer chase
        pv = np.zeros((self.op_ // 32, 3, 32), np.float32)
        pv[:, 0, :] = self.vec.cpu().numpy().reshape(-1, 32)
        self._pv = pv.reshape(-1).view(np.uint8)
        self._tag = 1 if os.environ.get("SLIDE_PACKED_VECS", "1") != "0" else 0

    def epi(self, out, stats=None, pre_relu=False, pre_add=None, out_f32=False):
        """pre_add = (per-point tensor [points][ld], log2 K): + pre_add[row >> log2 K] before the ReLU (SlideEpi.pre_add)"""
        key = (out.device.index, out.data_ptr(), None if stats is None else (stats[0].data_ptr(), stats[1].data_ptr()), pre_relu,
               None if pre_add is None else (pre_add[0].data_ptr(), pre_add[1]), out_f32)
        e = self.epis.get(key)
        if e is None:
            if len(self.epis) >= 16:
                self.epis.clear()
            n_cob = self.op_ // 32
            esz = out.element_size()
            tab = (SlideEpi * n_cob)()
            for j in range(n_cob):
                t = tab[j]
                t.mode = EPI_RAW if stats is None else EPI_STATS
                t.flags = (F_PRE_RELU if pre_relu else 0) | (F_OUT_F32 if out_f32 else 0)
                t.out_ld = self.op_
                t.bias = self.vec.data_ptr() + 4 * 32 * j
                t.out = out.data_ptr() + esz * 32 * j
                if pre_add is not None:
                    t.pre_add = pre_add[0].data_ptr() + pre_add[0].element_size() * 32 * j
                    t.pre_add_ld = pre_add[0].shape[1]
                    t.pre_add_shift = pre_add[1]
                if stats is not None:
                    t.stats_sum = stats[0].data_ptr() + 4 * 32 * j
                    t.stats_sq = stats[1].data_ptr() + 4 * 32 * j
                    t.stats_bs = self.op_
                    t.stats_scale = 1.0
            e = _uploader.upload(np.concatenate([np.frombuffer(bytes(tab), dtype=np.uint8), self._pv]), out.device)
            self.epis[key] = e
        return e

    def epi_ptr(self, *args, **kw):
        """the pointer an op carries: the table of epi() with the packed-vectors bit"""
        return self.epi(*args, **kw).data_ptr() | self._tag

    def run(self, x, stats=None, pre_add=None, out_f32=False):
        """stats: None | "raw" | "relu" -- also publish per-256-row-tile channel sums of the output (of its ReLU) from the
        GEMM epilogue, for the GroupNorm that follows (saves its statistics pass); only when tiles do not straddle samples.
        pre_add = (Rows of per-point terms, K): + term[row // K] ahead of the ReLU (K a power of two)"""
        assert x.ld == self.kp and x.half == self.half, (x.ld, self.kp, x.half, self.half)
        pa = None
        if pre_add is not None:
            pr, K = pre_add
            assert pr.pending is None and pr.half == self.half and pr.rows * K == x.rows and pr.ld >= self.op_ and K & (K - 1) == 0
            pa = (pr.data, K.bit_length() - 1)
        rows = x.rows
        out = _empty(rows, self.op_, self.half and not out_f32, x.data.device)  # (out_f32: fp16 operands, float output rows)
        if rows == 0:  # empty batch: nothing to launch
            return Rows(out, x.B, x.S, self.O)
        st = None
        if stats is not None and x.S % 256 == 0 and fused_stats():
            st = (torch.empty(rows // 256, self.op_, device=out.device), torch.empty(rows // 256, self.op_, device=out.device))
        n_cob = self.op_ // 32
        ntr = (rows + 255) // 256
        cbw = 4 if (self.half and n_cob >= 4 and ntr * ((n_cob + 3) // 4) >= 256 and os.environ.get("SLIDE_MODULE_CBW4", "0") != "0") else 2
        sc = sh = add = None
        f = (0.0, 0.0, 0.0, 0.0)
        in_bs = 0
        if x.pending is not None:  # deferred normalisation of the producing layer, applied to the fragments
            ss, relu, addvec = x.pending
            sc, sh, in_bs = ss.data_ptr(), ss.data_ptr() + 4 * x.ld, 2 * x.ld
            add = None if addvec is None else addvec.data_ptr()
            f = (0.0, float(x.S // 256), float(addvec.shape[1]) if addvec is not None else 0.0,
                 float(2 * (addvec.shape[1] if addvec is not None else 0) + int(relu)))
        _run(make_op(OP_GEMM, i=(rows, self.kp, self.kp, n_cob, 8, in_bs, int(self.half), cbw, int(self.half), 0), f=f,
                     p=(x.data.data_ptr(), self.W.data_ptr(), self.epi_ptr(out, st, stats == "relu", pa, out_f32), sc, sh,
                        None, None, None, None, None, None, add)))
        return Rows(out, x.B, x.S, self.O, stats=None if st is None else (st[0], st[1], stats == "relu"))


def sample_sums(v):
    """per-sample column sums of v and v * v, v float [B, S, C] -> ([B, C], [B, C]), as a PAIRWISE TREE of elementwise additions: the
    order of every output's additions depends on S alone.  (torch.sum picks its reduction split from the whole tensor's shape, so a
    shape's statistics -- and through a near-tie of a later FPS / kNN its decoded cloud -- depended on the batch it was decoded in:
    found by tests/test_hip_cli.py::test_decode_is_sharded_over_the_ranks, round 6.)"""
    x = torch.cat([v, v * v], 2)
    while x.shape[1] > 1:
        S = x.shape[1]
        h = S // 2
        y = x[:, :h] + x[:, h:2 * h]
        x = torch.cat([y, x[:, 2 * h:]], 1) if S & 1 else y  # (odd: the last row is carried)
    C = v.shape[2]
    return x[:, 0, :C].contiguous(), x[:, 0, C:].contiguous()


def fused_stats():
    return os.environ.get("SLIDE_MODULE_STATS", "1") != "0"


def deferral():
    return half_mode() and os.environ.get("SLIDE_MODULE_DEFER", "1") != "0"


def materialise(x):
    """applies a deferred normalisation (one in-place pass); no-op otherwise"""
    if x is not None and x.pending is not None:
        ss, relu, addvec = x.pending
        x.pending = None
        _run(_rop(OP_ROWS_GN, x.half, (x.B, x.S, x.ld, 1, 0, GN_APPLY_ONLY | (GN_POST_RELU if relu else 0),
                                       addvec.shape[1] if addvec is not None else 0, 0, 0),
                  (x.data, None, None, addvec, None, None, x.data, None, None, ss)))
    return x


class WeightSlice:
    """a column slice [c0, c1) of a 1x1 convolution's weight as a layer of its own for conv() (bias: the module's, or none) -- the two
    halves of AttentionModule.weight_conv.2 over the virtual concatenation [q | k]"""

    def __init__(self, module, c0, c1, with_bias):
        self.module, self.c0, self.c1, self.with_bias = module, c0, c1, with_bias

    @property
    def weight(self):
        w = self.module.weight
        return w.reshape(w.shape[0], -1)[:, self.c0:self.c1]

    @property
    def bias(self):
        return self.module.bias if self.with_bias else None


def conv(x, module, stats=None, pre_add=None):
    """HipConv1x1 / HipLinear applied to Rows (weights re-packed when the parameter changes); stats, pre_add: see _ConvPlan.run.
    A deferred normalisation of x is applied by the GEMM itself when its tiles do not straddle samples."""
    if isinstance(x, LazyGroup) and x._buf is None and pre_add is None:
        return _pair_conv(x, module, stats)
    if x.pending is not None and not (x.half and x.S % 256 == 0):
        materialise(x)
    w, bias = module.weight, module.bias
    dev = x.data.device
    # (the packed weights, bias vector and epilogue tables live on ONE device and snapshot BOTH parameters: the key carries the
    #  bias' version / storage and the device as well -- ADVICE r2)
    key = (w._version, w.data_ptr(), None if bias is None else (bias._version, bias.data_ptr()), x.half, dev.type, dev.index)
    plan = module.__dict__.get("_rows_plan")
    if plan is None or plan[0] != key:
        plan = (key, _ConvPlan(w, module.bias, x.half, x.data.device))
        module.__dict__["_rows_plan"] = plan
    return plan[1].run(x, stats, pre_add)


def joint_norm_qk(q, k, K, gn):
    """GroupNorm `gn` over the VIRTUAL concatenation [q(point) x K | k(point, neighbour)] from the tile sums the two producing GEMMs
    published (conv(..., stats="relu")): leaves q and k with a deferred normalisation each (scale / shift in their own layouts) and
    never builds the concatenation (SLIDE_OP_ROWS_GN_JOINT).  Returns False when the producers' statistics are not there."""
    if q.stats is None or k.stats is None or not q.stats[2] or not k.stats[2] or q.pending is not None or k.pending is not None:
        return False
    B = q.B
    dev = q.data.device
    ssq = torch.empty(B * 2 * q.ld, device=dev, dtype=torch.float32)
    ssk = torch.empty(B * 2 * k.ld, device=dev, dtype=torch.float32)
    G, n_norm = gn.num_groups, gn.num_channels
    tq, tk = (q.S // 256 if q.S % 256 == 0 else 1), k.S // 256  # (a sample of fewer rows than a tile: ONE row of sums, see attention.py)
    op = make_op(OP_ROWS_GN_JOINT, i=(B, q.C, q.ld, tq, K, k.C, k.ld, tk, G, 0), f=(1.0 / k.S, float(n_norm)),
                 p=(q.stats[0].data_ptr(), q.stats[1].data_ptr(), k.stats[0].data_ptr(), k.stats[1].data_ptr(),
                    gn.weight.data_ptr(), gn.bias.data_ptr(), ssq.data_ptr(), ssk.data_ptr()))
    _run(op)
    q.pending, k.pending = (ssq, False, None), (ssk, False, None)
    q.stats = k.stats = None
    return True


# ----------------------------------------------------------------------------------------------------------- fused layers
def norm_act(x, gn=None, pre_relu=False, relu=False, addvec=None, residual=None, defer=False):
    """in place: x <- relu?(GroupNorm?(relu?(x))) + addvec[b] + residual.  gn: HipGroupNorm (num_groups over its first
    num_channels channels, the rest pass through) or None.
    defer=True (the caller knows the next consumer is a conv()): only the statistics and the per-sample scale / shift are
    computed; the tensor stays raw and the consumer GEMM normalises it while loading (x.pending).  pre_relu then requires that
    the stored values are already the ReLU'd ones (a conv(..., stats="relu") output, or a concat_qk output)."""
    materialise(residual)
    assert x.pending is None
    if gn is None and not (pre_relu or relu or addvec is not None or residual is not None):
        return x
    G, n_norm = (gn.num_groups, gn.num_channels) if gn is not None else (0, 0)
    st = x.stats if (G and x.stats is not None and x.stats[2] == bool(pre_relu)) else None
    x.stats = None  # (the data changes below)
    # scratch: partial sums [B][64][ld][2] (unused with producer-side statistics) + scale / shift [B][2][ld]
    part = torch.empty(x.B * 64 * x.ld * 2 + x.B * 2 * x.ld, device=x.data.device, dtype=torch.float32) if G else None
    if addvec is not None:
        addvec = addvec.contiguous().float()
        assert addvec.shape[0] == x.B and addvec.shape[1] <= x.ld
    if residual is not None:
        assert residual.rows == x.rows and residual.half == x.half and residual.ld >= x.ld
    flags = (GN_PRE_RELU if pre_relu else 0) | (GN_POST_RELU if relu else 0)
    ss = None
    if defer and G and residual is None and deferral() and x.S % 256 == 0 and (st is not None or not pre_relu):
        ss = torch.empty(x.B * 2 * x.ld, device=x.data.device, dtype=torch.float32)
        flags |= GN_STATS_ONLY
    _run(_rop(OP_ROWS_GN, x.half, (x.B, x.S, x.ld, G, n_norm, flags, addvec.shape[1] if addvec is not None else 0,
                                   residual.ld if residual is not None else 0, x.S // 256 if st is not None else 0),
              (x.data, gn.weight if G else None, gn.bias if G else None, addvec, residual.data if residual is not None else None,
               part, x.data, None if st is None else st[0], None if st is None else st[1], ss)))
    if ss is not None:
        x.pending = (ss, bool(relu), addvec)
    return x


def _counts32(counts, pts):
    if counts is None or isinstance(counts, str):
        return None
    c = counts.reshape(-1).to(torch.int32).contiguous()
    assert c.numel() == pts
    return c


def group(xyz, new_xyz, feat, idx, flags, d2=None, empty_counts=None, half=None):
    """grouped input of an SA / feature-map block (QueryAndGroup) or of a kNN feature-propagation block (group_knn):
    xyz (B,N,3), new_xyz (B,np,3), feat Rows [B*N] or None, idx (B,np,K) int64 (kNN) or int32 (ball query) -> Rows
    [B*np*K].  empty_counts (B,np): centres with count 0 become their own single neighbour with zero features."""
    B, N = xyz.shape[:2]
    npnt, K = idx.shape[1:]
    C = feat.C if feat is not None else 0
    ncoord = 11 if flags & GROUP_FP else 0 if flags & GROUP_NO_XYZ else 3 + (3 if flags & GROUP_ABS else 0) + (3 if flags & GROUP_CENTER else 0)
    half = feat.half if feat is not None else (half_mode() if half is None else half)
    assert idx.dtype in (torch.int64, torch.int32)
    idx = idx.contiguous()
    if idx.dtype == torch.int32:
        flags |= GROUP_IDX32
    spec = (xyz.contiguous().float(), new_xyz.contiguous().float(), feat, idx, flags, d2.contiguous() if d2 is not None else None,
            empty_counts, half)
    if (half and empty_counts is None and not (flags & GROUP_NO_XYZ) and B * npnt * K > 0 and ru(C + ncoord) <= 1024
            and os.environ.get("SLIDE_MODULE_PAIR", "1") != "0"):
        return LazyGroup(spec, B, npnt * K, C + ncoord, ru(C + ncoord), half)
    return Rows(_group_now(*spec), B, npnt * K, C + ncoord)


def _group_now(xyz, new_xyz, feat, idx, flags, d2, empty_counts, half):
    """the grouped matrix [B * npoint * K][ld] (SLIDE_OP_ROWS_GROUP)"""
    materialise(feat)
    B, N = xyz.shape[:2]
    npnt, K = idx.shape[1:]
    C = feat.C if feat is not None else 0
    ncoord = 11 if flags & GROUP_FP else 0 if flags & GROUP_NO_XYZ else 3 + (3 if flags & GROUP_ABS else 0) + (3 if flags & GROUP_CENTER else 0)
    out = _empty(B * npnt * K, ru(C + ncoord), half, xyz.device)
    _run(_rop(OP_ROWS_GROUP, half, (B, N, npnt, K, C, feat.ld if feat is not None else 8, out.shape[1], flags),
              (xyz, new_xyz, feat.data if feat is not None else None, idx, d2, out, _counts32(empty_counts, B * npnt))))
    return out


def _pair_coef(w2, C, flags, op_):
    """coefficients of the coordinate terms of a convolution over a grouped input, fp32 [op_][8]: (W_rel + W_abs | W_centre - W_rel | w_d2 |
    w_w) per output channel, from the weight columns behind the C feature columns (layouts: rows_group_kernel, rows_ops.hip)"""
    O = w2.shape[0]
    z = torch.zeros(O, 3, device=w2.device, dtype=torch.float32)
    wd = ww = torch.zeros(O, device=w2.device, dtype=torch.float32)
    if flags & GROUP_FP:   # [feat | d2 | w | abs | rel | centre]
        wd, ww = w2[:, C], w2[:, C + 1]
        w_abs, w_rel, w_ctr = w2[:, C + 2:C + 5], w2[:, C + 5:C + 8], w2[:, C + 8:C + 11]
    else:                  # [feat | rel | abs? | centre?]
        w_rel = w2[:, C:C + 3]
        c = C + 3
        w_abs = w2[:, c:c + 3] if flags & GROUP_ABS else z
        c += 3 if flags & GROUP_ABS else 0
        w_ctr = w2[:, c:c + 3] if flags & GROUP_CENTER else z
    coef = torch.zeros(op_, 8, device=w2.device, dtype=torch.float32)
    coef[:O, 0:3] = w_rel + w_abs
    coef[:O, 3:6] = w_ctr - w_rel
    coef[:O, 6] = wd
    coef[:O, 7] = ww
    return coef.contiguous()


def _pair_conv(x, module, stats):
    """conv(group(...), module, stats) WITHOUT the grouped matrix (LazyGroup above; SLIDE_OP_ROWS_PAIR_EXPAND): the convolution's feature
    columns applied per SOURCE point (one small GEMM -> fp32 table), expanded over the neighbour table with the coordinate terms in fp32,
    ReLU and the per-tile channel sums of the following GroupNorm on the way.  Reference: QueryAndGroup / group_knn feeding
    Mlp_plus_t_emb.first_mlp / res_connect and AttentionModule.grouped_feat_conv (pointnet2_utils.py:383-430, :497-524;
    pointnet2_modules.py:119-176; attention.py:70-85)."""
    xyz, new_xyz, feat, idx, flags, d2, _, _ = x.spec
    w, bias = module.weight, module.bias
    dev = xyz.device
    C = feat.C if feat is not None else 0
    w2 = w.detach().reshape(w.shape[0], -1).float()
    assert w2.shape[1] == x.C, (w2.shape, x.C)
    O, op_ = w2.shape[0], ru(w2.shape[0])
    key = (w._version, w.data_ptr(), None if bias is None else (bias._version, bias.data_ptr()), flags & ~GROUP_IDX32, C, dev.type, dev.index)
    plan = module.__dict__.get("_rows_pair_plan")
    if plan is None or plan[0] != key:
        fplan = _ConvPlan(w2[:, :C], None, True, dev) if C > 0 else None
        bvec = torch.zeros(op_, device=dev, dtype=torch.float32)
        if bias is not None:
            bvec[:O] = bias.detach().float()
        plan = (key, fplan, bvec, _pair_coef(w2, C, flags, op_))
        module.__dict__["_rows_pair_plan"] = plan
    _, fplan, bvec, coef = plan
    B, N = xyz.shape[:2]
    # the per-source-point table, fp32 [B * N][op_]: W_features . features (fp16 operands, as the grouped GEMM had them) + the neighbour's
    # coordinate term (W_rel + W_abs) . xyz in fp32
    cq = coef[:, 0:3].t()
    if fplan is not None:
        if feat.pending is not None and not (feat.half and feat.S % 256 == 0):
            materialise(feat)
        A = fplan.run(feat, out_f32=True).data
        A.addmm_(xyz.reshape(B * N, 3), cq)
    else:
        A = torch.mm(xyz.reshape(B * N, 3), cq)
    npnt, K = idx.shape[1:]
    rows = B * npnt * K
    out = _empty(rows, op_, True, dev)
    st = None
    if stats is not None and x.S % 256 == 0 and fused_stats():
        st = (torch.empty(rows // 256, op_, device=dev), torch.empty(rows // 256, op_, device=dev))
    kflags = (1 if stats == "relu" else 0) | (2 if flags & GROUP_FP else 0) | (16 if flags & GROUP_IDX32 else 0)
    _run(_rop(OP_ROWS_PAIR_EXPAND, True, (B, N, npnt, K, op_, op_, kflags),
              (A, bvec, coef, xyz, new_xyz, idx, d2, out, None if st is None else st[0], None if st is None else st[1])))
    return Rows(out, B, npnt * K, O, stats=None if st is None else (st[0], st[1], stats == "relu"))


def gather_rows(feat, idx):
    """feat Rows [B*N], idx (B, m) int -> Rows [B*m]: rows idx of every sample"""
    B, m = idx.shape
    z = torch.zeros(B, max(feat.S, m), 3, device=feat.data.device)
    return group(z[:, :feat.S], z[:, :m], feat, idx.long().reshape(B, m, 1).contiguous(), GROUP_NO_XYZ)


def concat_qk(q, k, K):
    """relu([q(point) broadcast over the K neighbours | k(point, neighbour)]) -> Rows [rows of k]"""
    materialise(q), materialise(k)
    assert k.rows == q.rows * K and q.half == k.half
    C = q.C + k.C
    out = _empty(k.rows, ru(C), k.half, k.data.device)
    _run(_rop(OP_ROWS_CONCAT_QK, k.half, (k.rows, K, q.C, q.ld, k.C, k.ld, out.shape[1]), (q.data, k.data, out)))
    return Rows(out, k.B, k.S, C)


def attend(scores, values, K, counts=None):
    """softmax over the K neighbour rows of each point (the first max(1, count) of them when counts (B, np) is given),
    weighted sum of the values -> Rows [B * S / K]"""
    materialise(scores)
    vss, v_relu = None, False
    if values.pending is not None and values.pending[2] is None:  # deferred GroupNorm (+ ReLU) of the values: applied here
        vss, v_relu, _ = values.pending
        values.pending = None
    else:
        materialise(values)
    assert scores.rows == values.rows and scores.C == values.C and scores.half == values.half
    pts = scores.rows // K
    out = _empty(pts, ru(scores.C), scores.half, scores.data.device)
    _run(_rop(OP_ROWS_ATTN, scores.half, (pts, K, scores.C, scores.ld, values.ld, out.shape[1], scores.S // K, int(v_relu)),
              (scores.data, values.data, out, _counts32(counts, pts), vss)))
    return Rows(out, scores.B, scores.S // K, scores.C)


def conv_attend(u, module, values, K, counts=None):
    """attend(conv(u, module), values, K, counts) with the score map kept on chip (SLIDE_OP_GEMM_ATTEND, round 6): the score
    convolution's 256 x 64 output tile is soft-maxed over the neighbour rows and contracted with the value rows in the GEMM's epilogue.
    fp16 rows, K in (4, 8, 16, 32), tiles that do not straddle samples; otherwise (or SLIDE_MODULE_FUSE_ATTEND=0) the two launches."""
    ok = (u.half and K in (4, 8, 16, 32) and u.rows % 256 == 0 and u.rows == values.rows and u.rows > 0 and
          os.environ.get("SLIDE_MODULE_FUSE_ATTEND", "1") != "0" and (u.pending is None or u.S % 256 == 0) and u.ld <= 1536)
    if not ok:
        return attend(conv(u, module), values, K, counts)
    w, bias = module.weight, module.bias
    dev = u.data.device
    key = (w._version, w.data_ptr(), None if bias is None else (bias._version, bias.data_ptr()), u.half, dev.type, dev.index)
    plan = module.__dict__.get("_rows_plan")
    if plan is None or plan[0] != key:
        plan = (key, _ConvPlan(w, module.bias, u.half, dev))
        module.__dict__["_rows_plan"] = plan
    cp = plan[1]
    assert u.ld == cp.kp and values.C == cp.O and values.half
    vss, v_relu = None, False
    if values.pending is not None and values.pending[2] is None:  # deferred GroupNorm (+ ReLU) of the values: applied in the epilogue
        vss, v_relu, _ = values.pending
        values.pending = None
    else:
        materialise(values)
    pts = u.rows // K
    out = _empty(pts, cp.op_, True, dev)
    assert values.ld >= cp.op_
    sc = sh = add = None
    f = (0.0, 0.0, 0.0, 0.0)
    in_bs = 0
    if u.pending is not None:
        ss, relu, addvec = u.pending
        sc, sh, in_bs = ss.data_ptr(), ss.data_ptr() + 4 * u.ld, 2 * u.ld
        add = None if addvec is None else addvec.data_ptr()
        f = (0.0, float(u.S // 256), float(addvec.shape[1]) if addvec is not None else 0.0,
             float(2 * (addvec.shape[1] if addvec is not None else 0) + int(relu)))
    c32 = _counts32(counts, pts)
    _run(make_op(OP_GEMM_ATTEND, i=(u.rows, cp.kp, cp.kp, cp.op_ // 32, K, in_bs, values.ld, out.shape[1], u.S // K, int(v_relu), cp.O), f=f,
                 p=(u.data.data_ptr(), cp.W.data_ptr(), cp.epi_ptr(out), sc, sh, values.data.data_ptr(), out.data_ptr(),
                    None if c32 is None else c32.data_ptr(), None if vss is None else vss.data_ptr(), None, None, add)))
    return Rows(out, u.B, u.S // K, cp.O)


def pool(x, K, mode, counts=None):
    """max / mean / [max | mean] over the K neighbour rows of each point -> Rows [B * S / K]"""
    materialise(x)
    pts = x.rows // K
    out = _empty(pts, x.ld, x.half, x.data.device)
    _run(_rop(OP_ROWS_POOL, x.half, (pts, K, x.C, x.ld, out.shape[1], mode), (x.data, out, _counts32(counts, pts))))
    return Rows(out, x.B, x.S // K, x.C)
