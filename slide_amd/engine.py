"""Host side of the fused denoiser engine (include/slide_engine.h).

`DenoiserEngine` turns a reference `pointnet_config` + a reference-named state dict
 into a *plan*: packed MFMA weights, per-32-channel epilogue tables and a
flat list of `SlideOp` launches that computes PointNet2CloudCondition.forward
(pointnet2/models/pointnet2_with_pcld_condition.py:286-489) for the configuration family every shipped
DDPM config uses (16 latent points, 'nn' grouping, kNN feature propagation, attention, GroupNorm,
bias, res_connect, no condition cloud).  The plan is replayed eagerly or from a captured hipGraph.

torch is used for device memory and streams only; all arithmetic runs in libslide_hip.so.
"""
import ctypes
import os

import numpy as np
import torch

from ._lib import SlideHipError, check, lib

EPI_RAW, EPI_NORM, EPI_STATS = 0, 1, 2
F_PRE_RELU, F_POST_RELU, F_OUT_F32, F_RES_PAIR, F_RES_PAIR_NBR, F_OUT_FM = 1, 2, 4, 8, 16, 32
# "split": the fp32 plan (float storage, same ops) with its contractions on the fp16 matrix pipe as two-term operand splits --
# fp32-grade results (include/slide_engine.h: SLIDE_PREC_SPLIT)
PREC = {"fp32": 0, "fp16": 1, "split": 2}
(OP_GEMM, OP_PREP_POINTS, OP_ASSEMBLE_SA, OP_ASSEMBLE_FP, OP_FINALIZE_GN, OP_ATTN_COMBINE, OP_COPY_COLS, OP_TEMB,
 OP_COND, OP_UPDATE_POS, OP_UPDATE_FEAT, OP_ADVANCE_T) = range(1, 13)
OP_SYNC = 14
OP_ATTN_TAIL = 16
OP_GEMM_GX, OP_PAIR_NORM, OP_SA_CHAIN, OP_BLOCK_BODY, OP_PAIR_FIRST, OP_GEMM_CHAIN, OP_HEAD_UPDATE, OP_GEMM_GX_DUAL, OP_SA_CHAIN_P = 17, 18, 19, 30, 31, 32, 33, 34, 35
OP_PP_STAGE = 36
OP_POINT_CHAIN = 37


class SlideEpi(ctypes.Structure):
    _fields_ = [("mode", ctypes.c_int32), ("flags", ctypes.c_int32), ("gs", ctypes.c_int32), ("n_norm", ctypes.c_int32),
                ("inv_count", ctypes.c_float), ("stats_scale", ctypes.c_float),
                ("out_ld", ctypes.c_int32), ("res_ld", ctypes.c_int32),
                ("addvec_bs", ctypes.c_int32), ("stats_bs", ctypes.c_int32), ("pre_add_ld", ctypes.c_int32),
                ("pre_add_shift", ctypes.c_int32), ("addvec_idx_stride", ctypes.c_int32), ("pad0", ctypes.c_int32),
                ("bias", ctypes.c_void_p), ("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p),
                ("addvec", ctypes.c_void_p), ("addvec_idx", ctypes.c_void_p), ("residual", ctypes.c_void_p),
                ("pre_add", ctypes.c_void_p), ("out", ctypes.c_void_p),
                ("stats_sum", ctypes.c_void_p), ("stats_sq", ctypes.c_void_p),
                ("res_b", ctypes.c_void_p), ("res_vd", ctypes.c_void_p), ("res_vw", ctypes.c_void_p)]


class SlideGnFin(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("sum", "sq", "gid", "gstart", "gend", "gamma", "beta", "scale", "shift")] + [
        ("inv_count", ctypes.c_float), ("C", ctypes.c_int32), ("bs", ctypes.c_int32), ("G", ctypes.c_int32)]


class BodySlot(ctypes.Structure):  # csrc/block_body.hip
    _fields_ = [("src", ctypes.c_void_p), ("chunk_stride", ctypes.c_int32), ("nrows", ctypes.c_int32),
                ("kind", ctypes.c_int32), ("nvalid", ctypes.c_int32)]


class BodyArgs(ctypes.Structure):  # csrc/block_body.hip (same field order: natural alignment on both sides)
    _fields_ = [("slots", ctypes.c_void_p), ("n_slots", ctypes.c_int32),
                ("ta", ctypes.c_void_p), ("tb", ctypes.c_void_p),
                ("t_ld", ctypes.c_int32), ("off1", ctypes.c_int32), ("k1", ctypes.c_int32), ("offr", ctypes.c_int32),
                ("offk", ctypes.c_int32), ("kk", ctypes.c_int32),
                ("vv", ctypes.c_void_p), ("vbs", ctypes.c_int32), ("rv", ctypes.c_void_p),
                ("nbr", ctypes.c_void_p), ("d2", ctypes.c_void_p), ("w", ctypes.c_void_p),
                ("add0", ctypes.c_void_p), ("add0_idx", ctypes.c_void_p), ("add0_stride", ctypes.c_int32), ("add0_bs", ctypes.c_int32),
                ("sc", ctypes.c_void_p), ("sh", ctypes.c_void_p), ("aff_bs", ctypes.c_int32),
                ("P", ctypes.c_void_p), ("p_ld", ctypes.c_int32),
                ("vec1", ctypes.c_void_p), ("n1", ctypes.c_int32), ("gs1", ctypes.c_int32), ("inv1", ctypes.c_float),
                ("add1", ctypes.c_void_p), ("add1_bs", ctypes.c_int32),
                ("vecm", ctypes.c_void_p), ("n_mo", ctypes.c_int32), ("gsm", ctypes.c_int32), ("invm", ctypes.c_float),
                ("addm", ctypes.c_void_p), ("addm_bs", ctypes.c_int32),
                ("vecu", ctypes.c_void_p), ("n_u", ctypes.c_int32), ("gsu", ctypes.c_int32), ("nnu", ctypes.c_int32), ("invu", ctypes.c_float),
                ("vect", ctypes.c_void_p), ("n_out", ctypes.c_int32), ("gsv", ctypes.c_int32), ("nnv", ctypes.c_int32), ("invv", ctypes.c_float),
                ("out", ctypes.c_void_p), ("out_ld", ctypes.c_int32), ("out2", ctypes.c_void_p), ("out2_ld", ctypes.c_int32), ("out2_n", ctypes.c_int32),
                ("B", ctypes.c_int32), ("dbg", ctypes.c_void_p)]


class SlideChainLayer(ctypes.Structure):  # include/slide_engine.h
    _fields_ = [("X", ctypes.c_void_p), ("W", ctypes.c_void_p), ("epi", ctypes.c_void_p),
                ("x_ld", ctypes.c_int32), ("k_pad", ctypes.c_int32), ("n_cob", ctypes.c_int32), ("pad", ctypes.c_int32)]


class SlideHeadArgs(ctypes.Structure):  # include/slide_engine.h
    _fields_ = [("X", ctypes.c_void_p), ("W0", ctypes.c_void_p), ("W1", ctypes.c_void_p), ("v0", ctypes.c_void_p),
                ("b1", ctypes.c_void_p), ("eps_out", ctypes.c_void_p),
                ("rows", ctypes.c_int32), ("x_ld", ctypes.c_int32), ("k0", ctypes.c_int32), ("n1c", ctypes.c_int32),
                ("eps_ld", ctypes.c_int32),
                ("kind", ctypes.c_int32), ("C", ctypes.c_int32), ("kdim", ctypes.c_int32), ("ldf", ctypes.c_int32),
                ("half_out", ctypes.c_int32), ("n_copies", ctypes.c_int32),
                ("clamp", ctypes.c_float), ("seed_lo", ctypes.c_uint32), ("seed_hi", ctypes.c_uint32),
                ("x", ctypes.c_void_p), ("noise", ctypes.c_void_p), ("t_dev", ctypes.c_void_p),
                ("keypoint", ctypes.c_void_p), ("t0", ctypes.c_void_p), ("t1", ctypes.c_void_p), ("t2", ctypes.c_void_p),
                ("t3", ctypes.c_void_p), ("t4", ctypes.c_void_p), ("complete_x0", ctypes.c_void_p), ("kmask", ctypes.c_void_p),
                ("feat0", ctypes.c_void_p), ("copies", ctypes.c_void_p)]


class SlidePointChainArgs(ctypes.Structure):  # include/slide_engine.h
    _fields_ = [("Z", ctypes.c_void_p), ("Wz", ctypes.c_void_p), ("W2", ctypes.c_void_p), ("W0", ctypes.c_void_p), ("W1", ctypes.c_void_p),
                ("vz", ctypes.c_void_p), ("v2", ctypes.c_void_p), ("v0", ctypes.c_void_p), ("b1", ctypes.c_void_p),
                ("tvec", ctypes.c_void_p), ("t_idx", ctypes.c_void_p), ("cvec", ctypes.c_void_p), ("X", ctypes.c_void_p),
                ("eps", ctypes.c_void_p),
                ("Wz_lo", ctypes.c_void_p), ("W2_lo", ctypes.c_void_p), ("W0_lo", ctypes.c_void_p), ("W1_lo", ctypes.c_void_p),
                ("rows", ctypes.c_int32), ("z_ld", ctypes.c_int32), ("kz", ctypes.c_int32), ("x_ld", ctypes.c_int32),
                ("k0", ctypes.c_int32), ("n1c", ctypes.c_int32), ("eps_ld", ctypes.c_int32), ("t_stride", ctypes.c_int32),
                ("t_bs", ctypes.c_int32), ("c_bs", ctypes.c_int32), ("fuse_update", ctypes.c_int32), ("upd", SlideHeadArgs)]


class SlideOp(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("i", ctypes.c_int32 * 11), ("f", ctypes.c_float * 4),
                ("p", ctypes.c_void_p * 14)]


def ru(x, m=32):
    return (x + m - 1) // m * m


def make_op(kind, i=(), f=(), p=()):
    o = SlideOp()
    o.kind = kind
    for k, v in enumerate(i):
        o.i[k] = int(v)
    for k, v in enumerate(f):
        o.f[k] = float(v)
    for k, v in enumerate(p):
        o.p[k] = None if v is None else int(v)
    return o


def gn_layout(C):
    """MyGroupNorm(min(32, C), C) (pointnet2_modules.py:24-42): the first n_norm = C - C % G channels are
    normalised in G groups, the rest pass through.  Returns the PHYSICAL channel layout the engine uses so that
    every group is a power-of-two run inside one 32-channel block:
    (phys_index[c] for logical c, C_phys, n_norm_phys, gs_phys, gs_logical)."""
    G = min(32, C)
    n_norm = C - C % G
    gs = n_norm // G
    gs_p = 1
    while gs_p < gs:
        gs_p *= 2
    assert gs_p <= 32, "GroupNorm group size > 32 not supported by the epilogue"
    idx = np.empty(C, np.int64)
    for c in range(C):
        idx[c] = (c // gs) * gs_p + c % gs if c < n_norm else G * gs_p + (c - n_norm)
    return idx, int(G * gs_p + (C - n_norm)), int(G * gs_p), int(gs_p), int(gs)


class _Arena:
    """keeps device tensors alive and hands out raw pointers"""

    def __init__(self, device):
        self.device = device
        self.keep = []

    def zeros(self, *shape, dtype=torch.float32):
        t = torch.zeros(*shape, device=self.device, dtype=dtype)
        self.keep.append(t)
        return t

    def put(self, arr, dtype=None):
        t = torch.from_numpy(np.ascontiguousarray(arr))
        if dtype is not None:
            t = t.to(dtype)
        t = t.to(self.device)
        self.keep.append(t)
        return t


class SlidePrepCopy(ctypes.Structure):
    _fields_ = [("dst", ctypes.c_void_p), ("ld", ctypes.c_int32), ("kind", ctypes.c_int32), ("n", ctypes.c_int32),
                ("pad", ctypes.c_int32)]


class DenoiserEngine:
    NP = 16  # latent points per sample
    # chunk-major storage (see _buf): off unless __init__ enables it (bare plan builders in tests / tools stay row-major)
    use_cm = False
    use_gx = False
    use_gxs = False
    _cm = frozenset()
    _fm = frozenset()
    _cm_copy = {}

    def __init__(self, hp, state_dict, batch, device, prec="fp32", per_sample_t=True, t_table=0):
        """per_sample_t=True : `forward(x, ts, label)` API, the t-embedding MLP runs every call (one workgroup / sample).
        per_sample_t=False: sampler mode -- the whole batch shares the device-side timestep t_dev[0]; the t-embedding
        path is evaluated ONCE for all t in [0, t_table) into a table that the GEMM epilogues index with t_dev[0]."""
        lib()  # fail loudly if the HIP library is missing
        # a plan may be BUILT on the CPU device (structure / FLOP checks in the CPU tests); it can only RUN on a GPU
        self.hp, self.B, self.device = hp, int(batch), device
        self.prec = PREC[prec]
        self.per_sample_t = per_sample_t
        self.t_table = int(t_table)
        assert per_sample_t or t_table > 0
        self.adt = torch.float16 if self.prec == 1 else torch.float32  # activation storage type
        import os as _os
        self.use_glds = _os.environ.get("SLIDE_GLDS", "1") != "0"  # LDS-DMA GEMM variant for fp16 GEMMs
        self.glds_nst = int(_os.environ.get("SLIDE_GLDS_WIDE", "0"))  # 1: 64-deep K chunks (full cache lines)
        self.sd = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)).astype(np.float32)
                   for k, v in state_dict.items()}
        arch = hp["architecture"]
        assert not hp.get("include_local_feature", True) and not hp.get("include_global_feature", False)
        assert arch["neighbor_definition"] == "nn" and arch.get("use_knn_FP", False) and not arch.get("include_grouper", False)
        assert hp["attach_position_to_input_feature"] and hp["include_abs_coordinate"] and hp.get("include_center_coordinate", False)
        assert hp["bias"] and hp["res_connect"] and not hp["bn_first"] and hp.get("bn", True)
        assert all(n >= self.NP for n in arch["npoint"]), "engine is specialised to N == npoint == 16 (no FPS)"
        assert all(ns >= self.NP for ns in arch["nsample"]) and arch["K"] == 8
        self.cx = 3 + hp["in_fea_dim"]
        self.out_dim = hp["out_dim"]
        self.t_dim = hp["t_dim"]
        self.A = _Arena(device)
        # chunk-major activations / weights for the 128- and 256-row ring kernels (SLIDE_CM=0: row-major everywhere, A/B)
        self.use_cm = (self.prec == 1 and self.use_glds and self.glds_nst != 1 and _os.environ.get("SLIDE_CM", "1") != "0"
                       and not _os.environ.get("SLIDE_XS", ""))
        # pair decomposition of the blocks' first layers + generated-X GEMMs (csrc/gemm_gx.hip; SLIDE_GX=0: the round-2 plan)
        self.use_gx = self.use_cm and _os.environ.get("SLIDE_GX", "1") != "0"
        # round 5: the pair decomposition in the SPLIT arithmetic (csrc/gemm_gxs.hip): float pair tables, generated-X split GEMMs,
        # PAIR residuals on float rows, the split attention tail -- the K-expanded first-layer outputs, scores and values of a block
        # never reach memory (SLIDE_GXS=0: the fp32-structured plan of round 4)
        self.use_gxs = self.prec == 2 and _os.environ.get("SLIDE_GXS", "1") != "0"
        if self.use_gxs:
            # its kernels run the split products on ONE accumulator set with the weight's high term scaled by 2^11 in fp16 (DESIGN.md
            # section 5): exact while |w| < 32.  A checkpoint with a larger convolution weight takes the fp32-structured split plan.
            wmax = max([float(np.abs(v).max()) for k, v in self.sd.items() if k.endswith(".weight") and v.ndim >= 3 and v.size] + [0.0])
            if wmax >= 31.0:
                import warnings
                warnings.warn("a convolution weight of magnitude %.1f >= 31: the split plan falls back to its fp32-structured form "
                              "(SLIDE_GXS=0 behaviour)" % wmax)
                self.use_gxs = False
        self._cm = set()
        self._fm = set()  # chunk-major buffers whose 32-row groups are fragment-major (_tail_fm)
        self._cm_copy = {}  # per-point table (data_ptr) -> its chunk-major copy, written by the table's producer as well
        self.ops = []
        self._w16 = {}
        self._w16_next = None
        self._lane = 0
        self.two_lanes = _os.environ.get("SLIDE_TWO_LANES", "0") != "0"  # measured: no gain at batch 256
        self._tvec = []   # (name, width) of every Mlp .fc           -> offsets into the t vector
        self._cvec = []   # (name, width) of every Mlp .fc_condition -> offsets into the condition vector
        self._build()

    # ------------------------------------------------------------------ small helpers
    def _w(self, name):
        w = self.sd[name]
        return w.reshape(w.shape[0], -1)

    def _buf(self, rows, ch, dtype=None, cm=False, fm=False):
        """activation matrix [rows][ld].  cm=True (fp16 LDS-DMA plans): CHUNK-MAJOR storage [ld / 32][rows][32] -- the same
        bytes, but the 64-byte piece a ring kernel's DMA lane group fetches for row r + 1 follows the one of row r, so one
        LDS-DMA instruction reads 1 KB of consecutive memory (2.2x the L2 -> LDS rate, tools/lds_fill.hip) and one epilogue
        store instruction writes 2 KB of it.  Only for buffers whose every producer is a GEMM epilogue and every consumer a
        ring-kernel loader / epilogue (pointer + leading dimension per 32-channel block express the layout: ld == 32)."""
        t = self.A.zeros(rows, ru(ch), dtype=self.adt if dtype is None else dtype)
        if cm and self.use_cm:
            self._cm.add(t.data_ptr())
            if fm:
                assert t.dtype == torch.float16 and rows % 32 == 0
                self._fm.add(t.data_ptr())
        return t

    def _is_cm(self, t):
        return t.data_ptr() in self._cm

    def _is_fm(self, t):
        return t.data_ptr() in self._fm

    def _tail_fm(self, mpfx, apfx, npx_log2, n_mo):
        """round 6: are the K-expanded inputs u / mo of this attention block's fused tail stored FRAGMENT-major (SLIDE_F_OUT_FM,
        include/slide_engine.h: a chunk-major slab whose 32-row groups are ordered as the MFMA fragments the register-X tail
        kernel loads -- 1 KB of consecutive memory per wave load)?  Exactly when run_attn_tail (csrc/engine.hip) picks
        attn_tail_rx_kernel for it: fp16 pair-decomposition plan (its generated-X kernels and the SA chain are the producers that
        can store the layout), fused tail, values' chunk count a multiple of 4, none
        of the opt-in tail forms.  SLIDE_FM=0: chunk-major u / mo (A/B)."""
        env = os.environ.get
        cout = self.sd[apfx + ".weight_conv.5.weight"].shape[0]
        # mo's producer must be one that stores the layout: the generated-X GEMM of a two-layer Mlp, or the SA chain -- a rest_mlp
        # outside the chain is a ring GEMM over a stored h2 (engine.hip: its epilogue instantiations do not carry the layout)
        if (mpfx + ".rest_mlp.0.weight") in self.sd and not self._sa_chain_shapes(mpfx, npx_log2):
            return False
        return bool(self.use_cm and self.use_gx and self.prec == 1 and self.use_glds and env("SLIDE_FM", "1") != "0"
                    and env("SLIDE_ATTN_TAIL", "1") != "0" and env("SLIDE_TAIL_RX", "1") != "0" and env("SLIDE_TAIL8", "0") == "0"
                    and env("SLIDE_TAIL_OCC3", "0") == "0" and "SLIDE_TAIL_WIDE" not in os.environ and env("SLIDE_BODY", "0") == "0"
                    and npx_log2 in (7, 8) and (ru(n_mo) // 32) % 4 == 0 and np.array_equal(gn_layout(cout)[0], np.arange(cout)))

    def _ldp(self, t):
        """leading dimension as the kernels see it"""
        return 32 if self._is_cm(t) else t.shape[1]

    def _colptr(self, t, col):
        """address of channel `col` of row 0"""
        if self._is_cm(t):
            return t.data_ptr() + t.element_size() * ((col // 32) * t.shape[0] * 32 + col % 32)
        return t.data_ptr() + t.element_size() * col

    def _emit(self, op):
        op.i[10] = self._lane if self.two_lanes else 0
        self.ops.append(op)
        w16 = getattr(self, "_w16_next", None)
        if w16 is not None:
            self._w16[id(op)] = (op, w16)
            self._w16_next = None

    def _sync(self, frm, to):
        if self.two_lanes:
            self.ops.append(make_op(OP_SYNC, i=(frm, to)))

    def _sched(self):
        """tile counters of one persistent GEMM launch (9 ints, zero; the kernel re-arms them itself)"""
        return self.A.put(np.zeros(16, np.int32))

    def _tvec_off(self, prefix, width):
        off = sum(w for _, w in self._tvec)
        self._tvec.append((prefix, width))
        return off

    def _cvec_off(self, prefix, width):
        off = sum(w for _, w in self._cvec)
        self._cvec.append((prefix, width))
        return off

    def _gemm(self, X, npx_log2, segs, in_cols=None, in_affine=None, k_logical=None, gather=None, gn_fin=None,
              pre_gather=None, gx=None, pair_tabs=None, pair_fused=None, defer=None, chain=None):
        """X: input buffer [rows][ld].  segs: list of dicts describing consecutive output segments:
             w (O,I) bias (O) | out (tensor) out_coff | mode flags | gn=(gamma,beta) for NORM | layout (gn_layout) |
             addvec=(tensor, off, bs) | residual tensor | bcast | stats=(sum,sq tensors, coff, scale)
           in_cols: physical column index of every logical input channel (None = identity)."""
        if defer is not None:  # a layer CHAINED onto a split generated-X GEMM (csrc/gemm_gxs.hip): packed, not launched
            rows, ld, x_ld = defer["rows"], defer["k_pad"], defer["k_pad"]
            assert X is None and self.use_gxs and gather is None and gn_fin is None and gx is None and in_affine is None
        elif gx is not None:  # generated-X GEMM of the pair decomposition (SLIDE_OP_GEMM_GX): X is never stored
            rows, ld, x_ld = gx["rows"], gx["k_pad"], 32
            assert X is None and gather is None and gn_fin is None and npx_log2 in (7, 8) and (self.use_cm or self.use_gxs)
        else:
            rows, ld = X.shape
            x_ld = self._ldp(X)
            assert not self._is_cm(X) or (npx_log2 >= 7 and gather is None)
            assert not self._is_fm(X), "fragment-major buffers are read by the register-X attention tail only"
        if gather is not None:  # (feature table, neighbour table, K, chunks read from the table): X holds the remaining columns
            ld = gather[3] * 32 + ld
        npx = 1 << npx_log2
        wrows, tables = [], []
        vec_list = []
        for sg in segs:
            w = sg["w"]
            O, I = w.shape
            lay = sg.get("layout")
            if lay is None:
                oidx, Op, n_norm_p, gs_p, gs_l = np.arange(O), O, 0, 1, 1
            else:
                oidx, Op, n_norm_p, gs_p, gs_l = lay
            Opad = ru(Op)
            wp = np.zeros((Opad, ld), np.float32)
            cols = np.arange(I) if in_cols is None else in_cols
            wp[np.ix_(oidx, cols)] = w
            wrows.append(wp)
            vec = np.zeros((3, Opad), np.float32)
            if sg.get("bias") is not None:
                vec[0, oidx] = sg["bias"]
            if sg.get("gn") is not None:
                gam, bet = sg["gn"]
                nn = gam.shape[0]
                vec[1, oidx[:nn]] = gam
                vec[2, oidx[:nn]] = bet
            vec_list.append((vec, sg, Opad, n_norm_p, gs_p, gs_l))
        W = np.concatenate(wrows, axis=0)
        n_cob = W.shape[0] // 32
        # chunk-major weights [k / 32][n][32] for the ring kernels of the 128- / 256-row samples (the 16-row launches run the
        # split-K small-launch kernel, which reads row-major weights)
        w_cm = bool(self.use_cm and npx_log2 >= 7) or (gx is not None and self.prec == 1)
        Wst = np.ascontiguousarray(W.reshape(W.shape[0], ld // 32, 32).transpose(1, 0, 2)) if w_cm else W
        Wd = self.A.put(Wst, torch.float16 if self.prec == 1 else torch.float32)
        epis = (SlideEpi * n_cob)()
        blk = 0
        for vec, sg, Opad, n_norm_p, gs_p, gs_l in vec_list:
            vd = self.A.put(vec)
            out = sg["out"]
            coff = sg.get("out_coff", 0)
            flags = sg.get("flags", 0)
            if out is None:  # (the layer's output stays in the chaining kernel's registers)
                assert chain is not None and len(segs) == 1
            else:
                assert out.shape[1] >= coff + Opad and coff % (32 if self._is_cm(out) else 8) == 0, (out.shape, coff, Opad)
                assert out.shape[0] == rows, (out.shape, rows)
                if out.dtype == torch.float32 and self.prec == 1:
                    flags |= F_OUT_F32
                else:
                    assert out.dtype == self.adt
                if self._is_fm(out):
                    assert self.prec == 1 and npx_log2 >= 7 and out.dtype == torch.float16
                    flags |= F_OUT_FM
            for j in range(Opad // 32):
                e = epis[blk]
                e.mode = sg.get("mode", EPI_RAW)
                e.flags = flags
                e.gs = gs_p
                e.n_norm = int(min(32, max(0, n_norm_p - 32 * j)))
                e.inv_count = 1.0 / (gs_l * sg.get("npx", npx))  # (npx: a pair segment's GroupNorm runs over the K-expanded rows)
                e.out_ld = 0 if out is None else self._ldp(out)
                e.bias = vd.data_ptr() + 4 * (32 * j)
                e.gamma = vd.data_ptr() + 4 * (Opad + 32 * j)
                e.beta = vd.data_ptr() + 4 * (2 * Opad + 32 * j)
                e.out = None if out is None else self._colptr(out, coff + 32 * j)
                if sg.get("addvec") is not None:
                    t, off, bs, idx, idx_stride = sg["addvec"]
                    assert off % 4 == 0 and bs % 4 == 0 and idx_stride % 4 == 0
                    e.addvec = t.data_ptr() + 4 * (off + 32 * j)
                    e.addvec_bs = bs
                    if idx is not None:
                        e.addvec_idx = idx.data_ptr()
                        e.addvec_idx_stride = idx_stride
                if sg.get("res_pair") is not None:
                    # PAIR residual: residual(p, j) = ta[q] + tb[p] (+ d2 vd + w vw): (ta, tb fp16 tables, column offset,
                    # vd | vw fp32 [2][Opad] for the 8-neighbour samples or None)
                    rta, rtb, rcoff, rvv = sg["res_pair"]
                    assert rta.dtype == (torch.float16 if self.prec == 1 else torch.float32) and rta.shape == rtb.shape
                    assert rta.shape[0] == self.B * 16 and (rcoff + 32 * j) % 8 == 0 and rta.shape[1] >= rcoff + Opad
                    e.residual = rta.data_ptr() + rta.element_size() * (rcoff + 32 * j)
                    e.res_b = rtb.data_ptr() + rta.element_size() * (rcoff + 32 * j)
                    e.res_ld = rta.shape[1]
                    if rvv is None:
                        assert npx_log2 == 8
                        e.flags |= F_RES_PAIR
                    else:
                        assert npx_log2 == 7 and pair_tabs is not None
                        e.flags |= F_RES_PAIR_NBR
                        e.res_vd = rvv.data_ptr() + 4 * (32 * j)
                        e.res_vw = rvv.data_ptr() + 4 * (Opad + 32 * j)
                if sg.get("residual") is not None:
                    r = sg["residual"]
                    assert r.shape[0] == rows and r.shape[1] >= Opad and r.dtype == self.adt
                    e.residual = self._colptr(r, 32 * j)
                    e.res_ld = self._ldp(r)
                if sg.get("pre_add") is not None:
                    pa, shift = sg["pre_add"][:2]
                    pcoff = sg["pre_add"][2] if len(sg["pre_add"]) > 2 else 0
                    # shift < 0: row of the NEIGHBOUR point (K = 2^-shift), through the table passed as pre_gather
                    assert pa.shape[0] == (rows >> shift if shift >= 0 else rows >> npx_log2 << 4), (pa.shape, rows, shift)
                    assert (shift >= 0 or pre_gather is not None) and pa.shape[1] >= pcoff + Opad and pa.dtype == self.adt
                    assert not self._is_cm(pa)
                    e.pre_add = pa.data_ptr() + pa.element_size() * (pcoff + 32 * j)
                    e.pre_add_ld = pa.shape[1]
                    e.pre_add_shift = shift
                if sg.get("stats") is not None:
                    ssum, ssq, scoff, scale = sg["stats"]
                    e.stats_sum = ssum.data_ptr() + 4 * (scoff + 32 * j)
                    e.stats_sq = ssq.data_ptr() + 4 * (scoff + 32 * j)
                    e.stats_bs = ssum.shape[1]
                    e.stats_scale = scale
                blk += 1
        # PACKED VECTORS (SLIDE_EPI_PACKED_VECS, include/slide_engine.h): the blocks' [bias | gamma | beta] values behind the descriptor
        # array, bit 0 of the pointer the ops carry says so -- the kernels stage both by LDS-DMA instead of chasing the three pointers
        # of every block in every workgroup's prologue (SLIDE_PACKED_VECS=0: plain pointer, A/B)
        pv = np.zeros((n_cob, 3, 32), np.float32)
        blk = 0
        for vec, sg, Opad, _n, _g, _l in vec_list:
            v3 = np.asarray(vec, np.float32).reshape(3, Opad)
            for j in range(Opad // 32):
                pv[blk] = v3[:, 32 * j:32 * j + 32]
                blk += 1
        assert blk == n_cob
        ed = self.A.put(np.concatenate([np.frombuffer(bytes(epis), dtype=np.uint8), pv.reshape(-1).view(np.uint8)]))
        edp = ed.data_ptr() | (1 if os.environ.get("SLIDE_PACKED_VECS", "1") != "0" else 0)
        sc = sh = None
        in_bs = aff_off = 0
        if in_affine is not None:
            sc, sh, aff_off, in_bs = in_affine
        if defer is not None:
            return dict(W=Wd, epi=ed, epi_ptr=edp, n_cob=n_cob, k_pad=ld, flops=2 * rows * sum(int(s_["w"].size) for s_ in segs),
                        wr=sum(rows * v[2] * v[1]["out"].element_size() for v in vec_list), wbytes=W.size * 4)
        if gx is not None:
            return self._emit_gx(gx, npx_log2, rows, ld, n_cob, Wd, edp, segs, W, vec_list, in_affine, pair_tabs, chain=chain, keep=ed)
        if pair_fused is not None:  # SLIDE_OP_PAIR_FIRST: per-point GEMM + pair-table pass in one launch (_pair_first)
            pf = pair_fused
            assert npx_log2 == 4 and self.prec == 1 and in_affine is None and gather is None and gn_fin is None and not w_cm
            assert X.dtype == self.adt and not self._is_cm(X)
            self.gemm_flops[len(self.ops)] = 2 * rows * sum(int(s_["w"].size) for s_ in segs)
            self.gemm_bytes[len(self.ops)] = (rows * ld * 2 + W.size * 2, 2 * rows * pf["ld"] * 2 + rows * 32 * pf["cob0"] * 2)
            self.flops += 2 * rows * sum(int(s_["w"].size) for s_ in segs)
            self.kernel_names[len(self.ops)] = "pair_first_kernel<%s>" % ("true" if pf["K"] == 8 else "false")
            ptr = lambda t: None if t is None else t.data_ptr()
            self._emit(make_op(OP_PAIR_FIRST, i=(rows, x_ld, ld, n_cob, pf["cob0"], pf["ld"], pf["K"]),
                               p=(X.data_ptr(), Wd.data_ptr(), edp, self.xyz.data_ptr(), pf["wa"].data_ptr(),
                                  pf["wb"].data_ptr(), pf["ta"].data_ptr(), pf["tb"].data_ptr(), ptr(pf.get("nbr")), ptr(pf.get("d2")),
                                  ptr(pf.get("w")), ptr(pf.get("vv_in")), ptr(pf.get("vv")))))
            return
        assert X.dtype == self.adt
        # wide (128-channel) tiles only when the grid still covers the 256 CUs at least twice
        ntr = (rows + 255) // 256
        cbw = 4 if (self.prec == 1 and n_cob >= 4 and ntr * ((n_cob + 3) // 4) >= int(os.environ.get('SLIDE_CBW4_TILES', '1000'))) else 2
        # narrow launches: 32-channel tiles double the workgroups and halve each wave's epilogue while they still fit one round
        if (self.prec == 1 and self.use_glds and sc is None and npx_log2 >= 7 and ntr * n_cob <= int(os.environ.get('SLIDE_CBW1_TILES', '0'))):
            cbw = 1
        self.gemm_flops[len(self.ops)] = 2 * rows * sum(int(s["w"].size) for s in segs)
        esz = X.element_size()
        rd = rows * ld * esz + W.size * esz + sum(rows * v[2] * esz for v in vec_list if v[1].get("residual") is not None)
        rd += sum((rows >> max(v[1]["pre_add"][1], 0) if v[1]["pre_add"][1] >= 0 else rows >> npx_log2 << 4) * v[2] * esz
                  for v in vec_list if v[1].get("pre_add") is not None)  # per-point tables (unique rows)
        wr = sum(rows * v[2] * v[1]["out"].element_size() for v in vec_list)
        self.gemm_bytes[len(self.ops)] = (rd, wr)  # algorithmic HBM bytes (read, written) of this launch
        glds = int(self.use_glds and self.prec == 1 and (sc is None or npx_log2 >= 7))
        # X-stationary kernel (csrc/gemm_xs.hip: a workgroup keeps its input resident in LDS -- for the GATHERED first layers
        # of the SA / FP blocks only the 16-row point table + the coordinate chunk, 24 KB instead of a 144 KB X tile -- and
        # computes several column tiles of a row tile from it, weights through a small LDS-DMA ring; bit-identical to the ring
        # kernels, tests/test_hip_engine.py).  OPT-IN.  Measured at batch 256 (round 2): alone on the GPU
        # the gathered layers gain (SA1 101 -> 85 us, FP1 87 -> 72 us; one feature chain 0.94 -> 0.91 ms/step), but in bench.py's
        # arrangement (three feature sub-batches + the position chain in flight) every variant LOSES to the ring kernels:
        # 252 / 270 / 253 shapes/s ("auto" at <= 3 / 2 / 1 workgroups per CU) vs 286 -- the ring tiles' three small
        # workgroups per CU interleave with the other chains' kernels, these larger-footprint workgroups do not.
        # SLIDE_XS = "" (default: off) | "auto" (gathered layers + plain layers with >= 8 column tiles per workgroup) |
        # "7,8" (every eligible layer of those sample sizes).
        wfrag = None
        xs_mode = os.environ.get("SLIDE_XS", "")
        xs_levels = {7, 8} if xs_mode == "auto" else {int(v) for v in xs_mode.split(",") if v}
        xs_lds = ((ld - (gather[3] * 32 if gather is not None else 0)) // 32) * 16384  # resident X rows (gathered chunks: 1 KB each)
        if (glds and npx_log2 in xs_levels and gn_fin is None and not (gather is not None and sc is not None)
                and xs_lds + 20 * 1024 <= 160 * 1024):
            if xs_mode != "auto" or gather is not None or (npx_log2 == 8 and sc is None and n_cob >= 16 and xs_lds > 96 * 1024):
                wfrag = Wd  # (selects the kernel; it reads the same row-major weights)
                cbw = 4 if n_cob >= 4 and os.environ.get("SLIDE_XS_CBW", "2") == "4" else 2
        gtab = None if gather is None else self._cm_copy.get(gather[0].data_ptr(), gather[0])  # chunk-major copy: g_ldf == 32
        assert wfrag is None or gtab is None or gtab is gather[0]
        gf = (0.0, 0.0, 0.0) if gather is None else (float(gather[3]), float(32 if gtab is not gather[0] else gather[0].shape[1]),
                                                     float({8: 3, 16: 4}[gather[2]]))
        knob = self.glds_nst
        assert wfrag is None or not (w_cm or self._is_cm(X))
        if wfrag is not None and os.environ.get("SLIDE_XS_OCC"):
            knob = 10 + int(os.environ["SLIDE_XS_OCC"])  # cap the workgroups per CU of the X-stationary kernel (A/B timing)
        has_pair = any(v[1].get("res_pair") is not None for v in vec_list)  # -> the kernels compiled with the PAIR residual
        if self.use_gxs and npx_log2 == 4 and gather is None and gn_fin is None and pre_gather is None and wfrag is None and \
                not any(v[1].get("pre_add") is not None for v in vec_list):
            self._w16_next = W  # (a per-point layer of a split plan: _merge_pp may fold it into a SLIDE_OP_PP_STAGE launch)
        self._emit(make_op(OP_GEMM, i=(rows, x_ld, ld, n_cob, npx_log2, in_bs, self.prec, cbw, glds | (2 if w_cm else 0) | (4 if has_pair else 0), knob),
                           f=(-1.0 if self.persistent == 2 else float(os.environ.get('SLIDE_STAGGER_US', '0')),) + gf,
                                p=(X.data_ptr(), Wd.data_ptr(), edp,
                                   None if sc is None else sc.data_ptr() + 4 * aff_off,
                                   None if sh is None else sh.data_ptr() + 4 * aff_off, None,
                                   None if gn_fin is None else gn_fin.data_ptr(),
                                   self._sched().data_ptr() if self.persistent else None,
                                   None if gather is None else gtab.data_ptr(),
                                   (None if pre_gather is None else pre_gather.data_ptr()) if gather is None
                                   else gather[1].data_ptr(),
                                   None if wfrag is None else wfrag.data_ptr(), None,
                                   None if pair_tabs is None else pair_tabs[1].data_ptr(),
                                   None if pair_tabs is None else pair_tabs[2].data_ptr())))
        if pair_tabs is not None:  # PAIR_NBR residual: p[9] = neighbour table, p[12] / p[13] = squared distances / weights
            assert gather is None and pre_gather is None
            self.ops[-1].p[9] = pair_tabs[0].data_ptr()
        self.flops += 2 * rows * sum(int(s["w"].size) for s in segs)

    @staticmethod
    def _split1_ok(*ws):
        """the single-accumulator split kernels (csrc/gemm_gxs.hip) scale the weight's high term by 2^11 in fp16: |w| < 32"""
        return all(float(np.abs(w).max()) < 31.0 for w in ws if w.size)

    def _emit_gx(self, gx, npx_log2, rows, ld, n_cob, Wd, edp, segs, W, vec_list, in_affine, pair_tabs, chain=None, keep=None):
        """SLIDE_OP_GEMM_GX (include/slide_engine.h): gx = dict(ta, tb (fp16 tables [B*16][t_ld]), coff (first table column),
        k_pad, rows, mode, add=(tensor, offset, per-sample stride, idx tensor or None, idx stride) or None, vv = per-sample
        (vd | vw) fp32 [B][2][t_ld] of the 8-neighbour samples or None); pair_tabs = (neighbour, d2, w tables) for those"""
        ta, tb, coff = gx["ta"], gx["tb"], gx["coff"]
        tes = ta.element_size()
        assert ta.dtype == (torch.float16 if self.prec == 1 else torch.float32) and ta.shape == tb.shape and coff % 8 == 0
        assert ta.shape[1] >= coff + ld
        fl = 2 * rows * sum(int(s_["w"].size) for s_ in segs)
        self.gemm_flops[len(self.ops)] = fl
        rd = 2 * (rows >> npx_log2) * 16 * ld * tes + W.size * tes
        wr = sum(rows * v[2] * v[1]["out"].element_size() for v in vec_list if v[1]["out"] is not None)
        if chain is not None:  # the chained layer's work rides on this launch
            fl += chain["flops"]; rd += chain["wbytes"]; wr += chain["wr"]
            self.gemm_flops[len(self.ops)] = fl
        self.gemm_bytes[len(self.ops)] = (rd, wr)
        sc = sh = None
        in_bs = aff_off = 0
        if in_affine is not None:
            sc, sh, aff_off, in_bs = in_affine
        add = gx.get("add")
        vv = gx.get("vv")
        assert (npx_log2 == 7) == (pair_tabs is not None)
        # (the launcher's choice, csrc/gemm_gx.hip: mode-1 layers run 256 x 64 tiles when three workgroups fit a CU's LDS)
        nsamp, nvec = 256 >> npx_log2, (2 if gx["mode"] else 1) + (2 if npx_log2 == 7 else 0)
        # (LDS of the 64-channel form: ring stages of 8 KB of weights + 4 KB of streamed table images per sample, epilogue
        #  descriptors, per-sample vectors; three stages, else two)
        shm64 = lambda nst: nst * (8192 + 4096 * nsamp) + (2 * 40 + 2 * 96) * 4 + nsamp * nvec * ld * 2 + 16
        knob = os.environ.get("SLIDE_GX_N64", "1")
        n64, nst = 0, 3
        if knob != "0" and gx["mode"] == 1 and shm64(3) <= 53 * 1024:
            n64 = 1
        elif knob != "0" and os.environ.get("SLIDE_GX_N64W", "1") != "0" and (
                n_cob <= 2 or ((rows + 255) // 256) * ((n_cob + 3) // 4) <= 256):
            # (n_cob <= 2 -- the position net's 32- / 64-channel layers: a 128-channel tile would be half empty)
            # where the 128-channel grid would leave CUs empty -- the FP blocks at 88 samples: 96 workgroups -- 64-channel tiles at
            # two workgroups per CU, either mode (SLIDE_GX_N64W=0: 128-channel tiles).  Neutral while the staged tables pinned
            # these launches to one workgroup per CU (385.4 vs 386.6); with the tables streamed through the ring 394.3 vs 390.2
            n64 = 2
        if n64 == 0:  # 128-channel tiles: two workgroups per CU (80 KB each)
            shm128 = lambda k_: k_ * (16384 + 4096 * nsamp) + (4 * 40 + 4 * 96) * 4 + nsamp * nvec * ld * 2 + 16
            nst = 3 if shm128(3) <= 80 * 1024 else 2
        self.kernel_names[len(self.ops)] = "gemm_gx_%skernel<%d, %d, %d>" % (("", "n64_", "n64w_")[n64], npx_log2, nst, gx["mode"])
        if self.prec == 2:  # split arithmetic on float tables: csrc/gemm_gxs.hip (256 x 64 tiles)
            if not self._split1_ok(W):
                raise SlideHipError("a weight of magnitude >= 31 in a generated-X layer: the split pair-decomposition kernels scale the "
                                    "weights' high terms by 2^11 in fp16 -- build this plan with SLIDE_GXS=0")
            n64 = 3
            self.kernel_names[len(self.ops)] = ("gemm_gxs_chain_kernel<%d>" % npx_log2) if chain is not None else \
                "gemm_gxs_kernel<%d, %d>" % (npx_log2, gx["mode"])
            assert chain is None or (npx_log2 == 8 and gx["mode"] == 0 and n_cob <= 2 and chain["k_pad"] == n_cob * 32)
        self._emit(make_op(OP_GEMM_GX,
                           i=(rows, ta.shape[1], ld, n_cob, npx_log2, in_bs, gx["mode"], 0 if add is None else add[2],
                              0 if add is None else add[4], 0 if vv is None else 2 * vv.shape[2]),
                           f=(float(n64),) + ((float(chain["n_cob"]), float(chain["k_pad"])) if chain is not None else ()),
                           p=(ta.data_ptr() + tes * coff, Wd.data_ptr(), edp,
                              None if sc is None else sc.data_ptr() + 4 * aff_off,
                              None if sh is None else sh.data_ptr() + 4 * aff_off,
                              tb.data_ptr() + tes * coff,
                              None if add is None else add[0].data_ptr() + 4 * add[1],
                              None if add is None or add[3] is None else add[3].data_ptr(),
                              None if pair_tabs is None else pair_tabs[0].data_ptr(),
                              None if pair_tabs is None else pair_tabs[1].data_ptr(),
                              None if pair_tabs is None else pair_tabs[2].data_ptr(),
                              None if vv is None else vv.data_ptr() + 4 * coff,
                              None if chain is None else chain["W"].data_ptr(), None if chain is None else chain["epi_ptr"])))
        self.flops += fl

    # ------------------------------------------------------------------ blocks
    def _mlp_segments(self, pfx, tvec, cvec, out1, res_out):
        """first_mlp + res_connect segments of Mlp_plus_t_emb (pointnet2_modules.py:119-176) sharing one input"""
        sd = self.sd
        c1 = sd[pfx + ".first_mlp.0.weight"].shape[0]
        first = dict(w=self._w(pfx + ".first_mlp.0.weight"), bias=sd[pfx + ".first_mlp.0.bias"], mode=EPI_NORM,
                     flags=F_POST_RELU, layout=gn_layout(c1), out=out1,
                     gn=(sd[pfx + ".first_mlp.1.group_norm.weight"], sd[pfx + ".first_mlp.1.group_norm.bias"]))
        if (pfx + ".fc.weight") in sd:
            first["addvec"] = (tvec, self._tvec_off(pfx + ".fc", c1), self._t_bs,
                               None if self.per_sample_t else self.t_dev, self._n_fc)
        assert (pfx + ".res_connect.weight") in sd, "identity res_connect (mlp_spec[0]==mlp_spec[-1]) not planned"
        res = dict(w=self._w(pfx + ".res_connect.weight"), bias=sd[pfx + ".res_connect.bias"], mode=EPI_RAW, out=res_out)
        return first, res

    def _mlp_tail(self, pfx, npx_log2, h1, cvec, r, final_out, final_coff=0, pair=None):
        """second_mlp (+fc_condition) [+ rest_mlp] + residual, writing the module output.
        pair (the block's pair decomposition, _pair_first): h1 and r are not buffers -- the first GEMM generates its input
        from the pair tables (SLIDE_OP_GEMM_GX mode 0) and the residual enters as a PAIR residual"""
        sd = self.sd
        rows = h1.shape[0] if pair is None else pair["rows"]

        def first_gemm(seg_):
            if pair is None:
                return self._gemm(h1, npx_log2, [seg_])
            lay1 = pair["lay1"]
            self._gemm(None, npx_log2, [seg_], in_cols=lay1[0],
                       gx=dict(ta=pair["ta"], tb=pair["tb"], coff=pair["off1"], k_pad=ru(lay1[1]), rows=rows, mode=0,
                               add=pair["add1"], vv=pair["vv"]), pair_tabs=pair["tabs"])

        def with_res(seg_):
            if pair is None:
                seg_["residual"] = r
            else:
                seg_["res_pair"] = (pair["ta"], pair["tb"], pair["offr"], pair["rvv"])
            return seg_
        has_rest = (pfx + ".rest_mlp.0.weight") in sd
        c2 = sd[pfx + ".second_mlp.0.weight"].shape[0]
        seg = dict(w=self._w(pfx + ".second_mlp.0.weight"), bias=sd[pfx + ".second_mlp.0.bias"], mode=EPI_NORM,
                   flags=F_POST_RELU, layout=gn_layout(c2),
                   gn=(sd[pfx + ".second_mlp.1.group_norm.weight"], sd[pfx + ".second_mlp.1.group_norm.bias"]))
        if (pfx + ".fc_condition.weight") in sd:
            seg["addvec"] = (cvec, self._cvec_off(pfx + ".fc_condition", c2), self._c_bs, None, 0)
        if has_rest and pair is not None and self._sa_chain(pfx, npx_log2, pair, cvec, seg, final_out, final_coff):
            return
        if (has_rest and pair is not None and self.use_gxs and npx_log2 == 8 and ru(gn_layout(c2)[1]) <= 64
                and os.environ.get("SLIDE_GXS_CHAIN", "1") != "0"):
            # split plans (round 5): second_mlp -> rest_mlp of an SA block in ONE launch -- h2 stays in the generated-X kernel's
            # accumulators and feeds rest_mlp's contraction from there (csrc/gemm_gxs.hip, CHAIN); needs every channel of h2 in
            # one 64-channel tile.  SLIDE_GXS_CHAIN=0: two launches and an h2 round trip
            c3 = sd[pfx + ".rest_mlp.0.weight"].shape[0]
            assert np.array_equal(gn_layout(c2)[0], np.arange(c2)), "second_mlp width with padded GroupNorm groups"
            seg3 = with_res(dict(w=self._w(pfx + ".rest_mlp.0.weight"), bias=sd[pfx + ".rest_mlp.0.bias"], mode=EPI_NORM,
                                 flags=F_POST_RELU, layout=gn_layout(c3), out=final_out, out_coff=final_coff,
                                 gn=(sd[pfx + ".rest_mlp.1.group_norm.weight"], sd[pfx + ".rest_mlp.1.group_norm.bias"])))
            if self._split1_ok(seg3["w"]):
                layer2 = self._gemm(None, npx_log2, [seg3], pair_tabs=pair["tabs"], defer=dict(rows=rows, k_pad=ru(c2)))
                seg["out"] = None
                lay1 = pair["lay1"]
                self._gemm(None, npx_log2, [seg], in_cols=lay1[0],
                           gx=dict(ta=pair["ta"], tb=pair["tb"], coff=pair["off1"], k_pad=ru(lay1[1]), rows=rows, mode=0,
                                   add=pair["add1"], vv=pair["vv"]), pair_tabs=pair["tabs"], chain=layer2)
                return
        if has_rest:
            h2 = self._buf(rows, c2, cm=npx_log2 >= 7)
            seg["out"] = h2
            first_gemm(seg)
            c3 = sd[pfx + ".rest_mlp.0.weight"].shape[0]
            assert np.array_equal(gn_layout(c2)[0], np.arange(c2)), "second_mlp width with padded GroupNorm groups"
            seg3 = with_res(dict(w=self._w(pfx + ".rest_mlp.0.weight"), bias=sd[pfx + ".rest_mlp.0.bias"], mode=EPI_NORM,
                                 flags=F_POST_RELU, layout=gn_layout(c3), out=final_out, out_coff=final_coff,
                                 gn=(sd[pfx + ".rest_mlp.1.group_norm.weight"], sd[pfx + ".rest_mlp.1.group_norm.bias"])))
            self._gemm(h2, npx_log2, [seg3], pair_tabs=None if pair is None else pair["tabs"])
        else:
            with_res(seg)
            seg["out"] = final_out
            seg["out_coff"] = final_coff
            first_gemm(seg)

    def _sa_chain_shapes(self, pfx, npx_log2):
        """does SLIDE_OP_SA_CHAIN cover this Mlp's second_mlp -> rest_mlp (widths, identity GroupNorm layouts, group sizes)?"""
        sd = self.sd
        if (npx_log2 != 8 or os.environ.get("SLIDE_SA_CHAIN", "1") == "0" or self.prec != 1 or not self.use_gx
                or (pfx + ".rest_mlp.0.weight") not in sd):
            return False
        c1 = sd[pfx + ".first_mlp.0.weight"].shape[0]
        c2 = sd[pfx + ".second_mlp.0.weight"].shape[0]
        c3 = sd[pfx + ".rest_mlp.0.weight"].shape[0]
        l1, l2, l3 = gn_layout(c1), gn_layout(c2), gn_layout(c3)
        ident = lambda l, c: np.array_equal(l[0], np.arange(c)) and l[1] == c and l[2] == c
        return bool(c2 in (128, 256) and c3 % 256 == 0 and c1 % 64 == 0 and ident(l1, c1) and ident(l2, c2) and ident(l3, c3)
                    and l2[3] in (4, 8, 16) and l3[3] in (4, 8, 16))

    def _sa_chain(self, pfx, npx_log2, pair, cvec, seg, final_out, final_coff):
        """second_mlp -> rest_mlp of an SA block as ONE launch (SLIDE_OP_SA_CHAIN, csrc/gemm_gx.hip: h2 stays in registers).
        Returns False when the shapes are outside what the kernel covers (the two-launch path then runs)."""
        sd, B = self.sd, self.B
        if pair["vv"] is not None or not self._sa_chain_shapes(pfx, npx_log2):
            return False
        w1, w2 = self._w(pfx + ".second_mlp.0.weight"), self._w(pfx + ".rest_mlp.0.weight")
        c2, c1 = w1.shape
        c3 = w2.shape[0]
        lay1, l2, l3 = pair["lay1"], gn_layout(c2), gn_layout(c3)
        assert np.array_equal(lay1[0], np.arange(c1)) and lay1[1] == c1
        if not (self._is_cm(final_out) and final_coff == 0 and final_out.shape[1] == c3):
            return False
        cm = lambda w: np.ascontiguousarray(w.reshape(w.shape[0], -1, 32).transpose(1, 0, 2))
        vec = lambda b_, g_, bt_: np.stack([b_, g_, bt_]).astype(np.float32)
        d = [self.A.put(cm(w1), torch.float16), self.A.put(cm(w2), torch.float16),
             self.A.put(vec(sd[pfx + ".second_mlp.0.bias"], *seg["gn"])),
             self.A.put(vec(sd[pfx + ".rest_mlp.0.bias"], sd[pfx + ".rest_mlp.1.group_norm.weight"],
                            sd[pfx + ".rest_mlp.1.group_norm.bias"]))]
        add0, add1 = pair["add1"], seg.get("addvec")
        ta, tb = pair["ta"], pair["tb"]
        rows = B * 256
        fl = 2 * rows * (w1.size + w2.size)
        self.gemm_flops[len(self.ops)] = fl
        self.gemm_bytes[len(self.ops)] = (2 * B * 16 * (c1 + c3) * 2 + (w1.size + w2.size) * 2, rows * c3 * 2)
        self.flops += fl
        self.kernel_names[len(self.ops)] = "sa_chain_kernel<%d>" % (c2 // 32)
        self._emit(make_op(OP_SA_CHAIN,
                           i=(B, ta.shape[1], c1, c2, c3, l2[3], l3[3], 0 if add0 is None else add0[4], 0 if add0 is None else add0[2],
                              0 if add1 is None else add1[2]),
                           f=(1.0 / (l2[4] * 256), 1.0 / (l3[4] * 256), 1.0 if self._is_fm(final_out) else 0.0),
                           p=(ta.data_ptr() + 2 * pair["off1"], tb.data_ptr() + 2 * pair["off1"],
                              ta.data_ptr() + 2 * pair["offr"], tb.data_ptr() + 2 * pair["offr"],
                              d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(),
                              None if add0 is None else add0[0].data_ptr() + 4 * add0[1],
                              None if add0 is None or add0[3] is None else add0[3].data_ptr(),
                              None if add1 is None else add1[0].data_ptr() + 4 * add1[1], final_out.data_ptr())))
        return True

    def _attention_query(self, apfx, K):
        """buffers + GEMM segment of an attention block's per-point query branch (feat_conv): it depends on the block's input
        table only, so the query GEMMs of two blocks that read the same table (an SA block and the FP block that takes the
        same features as its skip input) are issued as ONE launch by whichever block comes first"""
        sd, B = self.sd, self.B
        C1 = sd[apfx + ".feat_conv.weight"].shape[0]
        C2 = sd[apfx + ".grouped_feat_conv.weight"].shape[0]
        ldT = ru(C1) + ru(C2)
        Tq = self.A.zeros(B * 16, ru(C1), dtype=self.adt)
        ssum, ssq = self.A.zeros(B, ldT), self.A.zeros(B, ldT)
        qseg = dict(w=self._w(apfx + ".feat_conv.weight"), bias=sd[apfx + ".feat_conv.bias"], mode=EPI_STATS,
                    flags=F_PRE_RELU, out=Tq, stats=(ssum, ssq, 0, float(K)))
        return dict(Tq=Tq, ssum=ssum, ssq=ssq, qseg=qseg)

    def _pair_first(self, npx_log2, K, feat_in, C, segs, coords, fin=None, lead_segs=()):
        """PAIR DECOMPOSITION of a block's shared first layer (csrc/gemm_gx.hip): the 1x1 convolutions over the grouped input
        [neighbour features (C) | coordinate channels] are linear, so their output for row (point p, slot j) is a[q] + b[p]
        (+ d2 vd + w vw), q the slot's neighbour.  Emits the 16-row GEMM y = Wf . feat + bias (1/K of the MACs) and
        SLIDE_OP_PAIR_NORM (tables a, b in fp16 with the GroupNorm of NORM segments folded in, sums of STATS segments);
        the K-expanded outputs are never stored.  segs: the segment dicts of the old shared GEMM (w over all grouped
        channels); coords: dict(rel, abs, ctr = first column of each 3-channel coordinate group, d2 / w = column or None).
        Returns the context the consumers (generated-X GEMMs, PAIR residuals) take."""
        B = self.B
        offs, ysegs, off = [], [], 0
        for sg in segs:
            lay = sg.get("layout")
            Op = ru(sg["w"].shape[0] if lay is None else lay[1])
            offs.append(off)
            off += Op
        ldy = off
        # SLIDE_PAIR_FUSED (default on): the per-point GEMM and the pair-table pass as ONE launch (SLIDE_OP_PAIR_FIRST,
        # csrc/engine.hip pair_first_kernel) -- y never goes through memory; bit-identical to the two-launch form
        fused = fin is None and self.prec == 1 and self.use_glds and os.environ.get("SLIDE_PAIR_FUSED", "1") != "0"
        Y = None if fused else self.A.zeros(B * 16, ldy)  # fp32
        wa, wb, vv_in = np.zeros((ldy, 4), np.float32), np.zeros((ldy, 4), np.float32), np.zeros((2, ldy), np.float32)
        psegs = []
        for sg, o_ in zip(segs, offs):
            lay = sg.get("layout")
            w = sg["w"]
            oidx = (np.arange(w.shape[0]) if lay is None else lay[0]) + o_
            ysegs.append(dict(w=w[:, :C], bias=sg.get("bias"), mode=EPI_RAW, out=Y, out_coff=o_,
                              layout=None if lay is None else (lay[0], lay[1], 0, 1, 1)))
            rel, ab, ctr = (w[:, coords[k_]:coords[k_] + 3] for k_ in ("rel", "abs", "ctr"))
            wa[oidx, :3] = rel + ab
            wb[oidx, :3] = ctr - rel
            if coords.get("d2") is not None:
                vv_in[0, oidx] = w[:, coords["d2"]]
                vv_in[1, oidx] = w[:, coords["w"]]
            ps = dict(sg)
            ps["w"] = w[:, :0]  # descriptors only: mode / flags / GroupNorm parameters / statistics
            ps["out"] = None
            psegs.append(ps)
        assert sum(3 for _ in ("rel", "abs", "ctr")) + (2 if coords.get("d2") is not None else 0) + C == segs[0]["w"].shape[1]
        tdt = torch.float16 if self.prec == 1 else torch.float32  # (split plans: float tables)
        ta = self.A.zeros(B * 16, ldy, dtype=tdt)
        tb = self.A.zeros(B * 16, ldy, dtype=tdt)
        fp = K == 8
        d = [self.A.put(wa), self.A.put(wb)]
        vv = rvv_all = None
        if fp:
            d.append(self.A.put(vv_in))
            vv = self.A.zeros(B, 2, ldy)
        if fused:
            fsegs = []
            for sg, o_ in zip(segs, offs):
                fs = dict(sg)
                fs.update(w=sg["w"][:, :C], out=ta, out_coff=o_, npx=1 << npx_log2)
                for k_ in ("addvec", "residual", "res_pair", "pre_add"):
                    assert fs.get(k_) is None or k_ == "addvec"
                fs.pop("addvec", None)  # (first_mlp's t-embedding rows are added by the consumers, after the ReLU)
                fsegs.append(fs)
            lead_cobs = sum(ru(sg["w"].shape[0] if sg.get("layout") is None else sg["layout"][1]) for sg in lead_segs) // 32
            self._gemm(feat_in, 4, list(lead_segs) + fsegs,
                       pair_fused=dict(cob0=lead_cobs, ld=ldy, K=K, wa=d[0], wb=d[1], ta=ta, tb=tb,
                                       nbr=self.kidx if fp else None, d2=self.kd2 if fp else None, w=self.kw if fp else None,
                                       vv_in=d[2] if fp else None, vv=vv))
            return dict(ta=ta, tb=tb, vv=vv, offs=offs, ldy=ldy, rows=B * 16 * K, vv_in=vv_in,
                        tabs=(self.kidx, self.kd2, self.kw) if fp else None)
        # (lead_segs: other per-point GEMM segments over the same table -- the attention queries -- ride on this launch)
        self._gemm(feat_in, 4, list(lead_segs) + ysegs)
        ed = self._epi_only(psegs, 1 << npx_log2)
        # loop-invariant when the coordinates are a fixed condition?  No: y changes every step.
        v2 = ldy <= 2048 and (os.environ.get("SLIDE_PAIR_NORM_V2", "0") != "0" or self.prec != 1)
        assert (fin is None or v2) and (self.prec == 1 or v2)
        self.kernel_names[len(self.ops)] = "pair_norm2_kernel<%s, %s>" % ("true" if fp else "false", "_Float16" if self.prec == 1 else "float")
        self._emit(make_op(OP_PAIR_NORM, i=(B, ldy, K, 2 if v2 else 1, int(self.prec != 1)),
                           p=(Y.data_ptr(), self.xyz.data_ptr(), d[0].data_ptr(), d[1].data_ptr(), ed.data_ptr(),
                              ta.data_ptr(), tb.data_ptr(), self.kidx.data_ptr() if fp else None,
                              self.kd2.data_ptr() if fp else None, self.kw.data_ptr() if fp else None,
                              d[2].data_ptr() if fp else None, vv.data_ptr() if fp else None,
                              None if fin is None else fin.data_ptr())))
        return dict(ta=ta, tb=tb, vv=vv, offs=offs, ldy=ldy, rows=B * 16 * K, vv_in=vv_in,
                    tabs=(self.kidx, self.kd2, self.kw) if fp else None)

    def _epi_only(self, segs, npx):
        """device array of SlideEpi descriptors (one per 32 physical channels) carrying only what SLIDE_OP_PAIR_NORM reads:
        mode, flags, GroupNorm layout / parameters, statistics pointers"""
        blocks = []
        for sg in segs:
            lay = sg.get("layout")
            O = sg["w"].shape[0]
            oidx, Op, n_norm_p, gs_p, gs_l = (np.arange(O), O, 0, 1, 1) if lay is None else lay
            Opad = ru(Op)
            vec = np.zeros((2, Opad), np.float32)
            if sg.get("gn") is not None:
                gam, bet = sg["gn"]
                vec[0, oidx[:gam.shape[0]]] = gam
                vec[1, oidx[:gam.shape[0]]] = bet
            vd = self.A.put(vec)
            for j in range(Opad // 32):
                e = SlideEpi()
                e.mode = sg.get("mode", EPI_RAW)
                e.flags = sg.get("flags", 0)
                e.gs = gs_p
                e.n_norm = int(min(32, max(0, n_norm_p - 32 * j)))
                e.inv_count = 1.0 / (gs_l * npx)
                e.gamma = vd.data_ptr() + 4 * (32 * j)
                e.beta = vd.data_ptr() + 4 * (Opad + 32 * j)
                if sg.get("stats") is not None:
                    ssum, ssq, scoff, scale = sg["stats"]
                    e.stats_sum = ssum.data_ptr() + 4 * (scoff + 32 * j)
                    e.stats_sq = ssq.data_ptr() + 4 * (scoff + 32 * j)
                    e.stats_bs = ssum.shape[1]
                    e.stats_scale = scale
                blocks.append(bytes(e))
        return self.A.put(np.frombuffer(b"".join(blocks), dtype=np.uint8).copy())

    def _body_shape(self, mpfx, apfx, npx_log2):
        """(rest, nb1, nbm, nbu) if SLIDE_OP_BLOCK_BODY has an instantiation for this block's widths, else None"""
        sd = self.sd
        # OPT-IN since round 5 (SLIDE_BODY=1, experiments build): with the split position plan beside the feature chains the one-launch
        # body of the FP0-sized blocks LOSES 0.9 % to its three separate launches (378.5 vs 382.0 shapes/s, four alternating pairs) --
        # one workgroup per CU at 0.024 MFMA-busy holds a CU for 31 us where the ring kernels' small workgroups interleave
        if not self.use_gx or os.environ.get("SLIDE_BODY", "0") == "0":
            return None
        rest = (mpfx + ".rest_mlp.0.weight") in sd
        c1 = sd[mpfx + ".first_mlp.0.weight"].shape[0]
        c2 = sd[mpfx + ".second_mlp.0.weight"].shape[0]
        n_mo = sd[mpfx + ".rest_mlp.0.weight"].shape[0] if rest else c2
        inter = sd[apfx + ".weight_conv.2.weight"].shape[0]
        cout = sd[apfx + ".weight_conv.5.weight"].shape[0]
        lay_u = gn_layout(inter)
        n_u = ru(lay_u[1])
        ident = lambda c: (lambda l: np.array_equal(l[0], np.arange(c)) and l[1] == c and l[2] == c)(gn_layout(c))
        okgs = lambda c: gn_layout(c)[3] in (4, 8, 16)
        if not (ident(c1) and ident(c2) and ident(n_mo) and ident(cout) and okgs(c2) and okgs(n_mo) and lay_u[3] in (4, 8, 16)
                and c1 % 32 == 0 and n_mo % 32 == 0 and cout % 64 == 0 and gn_layout(cout)[3] <= 32):
            return None
        shape = (npx_log2, rest, (c2 // 32) if rest else 0, n_mo // 32, n_u // 32)
        # ((7, False, 0, 8, 8) -- FP1 -- was built and measured slower than its three separate launches: one wave per SIMD)
        # ((8, True, 4, 8, 5) -- SA0 -- spills under its 256-register budget: opt-in with SLIDE_BODY=2)
        if shape not in (((7, False, 0, 4, 4), (8, True, 4, 8, 5)) if os.environ.get("SLIDE_BODY", "0") == "2" else ((7, False, 0, 4, 4),)):
            return None
        return shape

    def _emit_block_body(self, body, mpfx, npx_log2, K, out, cvec):
        """SLIDE_OP_BLOCK_BODY (csrc/block_body.hip): Mlp tail -> mo, keys -> u, attention tail, one workgroup per sample,
        mo / u in registers.  body: state left by _attention's score closure (P, joint GroupNorm rows, ...); ctx: pair tables"""
        sd, B = self.sd, self.B
        ctx = self.pair_ctx
        apfx = body["apfx"]
        npx = 1 << npx_log2
        rest = (mpfx + ".rest_mlp.0.weight") in sd
        cm = lambda w: np.ascontiguousarray(w.reshape(w.shape[0], -1, 32).transpose(1, 0, 2))
        vec3 = lambda n, lay, b_, g_, bt_: (lambda v: (v.__setitem__((0, lay[0]), b_), v.__setitem__((1, lay[0][:g_.shape[0]]), g_),
                                                    v.__setitem__((2, lay[0][:bt_.shape[0]]), bt_), v)[-1])(np.zeros((3, n), np.float32))
        a = BodyArgs()
        keep = []
        slots = []
        put16 = lambda w: (lambda t: (keep.append(t), t)[1])(self.A.put(cm(w), torch.float16))
        putf = lambda v: (lambda t: (keep.append(t), t)[1])(self.A.put(np.ascontiguousarray(v, np.float32)))

        # slot geometry (csrc/block_body.hip): eight-wave form (16 x 16-row samples) 128-row slabs / 64-channel column blocks,
        # four-wave form everything at once: 256-row images, one 32-deep chunk per slot
        srows, trows = (128, 64) if npx_log2 == 8 else (256, 128)

        def slab_slots(Wt, n, k):  # D[channel][row] stage
            cps = 256 // srows
            for r0 in range(0, n, srows):
                for kc in range(0, k // 32, cps):
                    slots.append((Wt.data_ptr() + 2 * ((kc * n + r0) * 32), n * 32, min(srows, n - r0), 0, min(cps, k // 32 - kc)))

        c1 = sd[mpfx + ".first_mlp.0.weight"].shape[0]
        w_sec = self._w(mpfx + ".second_mlp.0.weight")
        c2 = w_sec.shape[0]
        lay2 = gn_layout(c2)
        fl = 0
        if rest:
            W1 = put16(w_sec)
            slab_slots(W1, c2, c1)
            a.vec1 = putf(vec3(c2, lay2, sd[mpfx + ".second_mlp.0.bias"], sd[mpfx + ".second_mlp.1.group_norm.weight"],
                               sd[mpfx + ".second_mlp.1.group_norm.bias"])).data_ptr()
            a.n1, a.gs1, a.inv1 = c2, lay2[3], 1.0 / (lay2[4] * npx)
            if (mpfx + ".fc_condition.weight") in sd:
                a.add1 = cvec.data_ptr() + 4 * self._cvec_off(mpfx + ".fc_condition", c2)
                a.add1_bs = self._c_bs
            w_m = self._w(mpfx + ".rest_mlp.0.weight")
            n_mo = w_m.shape[0]
            laym = gn_layout(n_mo)
            Wm = put16(w_m)
            slab_slots(Wm, n_mo, c2)
            a.vecm = putf(vec3(n_mo, laym, sd[mpfx + ".rest_mlp.0.bias"], sd[mpfx + ".rest_mlp.1.group_norm.weight"],
                               sd[mpfx + ".rest_mlp.1.group_norm.bias"])).data_ptr()
            fl += w_sec.size + w_m.size
        else:
            n_mo, laym = c2, lay2
            Wm = put16(w_sec)
            slab_slots(Wm, n_mo, c1)
            a.vecm = putf(vec3(n_mo, laym, sd[mpfx + ".second_mlp.0.bias"], sd[mpfx + ".second_mlp.1.group_norm.weight"],
                               sd[mpfx + ".second_mlp.1.group_norm.bias"])).data_ptr()
            if (mpfx + ".fc_condition.weight") in sd:
                a.addm = cvec.data_ptr() + 4 * self._cvec_off(mpfx + ".fc_condition", c2)
                a.addm_bs = self._c_bs
            fl += w_sec.size
        a.n_mo, a.gsm, a.invm = n_mo, laym[3], 1.0 / (laym[4] * npx)
        # keys -> u
        lay_u = body["lay_u"]
        n_u = ru(lay_u[1])
        C1, C2p = body["C1"], body["C2p"]
        w2k = body["w2"][:, C1:]
        wk = np.zeros((n_u, C2p), np.float32)
        wk[lay_u[0], :w2k.shape[1]] = w2k
        Wu = put16(wk)
        slab_slots(Wu, n_u, C2p)
        gu = sd[apfx + ".weight_conv.4.group_norm.weight"]
        a.vecu = putf(vec3(n_u, lay_u, sd[apfx + ".weight_conv.2.bias"], gu, sd[apfx + ".weight_conv.4.group_norm.bias"])).data_ptr()
        a.n_u, a.gsu, a.nnu, a.invu = n_u, lay_u[3], lay_u[2], 1.0 / (lay_u[4] * npx)
        fl += w2k.size
        # tail
        cout = body["cout"]
        vlay = gn_layout(cout)
        w5 = np.zeros((cout, n_u), np.float32)
        w5[:, lay_u[0]] = self._w(apfx + ".weight_conv.5.weight")
        wv = self._w(apfx + ".feat_out_conv.0.weight")
        W5, Wv = put16(w5), put16(wv)
        assert cout % trows == 0 or npx_log2 == 8
        for cbk in range((cout + trows - 1) // trows):
            for Wt, k in ((W5, n_u), (Wv, n_mo)):
                cps = 256 // trows
                for kc in range(0, k // 32, cps):
                    slots.append((Wt.data_ptr() + 2 * ((kc * cout + cbk * trows) * 32), cout * 32, min(trows, cout - cbk * trows), 1,
                                  min(cps, k // 32 - kc)))
        vect = np.zeros((4, cout), np.float32)
        vect[0] = sd[apfx + ".weight_conv.5.bias"]
        vect[1] = sd[apfx + ".feat_out_conv.0.bias"]
        gam = sd[apfx + ".feat_out_conv.1.group_norm.weight"]
        vect[2, :gam.shape[0]] = gam
        vect[3, :gam.shape[0]] = sd[apfx + ".feat_out_conv.1.group_norm.bias"]
        a.vect = putf(vect).data_ptr()
        a.n_out, a.gsv, a.nnv, a.invv = cout, vlay[3], vlay[2], 1.0 / (vlay[4] * npx)
        fl += self._w(apfx + ".weight_conv.5.weight").size + wv.size
        # tables and vectors
        ta, tb = ctx["ta"], ctx["tb"]
        a.ta, a.tb, a.t_ld = ta.data_ptr(), tb.data_ptr(), ta.shape[1]
        a.off1, a.k1, a.offr, a.offk, a.kk = ctx["off1"], c1, ctx["offr"], ctx["offk"], C2p
        if K == 8:
            a.vv, a.vbs = ctx["vv"].data_ptr(), 2 * ctx["vv"].shape[2]
            a.rv = ctx["rvv"].data_ptr()
            a.nbr, a.d2, a.w = self.kidx.data_ptr(), self.kd2.data_ptr(), self.kw.data_ptr()
            assert ctx["rvv"].shape[1] == n_mo
        add0 = ctx["add1"]
        if add0 is not None:
            a.add0 = add0[0].data_ptr() + 4 * add0[1]
            a.add0_bs, a.add0_stride = add0[2], add0[4]
            a.add0_idx = None if add0[3] is None else add0[3].data_ptr()
        a.sc = body["scale"].data_ptr() + 4 * body["C1p"]
        a.sh = body["shift"].data_ptr() + 4 * body["C1p"]
        a.aff_bs = body["ldT"]
        a.P, a.p_ld = body["P"].data_ptr(), body["P"].shape[1]
        assert body["P"].shape[1] >= n_u and out.dtype == torch.float16 and not self._is_cm(out)
        a.out, a.out_ld = out.data_ptr(), out.shape[1]
        a.B = B
        sl = (BodySlot * len(slots))()
        for i, (src, cs, nr, kind, nv) in enumerate(slots):
            sl[i].src, sl[i].chunk_stride, sl[i].nrows, sl[i].kind, sl[i].nvalid = src, cs, nr, kind, nv
        sd_ = self.A.put(np.frombuffer(bytes(sl), dtype=np.uint8).copy())
        a.slots, a.n_slots = sd_.data_ptr(), len(slots)
        self.A.keep.append((a, keep))  # the op carries a HOST pointer to the argument block
        rows = B * 16 * K
        self.gemm_flops[len(self.ops)] = 2 * rows * fl
        self.gemm_bytes[len(self.ops)] = (2 * B * 16 * ta.shape[1] * 2, B * 16 * cout * 2)
        self.flops += 2 * rows * fl
        self._tail_of[out.data_ptr()] = len(self.ops)
        self._body_args[len(self.ops)] = a
        self.kernel_names[len(self.ops)] = "block_body_kernel<%d, %s, %d, %d, %d>" % (
            npx_log2, "true" if rest else "false", (c2 // 32) if rest else 0, n_mo // 32, n_u // 32)
        self._emit(make_op(OP_BLOCK_BODY, i=(npx_log2, int(rest)), p=(ctypes.addressof(a),)))

    def _attention(self, apfx, npx_log2, K, g, q_in, mo, mlp_first, mlp_res, out, out_ld_buf, gather=None, qctx=None,
                   extra_q=(), pair=None, body=None):
        """AttentionModule (attention.py:35-96).  g: grouped input [B*npx][ldg]; q_in: query features [B*16][ld];
        mo: the Mlp output buffer (values input), produced by the caller AFTER the shared first GEMM.

        total_feat = [feat_conv(query) broadcast over the K neighbours | grouped_feat_conv(g)] is never materialised:
        the query half is kept per point ([B*16][C1]); its GroupNorm statistics are its per-point sums x K; and
        because weight_conv.2 is linear, its query half is evaluated once per POINT (a 16-row GEMM) and enters the
        per-neighbour GEMM as a pre-activation add -- 1/K of the reference's MACs for that half.
        Returns a closure continuing after the caller has produced `mo`."""
        sd, B = self.sd, self.B
        rows = B * 16 * K
        npx = 1 << npx_log2
        kshift = {8: 3, 16: 4}[K]
        C1 = sd[apfx + ".feat_conv.weight"].shape[0]
        C2 = sd[apfx + ".grouped_feat_conv.weight"].shape[0]
        inter = sd[apfx + ".weight_conv.2.weight"].shape[0]
        cout = sd[apfx + ".weight_conv.5.weight"].shape[0]
        C1p, C2p = ru(C1), ru(C2)
        ldT = C1p + C2p                      # physical channel space of the (virtual) concatenation
        issued = qctx is not None  # the query GEMM rode on an earlier block's launch (same input table)
        if qctx is None:
            qctx = self._attention_query(apfx, K)
        Tq, ssum, ssq = qctx["Tq"], qctx["ssum"], qctx["ssq"]
        Tk = None if pair is not None else self._buf(rows, C2p, cm=True)
        kseg = dict(w=self._w(apfx + ".grouped_feat_conv.weight"), bias=sd[apfx + ".grouped_feat_conv.bias"],
                    mode=EPI_STATS, flags=F_PRE_RELU, out=Tk, stats=(ssum, ssq, C1p, 1.0))
        # GroupNorm over the concatenation [q | k] (weight_conv.1): groups may straddle the two producers
        Ct = C1 + C2
        G = min(32, Ct)
        n_norm = Ct - Ct % G
        gs = n_norm // G
        phys = np.array([c if c < C1 else C1p + (c - C1) for c in range(Ct)], np.int64)
        gid = np.full(ldT, -1, np.int32)
        gam, bet = np.zeros(ldT, np.float32), np.zeros(ldT, np.float32)
        gid[phys[:n_norm]] = np.arange(n_norm) // gs
        gam[phys[:n_norm]] = sd[apfx + ".weight_conv.1.group_norm.weight"]
        bet[phys[:n_norm]] = sd[apfx + ".weight_conv.1.group_norm.bias"]
        gstart = np.array([phys[gq * gs] for gq in range(G)], np.int32)
        gend = np.array([phys[(gq + 1) * gs - 1] + 1 for gq in range(G)], np.int32)
        scale, shift = self.A.zeros(B, ldT), self.A.zeros(B, ldT)
        fin_d = [self.A.put(a) for a in (gid, gstart, gend, gam, bet)]

        def fin_struct():
            fin = SlideGnFin()
            for n_, t_ in zip(("sum", "sq", "gid", "gstart", "gend", "gamma", "beta", "scale", "shift"),
                              (ssum, ssq, fin_d[0], fin_d[1], fin_d[2], fin_d[3], fin_d[4], scale, shift)):
                setattr(fin, n_, t_.data_ptr())
            fin.inv_count, fin.C, fin.bs, fin.G = 1.0 / (gs * npx), ldT, ldT, G
            return self.A.put(np.frombuffer(bytes(fin), dtype=np.uint8).copy())

        # lane 1 (query / score branch) forks here: it only needs the module inputs
        self._sync(0, 1)
        q_rides = pair is not None and not issued and q_in is pair[0] and not self.two_lanes
        if not issued and not q_rides:
            self._lane = 1
            self._gemm(q_in, 4, [qctx["qseg"]] + [e["qseg"] for e in extra_q])
            self._lane = 0
        pair_fin = False
        # shared-input GEMM: [first_mlp | res_connect | grouped_feat_conv]
        if pair is not None:
            feat_tab, Cf, coords = pair
            # the per-sample table pass also finalises the joint GroupNorm (SLIDE_OP_PAIR_NORM version 2)
            # (SLIDE_PAIR_NORM_V2=1, opt-in: one 1024-thread workgroup per sample that also finalises the joint GroupNorm -- 3 %
            #  faster for a single chain, 5 % slower with four chains in flight: sixteen waves must find room on ONE CU)
            pair_fin = ldT <= 2048 and (os.environ.get("SLIDE_PAIR_NORM_V2", "0") != "0" or self.use_gxs)
            ctx = self._pair_first(npx_log2, K, feat_tab, Cf, [mlp_first, mlp_res, kseg], coords,
                                   fin=fin_struct() if pair_fin else None,
                                   lead_segs=([qctx["qseg"]] + [e["qseg"] for e in extra_q]) if q_rides else ())
            ctx.update(off1=ctx["offs"][0], offr=ctx["offs"][1], offk=ctx["offs"][2], lay1=mlp_first["layout"],
                       add1=mlp_first.get("addvec"))
            rres = ru(mlp_res["w"].shape[0])
            ctx["rvv"] = None
            if K == 8:  # coefficient vectors of the two per-slot scalars for the res_connect channels (RAW segment: unscaled)
                ctx["rvv"] = self.A.put(np.ascontiguousarray(ctx["vv_in"][:, ctx["offr"]:ctx["offr"] + rres]))
            self.pair_ctx = ctx
        elif gather is not None and gather[3] * 32 >= int(os.environ.get("SLIDE_SPLIT_FIRST", "1000000")):
            # (opt-in, SLIDE_SPLIT_FIRST=<min feature channels>: measured neutral to -1.5 % on the feature plan -- the
            # 256-row launch is bound by its epilogue, not by its K loop, and the per-point GEMM is one more launch)
            # The layer is linear in its input and the leading Cf input channels of row (point, neighbour) are the
            # NEIGHBOUR's feature row, the same for every query point: their products are evaluated once per point
            # (a 16-row GEMM over the feature table, 1/K of the MACs) and enter the per-neighbour GEMM -- now over the
            # coordinate channels only -- as a gathered pre-activation term.
            tab, kidx, _, nsplit = gather
            Cf = nsplit * 32
            segs, ysegs, off = [], [], 0
            for sg in (mlp_first, mlp_res, kseg):
                lay = sg.get("layout")
                Op = ru(sg["w"].shape[0] if lay is None else lay[1])
                segs.append((sg, off))
                off += Op
            Y = self._buf(B * 16, off)
            for sg, o_ in segs:
                lay = sg.get("layout")
                ysegs.append(dict(w=sg["w"][:, :Cf], mode=EPI_RAW, out=Y, out_coff=o_,
                                  layout=None if lay is None else (lay[0], lay[1], 0, 1, 1)))
            self._gemm(tab, 4, ysegs)
            tails = []
            for sg, o_ in segs:
                t_ = dict(sg)
                t_["w"] = sg["w"][:, Cf:]
                t_["pre_add"] = (Y, -kshift, o_)
                tails.append(t_)
            self._gemm(g, npx_log2, tails, pre_gather=kidx)
        else:
            self._gemm(g, npx_log2, [mlp_first, mlp_res, kseg], gather=gather)
        self._sync(0, 1)  # the key statistics are ready

        def finish_scores():
            # lane 1.  Joint GroupNorm of [q | k]: finalised by the pair-table pass, inside the per-point query GEMM below
            # (its small-launch kernel), or by its own launch
            self._lane = 1
            d = fin_d
            # (only the small-launch kernel finalises: same grid bound as run_gemm's dispatch)
            n_cob_p = ru(gn_layout(inter)[1]) // 32
            fuse_fin = (self.prec == 1 and self.use_glds and os.environ.get("SLIDE_FUSE_FIN", "1") != "0" and
                        ((B * 16 + 63) // 64) * ((n_cob_p + 1) // 2) <= 1024)
            gn_fin = None
            if pair_fin:
                pass
            elif fuse_fin:  # finalised inside the per-point query GEMM below (its small-launch kernel), one launch less
                gn_fin = fin_struct()
            else:
                self._emit(make_op(OP_FINALIZE_GN, i=(B, ldT, ldT), f=(1.0 / (gs * npx),),
                                        p=(ssum.data_ptr(), ssq.data_ptr(), d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(),
                                           d[3].data_ptr(), d[4].data_ptr(), scale.data_ptr(), shift.data_ptr())))
            lay = gn_layout(inter)
            w2 = self._w(apfx + ".weight_conv.2.weight")
            # query half, once per point: P = W2[:, :C1] . GN(relu(q))      (no bias, raw)
            P = self.A.zeros(B * 16, ru(lay[1]), dtype=self.adt)
            self._gemm(Tq, 4, [dict(w=w2[:, :C1], mode=EPI_RAW, layout=(lay[0], lay[1], 0, 1, 1), out=P)],
                       in_affine=(scale, shift, 0, ldT), gn_fin=gn_fin)
            if body is not None:  # the block body kernel (SLIDE_OP_BLOCK_BODY) takes it from here
                self._lane = 0
                body.update(P=P, lay_u=lay, scale=scale, shift=shift, C1=C1, C1p=C1p, C2p=C2p, ldT=ldT, w2=w2, apfx=apfx,
                            cout=cout, inter=inter)
                return ("body",)
            # neighbour half: u = GN4(relu(W2[:, C1:] . GN(relu(k)) + bias + P[point]))
            u = self._buf(rows, ru(lay[1]), cm=True, fm=self._is_fm(mo))
            useg = dict(w=w2[:, C1:], bias=sd[apfx + ".weight_conv.2.bias"], mode=EPI_NORM,
                        flags=F_PRE_RELU, layout=lay, out=u, pre_add=(P, kshift),
                        gn=(sd[apfx + ".weight_conv.4.group_norm.weight"], sd[apfx + ".weight_conv.4.group_norm.bias"]))
            if pair is not None:  # the keys max(a[q] + b[p], 0) * scale + shift are generated from the pair tables
                ctx = self.pair_ctx
                self._gemm(None, npx_log2, [useg], in_affine=(scale, shift, C1p, ldT),
                           gx=dict(ta=ctx["ta"], tb=ctx["tb"], coff=ctx["offk"], k_pad=C2p, rows=rows, mode=1, vv=ctx["vv"]),
                           pair_tabs=ctx["tabs"])
            else:
                self._gemm(Tk, npx_log2, [useg], in_affine=(scale, shift, C1p, ldT))
            vlay = gn_layout(cout)
            if (self.prec == 1 and self.use_glds and os.environ.get("SLIDE_ATTN_TAIL", "1") != "0" and
                    np.array_equal(vlay[0], np.arange(cout))):
                self._lane = 0
                return ("fused", u, lay)  # scores are computed inside the fused attention tail (finish)
            # round 5: the same tail in the split arithmetic on float rows (csrc/gemm_gxs.hip attn_tail_split_kernel;
            # SLIDE_TAIL_SPLIT=0: scores GEMM + values GEMM + combine launch).  Its single-accumulator split needs |w| < 32.
            if (self.use_gxs and os.environ.get("SLIDE_TAIL_SPLIT", "1") != "0" and np.array_equal(vlay[0], np.arange(cout))
                    and self._split1_ok(self._w(apfx + ".weight_conv.5.weight"), self._w(apfx + ".feat_out_conv.0.weight"))):
                self._lane = 0
                return ("fused", u, lay)
            S = self._buf(rows, cout)
            self._gemm(u, npx_log2, [dict(w=self._w(apfx + ".weight_conv.5.weight"), bias=sd[apfx + ".weight_conv.5.bias"],
                                          mode=EPI_RAW, out=S)], in_cols=lay[0])
            self._lane = 0
            return S

        def finish(S):
            if isinstance(S, tuple):  # fused tail: scores GEMM + values GEMM + softmax-weighted sum in one launch
                _, u, lay = S
                vlay = gn_layout(cout)
                Cp = ru(cout)
                w5 = np.zeros((Cp, u.shape[1]), np.float32)
                w5[np.ix_(np.arange(cout), lay[0])] = self._w(apfx + ".weight_conv.5.weight")
                wv_l = self._w(apfx + ".feat_out_conv.0.weight")
                wv = np.zeros((Cp, mo.shape[1]), np.float32)
                wv[:cout, :wv_l.shape[1]] = wv_l
                vec = np.zeros((4, Cp), np.float32)
                vec[0, :cout] = sd[apfx + ".weight_conv.5.bias"]
                vec[1, :cout] = sd[apfx + ".feat_out_conv.0.bias"]
                gam = sd[apfx + ".feat_out_conv.1.group_norm.weight"]
                vec[2, :gam.shape[0]] = gam
                vec[3, :gam.shape[0]] = sd[apfx + ".feat_out_conv.1.group_norm.bias"]
                cmw = (lambda w: np.ascontiguousarray(w.reshape(w.shape[0], -1, 32).transpose(1, 0, 2))) if self.use_cm else (lambda w: w)
                wdt = torch.float16 if self.prec == 1 else torch.float32  # (split tail: float row-major weights)
                d = [self.A.put(cmw(w5), wdt), self.A.put(cmw(wv), wdt), self.A.put(vec)]
                if self.prec != 1:
                    self.kernel_names[len(self.ops)] = "attn_tail_split_kernel<%d>" % npx_log2
                assert out.dtype == self.adt and u.dtype == self.adt and mo.dtype == self.adt and not self._is_cm(out)
                assert self._is_fm(u) == self._is_fm(mo)
                self._sync(1, 0)
                self.flops += 2 * rows * (w5.size + wv.size)
                self.gemm_flops[len(self.ops)] = 2 * rows * cout * (len(lay[0]) + wv_l.shape[1])  # logical channels
                out_cm = None
                if self.use_cm and os.environ.get("SLIDE_CM_TABLES", "0") != "0" and npx_log2 == 8:  # (SA outputs feed gathers)
                    out_cm = self._cm_copy[out.data_ptr()] = self.A.zeros(out.shape[0], out.shape[1], dtype=self.adt)
                self._tail_of[out.data_ptr()] = len(self.ops)
                self._emit(make_op(OP_ATTN_TAIL, i=(rows, self._ldp(u), u.shape[1], self._ldp(mo), mo.shape[1], Cp // 32, npx_log2,
                                                    vlay[3], vlay[2], out.shape[1]),
                                        # f[1]: 1 = chunk-major operands, + 2 = two-stage ring at three workgroups per CU (opt-in,
                                        # SLIDE_TAIL_OCC3=1: measured neutral, 373.0 vs 372.3 shapes/s)
                                        f=(1.0 / (vlay[4] * npx), (1.0 if self.use_cm else 0.0) + (8.0 if self.prec != 1 else 0.0) +
                                           (2.0 if os.environ.get("SLIDE_TAIL_OCC3", "0") != "0" else 0.0) +
                                           (4.0 if (Cp // 32) % 4 == 0 and Cp // 32 >= int(os.environ.get("SLIDE_TAIL_WIDE", "1000")) else 0.0) +
                                           (16.0 if self._is_fm(mo) else 0.0)),  # bit 4: u / mo fragment-major
                                        p=(u.data_ptr(), d[0].data_ptr(), mo.data_ptr(), d[1].data_ptr(), out.data_ptr(),
                                           d[2].data_ptr(), None if out_cm is None else out_cm.data_ptr())))
                return
            # lane 0 (value branch), then the join
            V = self._buf(rows, cout)
            self._gemm(mo, npx_log2, [dict(w=self._w(apfx + ".feat_out_conv.0.weight"), bias=sd[apfx + ".feat_out_conv.0.bias"],
                                           mode=EPI_NORM, flags=F_POST_RELU, layout=gn_layout(cout), out=V,
                                           gn=(sd[apfx + ".feat_out_conv.1.group_norm.weight"],
                                               sd[apfx + ".feat_out_conv.1.group_norm.bias"]))])
            assert out.dtype == self.adt
            self._sync(1, 0)
            self._emit(make_op(OP_ATTN_COMBINE, i=(rows // K, cout, S.shape[1], V.shape[1], out.shape[1], K, self.prec),
                                    p=(S.data_ptr(), V.data_ptr(), out.data_ptr())))
        return (finish_scores, finish), cout

    def _grouped_input(self, kind, feat_in, C, Cg, K):
        """the grouped input of an SA / FP block.  fp16 LDS-DMA path: only the last chunks (left-over feature columns +
        coordinate channels) are assembled; the GEMM gathers the leading 32-column chunks from the neighbours' rows of
        `feat_in` itself (returns (tail buffer, gather tuple)).  Otherwise the whole [rows][Cg] matrix is assembled."""
        B = self.B
        rows = B * 16 * K
        ldg = ru(Cg)
        nsplit = 0
        if self.prec == 1 and self.use_glds and os.environ.get("SLIDE_GATHER", "1") != "0":
            nsplit = (C // 8) * 8 // 32
        c_begin = nsplit * 32
        g = self.A.zeros(rows, ldg - c_begin, dtype=self.adt)
        ptrs = (self.xyz.data_ptr(), feat_in.data_ptr(), self.kidx.data_ptr())
        if kind == OP_ASSEMBLE_FP:
            ptrs += (self.kd2.data_ptr(),)
        if nsplit and c_begin == C:  # only coordinate channels left: loop-invariant when the coordinates are a fixed condition
            self.xyz_copy_idx.append(len(self.ops))
        self._emit(make_op(kind, i=(B, C, feat_in.shape[1], ldg, K, self.prec, c_begin, ldg - c_begin), p=ptrs + (g.data_ptr(),)))
        return g, ((feat_in, self.kidx, K, nsplit) if nsplit else None), rows

    def _sa_module(self, i, feat_in, C, extra_q=()):
        sd, B = self.sd, self.B
        pfx = "SA_modules.%d" % i
        mp, ap = pfx + ".mlps.0", pfx + ".attention_modules.0"
        K = 16
        rows = B * 16 * K
        Cg = C + 9
        assert sd[mp + ".first_mlp.0.weight"].shape[1] == Cg
        c1 = sd[mp + ".first_mlp.0.weight"].shape[0]
        c_last = sd[mp + ".res_connect.weight"].shape[0]
        mo = self._buf(rows, c_last, cm=True, fm=self._tail_fm(mp, ap, 8, c_last))
        out = self._buf(B * 16, c_last)
        if self.use_gx or self.use_gxs:
            # pair decomposition (csrc/gemm_gx.hip): no grouped input, no h1 / r / key buffers; rows in natural neighbour order
            first, res = self._mlp_segments(mp, self.tvec, self.cvec, None, None)
            pair = (feat_in, C, dict(rel=C, abs=C + 3, ctr=C + 6))
            body = {} if self._body_shape(mp, ap, 8) is not None else None
            (scores, finish), cout = self._attention(ap, 8, K, None, feat_in, mo, first, res, out, None, extra_q=extra_q, pair=pair,
                                                     body=body)
            S = scores()
            if body is not None:
                self._emit_block_body(body, mp, 8, K, out, self.cvec)
                return out, cout
            self._mlp_tail(mp, 8, None, self.cvec, None, mo, pair=self.pair_ctx)
            finish(S)
            return out, cout
        g, gather, _ = self._grouped_input(OP_ASSEMBLE_SA, feat_in, C, Cg, K)
        h1, r = self._buf(rows, c1, cm=True), self._buf(rows, c_last, cm=True)
        first, res = self._mlp_segments(mp, self.tvec, self.cvec, h1, r)
        (scores, finish), cout = self._attention(ap, 8, K, g, feat_in, mo, first, res, out, None, gather=gather, extra_q=extra_q)
        S = scores()                                  # lane 1: finalize, P, weight_conv.2, weight_conv.5
        self._mlp_tail(mp, 8, h1, self.cvec, r, mo)   # lane 0: second / rest mlp
        finish(S)                                     # lane 0: values; join; softmax-combine
        return out, cout

    def _fp_module(self, j, U, CU, Kf, C2, out_buf=None, qctx=None):
        """PointnetKnnFPModule.forward (pointnet2_modules.py:771-873).  U: unknown (skip) features, Kf: known features."""
        sd, B = self.sd, self.B
        pfx = "FP_modules.%d" % j
        m1, m2, ap = pfx + ".mlp1", pfx + ".mlp2", pfx + ".attention_module"
        K = 8
        rows = B * 16 * K
        Cg = C2 + 11
        assert sd[m1 + ".first_mlp.0.weight"].shape[1] == Cg
        c1 = sd[m1 + ".first_mlp.0.weight"].shape[0]
        c_last = sd[m1 + ".res_connect.weight"].shape[0]
        mo = self._buf(rows, c_last, cm=True, fm=self._tail_fm(m1, ap, 7, c_last))
        pair = h1 = r = g = gather = None
        if self.use_gx or self.use_gxs:  # pair decomposition: group_knn's channels are [feats | d2 | w | abs | rel | centre]
            pair = (Kf, C2, dict(d2=C2, w=C2 + 1, abs=C2 + 2, rel=C2 + 5, ctr=C2 + 8))
        else:
            g, gather, _ = self._grouped_input(OP_ASSEMBLE_FP, Kf, C2, Cg, K)
            h1, r = self._buf(rows, c1, cm=True), self._buf(rows, c_last, cm=True)
        first, res = self._mlp_segments(m1, self.tvec, self.cvec, h1, r)
        # mlp2 input: [interpolated (c_last) | unknown feats (CU) | xyz (3)]  (pointnet2_modules.py:842-855)
        zin = c_last + CU + 3
        assert sd[m2 + ".first_mlp.0.weight"].shape[1] == zin
        Z = self._buf(B * 16, zin)
        body = {} if pair is not None and self._body_shape(m1, ap, 7) is not None else None
        (scores, finish), cout = self._attention(ap, 7, K, g, U, mo, first, res, Z, None, gather=gather, qctx=qctx, pair=pair,
                                                 body=body)
        S = scores()
        pctx = self.pair_ctx if pair is not None else None
        # skip features and coordinates: columns of Z the attention output does not touch -- on the score lane, beside
        # the value branch (joined by finish)
        es = Z.element_size()
        self._lane = 1
        # skip-feature columns: written by the producer of U itself where it can (the point preparation for the input
        # features, the fused attention tail of the SA block that made U) -- a COPY launch otherwise
        tail_idx = self._tail_of.get(U.data_ptr())
        if self.fold_copies and U is self.feat0:
            self._prep_copies.append((Z.data_ptr() + es * c_last, Z.shape[1], 0, CU))
        elif self.fold_copies and tail_idx is not None and tail_idx in self._body_args:
            ba = self._body_args[tail_idx]
            ba.out2, ba.out2_ld, ba.out2_n = Z.data_ptr() + es * c_last, Z.shape[1], CU
        elif self.fold_copies and tail_idx is not None:
            t_op = self.ops[tail_idx]
            t_op.p[7], t_op.f[2], t_op.f[3] = Z.data_ptr() + es * c_last, float(Z.shape[1]), float(CU)
        else:
            self._emit(make_op(OP_COPY_COLS, i=(B * 16, CU, U.shape[1], Z.shape[1], int(self.prec == 1), int(self.prec == 1)),
                                    p=(U.data_ptr(), Z.data_ptr() + es * c_last)))
        if self.fold_copies:
            self._prep_copies.append((Z.data_ptr() + es * (c_last + CU), Z.shape[1], 1, 3))
        else:
            self.xyz_copy_idx.append(len(self.ops))  # loop-invariant when the coordinates are a fixed condition
            self._emit(make_op(OP_COPY_COLS, i=(B * 16, 3, 3, Z.shape[1], 0, int(self.prec == 1)),
                                    p=(self.xyz.data_ptr(), Z.data_ptr() + es * (c_last + CU))))
        self._lane = 0
        if body is not None:
            self._emit_block_body(body, m1, 7, K, Z, self.cvec)
        else:
            self._mlp_tail(m1, 7, h1, self.cvec, r, mo, pair=pctx)
            finish(S)
        n1 = sd[m2 + ".first_mlp.0.weight"].shape[0]
        n2 = sd[m2 + ".res_connect.weight"].shape[0]
        hz, rz = self._buf(B * 16, n1), self._buf(B * 16, n2)
        f2, r2 = self._mlp_segments(m2, self.tvec, self.cvec, hz, rz)
        i0 = len(self.ops)
        self._gemm(Z, 4, [f2, r2])
        out = out_buf if out_buf is not None else self._buf(B * 16, n2)
        self._mlp_tail(m2, 4, hz, self.cvec, rz, out)
        if out_buf is not None:  # the last FP block: its second Mlp may join the output head in one launch (_point_chain)
            self._chain_front = dict(idx=list(range(i0, len(self.ops))), Z=Z, zin=zin, m2=m2, n1=n1, n2=n2)
        return out, n2

    def _remap_point_chain(self, remap):
        pc = getattr(self, "point_chain", None)
        if pc is not None:
            pc["idx"] = [remap[q] for q in pc["idx"]]
            if len(set(pc["idx"])) != 4 or pc["idx"] != list(range(pc["idx"][0], pc["idx"][0] + 4)):
                self.point_chain = None  # (its launches went into another merged launch)

    def _point_chain(self, dec0, head_i0, w0, w1, lay0):
        """round 5: the last FP block's second Mlp + the output head as ONE launch (SLIDE_OP_POINT_CHAIN, csrc/point_chain.hip) -- what
        the kernel needs, or None when the shapes are outside what it covers (fp16 plans; every width 128 with identity GroupNorm layout;
        per-timestep t-embedding table).  The samplers swap it in for the four per-point GEMM launches (diffusion.py); the plan itself
        keeps them (forward() of the engine, parity tests of the layers)."""
        sd, fr = self.sd, getattr(self, "_chain_front", None)
        if fr is None or self.prec != 1 or os.environ.get("SLIDE_POINT_CHAIN", "1") == "0":
            return None
        m2 = fr["m2"]
        ident = lambda c: (lambda l: np.array_equal(l[0], np.arange(c)) and l[1] == c and l[2] == c and l[3] == 4)(gn_layout(c))
        c2 = sd[m2 + ".second_mlp.0.weight"].shape[0]
        has_t, has_c = (m2 + ".fc.weight") in sd, (m2 + ".fc_condition.weight") in sd
        if not (fr["n1"] == 128 and fr["n2"] == 128 and c2 == 128 and (m2 + ".rest_mlp.0.weight") not in sd and ident(128)
                and len(fr["idx"]) == 2 and fr["idx"][1] + 1 == head_i0 and fr["Z"].shape[1] <= 192 and w0.shape[0] == 128
                and np.array_equal(lay0[0], np.arange(128)) and lay0[3] == 4 and lay0[2] == 128 and 128 < dec0.shape[1] <= 160
                and w1.shape[1] == 128 and self.out_dim <= 64 and sd["fc_lyaer.1.weight"].shape[0] == 128):
            return None
        A = self.A
        kz, zin = fr["Z"].shape[1], fr["zin"]
        Wz = np.zeros((256, kz), np.float32)
        Wz[:128, :zin] = self._w(m2 + ".first_mlp.0.weight")
        Wz[128:, :zin] = self._w(m2 + ".res_connect.weight")
        vz = np.stack([sd[m2 + ".first_mlp.0.bias"], sd[m2 + ".first_mlp.1.group_norm.weight"], sd[m2 + ".first_mlp.1.group_norm.bias"],
                       sd[m2 + ".res_connect.bias"]]).astype(np.float32)
        v2 = np.stack([sd[m2 + ".second_mlp.0.bias"], sd[m2 + ".second_mlp.1.group_norm.weight"],
                       sd[m2 + ".second_mlp.1.group_norm.bias"]]).astype(np.float32)
        n1c = ru(self.out_dim) // 32
        W0 = np.zeros((128, dec0.shape[1]), np.float32); W0[:, :w0.shape[1]] = w0
        W1 = np.zeros((n1c * 32, 128), np.float32); W1[:w1.shape[0]] = w1
        b1 = np.zeros(n1c * 32, np.float32); b1[:w1.shape[0]] = sd["fc_lyaer.3.bias"]
        v0 = np.stack([sd["fc_lyaer.0.bias"], sd["fc_lyaer.1.weight"], sd["fc_lyaer.1.bias"]]).astype(np.float32)
        off = lambda lst, pfx: sum(w for _, w in lst[:[p_ for p_, _ in lst].index(pfx)])
        W2 = self._w(m2 + ".second_mlp.0.weight")
        d = dict(idx=fr["idx"] + [head_i0, head_i0 + 1], Z=fr["Z"], kz=kz, X=dec0, k0=dec0.shape[1], n1c=n1c,
                 Wz=A.put(Wz, torch.float16), W2=A.put(W2, torch.float16),
                 W0=A.put(W0, torch.float16), W1=A.put(W1, torch.float16), vz=A.put(vz), v2=A.put(v2), v0=A.put(v0), b1=A.put(b1),
                 t_off=off(self._tvec, m2 + ".fc") if has_t else None,
                 c_off=off(self._cvec, m2 + ".fc_condition") if has_c else None)
        # round 6: the chain in the SPLIT arithmetic -- low fragments lo' = fp16((w - fp16(w)) 2^11) of the four
        # weight matrices.  These layers are the end of the network: their operand rounding reaches the prediction unattenuated
        # (tools/prec_select_feat.py), at 0.3 % of the FLOPs.  SLIDE_POINT_CHAIN_WIDE=0: fp16 operands as in round 5.
        if os.environ.get("SLIDE_POINT_CHAIN_WIDE", "1") != "0":
            def lo(w):
                w = np.asarray(w, np.float32)
                return ((w - w.astype(np.float16).astype(np.float32)) * np.float32(2048.0)).astype(np.float32)
            d.update(Wz_lo=A.put(lo(Wz), torch.float16), W2_lo=A.put(lo(W2), torch.float16), W0_lo=A.put(lo(W0), torch.float16),
                     W1_lo=A.put(lo(W1), torch.float16))
        return d

    def point_chain_args(self, eps_out=None):
        """SlidePointChainArgs of this plan's point chain (None when the plan has none): the caller keeps the returned block alive and
        puts its address into a SLIDE_OP_POINT_CHAIN op in place of the four launches `self.point_chain["idx"]`."""
        pch = getattr(self, "point_chain", None)
        if pch is None:
            return None
        c = SlidePointChainArgs()
        for k_ in ("Z", "Wz", "W2", "W0", "W1", "vz", "v2", "v0", "b1", "X", "Wz_lo", "W2_lo", "W0_lo", "W1_lo"):
            if k_ in pch:
                setattr(c, k_, pch[k_].data_ptr())
        c.eps = (eps_out if eps_out is not None else self.eps_pad).data_ptr()
        c.rows, c.z_ld, c.kz, c.x_ld, c.k0, c.n1c = self.B * 16, pch["Z"].shape[1], pch["kz"], pch["X"].shape[1], pch["k0"], pch["n1c"]
        c.eps_ld = self.eps_pad.shape[1]
        if pch["t_off"] is not None:
            c.tvec = self.tvec.data_ptr() + 4 * pch["t_off"]
            if self.per_sample_t:  # forward(): every sample's own t-embedding row
                c.t_idx, c.t_stride, c.t_bs = None, 0, self._t_bs
            else:                  # samplers: row t of the per-timestep table (the engine's t_dev[0])
                c.t_idx, c.t_stride, c.t_bs = self.t_dev.data_ptr(), self._n_fc, 0
        if pch["c_off"] is not None:
            c.cvec, c.c_bs = self.cvec.data_ptr() + 4 * pch["c_off"], self._c_bs
        return c

    # ------------------------------------------------------------------ whole network
    def _build(self):
        hp, sd, B, A = self.hp, self.sd, self.B, self.A
        arch = hp["architecture"]
        self.flops = 0
        self.gemm_flops = {}
        self.gemm_bytes = {}
        self.kernel_names = {}  # rocprofv3 kernel name of the round-3 ops (bench.py's roofline attribution), by op index
        self.xyz_copy_idx = []
        self.persistent = int(os.environ.get('SLIDE_PERSISTENT', '0'))  # 1: tile counter per XCD, 2: static tile lists
        # persistent I/O + per-step state
        self.x = A.zeros(B, 16, self.cx)
        self.ts = A.zeros(B)
        self.label = A.zeros(B, dtype=torch.int64)
        self.t_dev = A.zeros(8, dtype=torch.int32)  # [t, step counter, blocks-done counter of the update kernel, chain nonce, global index of the chain's first sample, -, -, -]
        self.xyz = A.zeros(B * 16, 3)
        C0 = self.cx  # in_fea_dim + 3 (position attached as feature)
        self.feat0 = self._buf(B * 16, C0)  # activation storage type
        self.kidx = A.zeros(B * 16, 16, dtype=torch.int32)
        self.kd2 = A.zeros(B * 16, 16)
        self.kw = A.zeros(B * 16, 16)  # group_knn's interpolation weights of the 8 nearest (pair decomposition)
        # t / condition vectors: widths are only known after the walk, so allocate generously and fix up below
        n_fc = sum(v.shape[0] for k, v in sd.items() if k.endswith(".fc.weight"))
        n_fcc = sum(v.shape[0] for k, v in sd.items() if k.endswith(".fc_condition.weight"))
        self._n_fc = n_fc
        self._t_bs = n_fc if self.per_sample_t else 0
        self._c_bs = n_fcc
        self.tvec = A.zeros(B if self.per_sample_t else self.t_table, n_fc)
        self.cvec = A.zeros(B, n_fcc)
        temb_slot = None
        if self.per_sample_t:
            temb_slot = len(self.ops)
            self.ops.append(None)  # TEMB placeholder (needs the .fc order of the walk)
        # (opt-in, SLIDE_CM_TABLES=1: chunk-major second copies of the per-point tables the first layers gather from, written
        # by prep_points / the attention tails -- measured neutral, 298-301 shapes/s either way: the gathers hit the vector cache)
        feat0_cm = None
        if self.use_cm and os.environ.get("SLIDE_CM_TABLES", "0") != "0":
            feat0_cm = self._cm_copy[self.feat0.data_ptr()] = self._buf(B * 16, C0)
        # COPY launches folded into the producers of their sources (SLIDE_FOLD_COPIES=0: separate launches)
        self.fold_copies = os.environ.get("SLIDE_FOLD_COPIES", "1") != "0"
        self._prep_copies, self._tail_of = [], {}
        self._body_args = {}
        self._prep_idx = len(self.ops)
        self._emit(make_op(OP_PREP_POINTS, i=(B, self.cx, self.feat0.shape[1], self.prec),
                                p=(self.x.data_ptr(), self.xyz.data_ptr(), self.feat0.data_ptr(), self.kidx.data_ptr(),
                                   self.kd2.data_ptr(), None if feat0_cm is None else feat0_cm.data_ptr(), None,
                                   self.kw.data_ptr())))
        feats, chans = [self.feat0], [C0]
        nsa, nfp = len(arch["npoint"]), len(arch["decoder_feature_dim"]) - 1
        # FP block j takes feats[nsa + j - nfp] as its skip / query input -- the table SA block nsa + j - nfp reads as well:
        # its query GEMM rides on that SA block's (SLIDE_MERGE_Q=0: one launch each)
        fp_q = {}
        for i in range(nsa):
            j = i + nfp - nsa
            extra = ()
            if 0 <= j < nfp and os.environ.get("SLIDE_MERGE_Q", "1") != "0":
                fp_q[j] = self._attention_query("FP_modules.%d.attention_module" % j, 8)
                extra = (fp_q[j],)
            o, c = self._sa_module(i, feats[i], chans[i], extra_q=extra)
            feats.append(o); chans.append(c)
        # per-level per-point outputs [B*16][ld] (diagnostics: tools/prec_probe.py compares them between precisions)
        self.levels = {"sa%d" % i: (feats[i + 1], chans[i + 1]) for i in range(nsa)}
        dec0 = None
        for i in range(-1, -(nfp + 1), -1):
            j = nfp + i
            out_buf = None
            if j == 0:  # the last FP module writes straight into the head's input [features | xyz]
                n2 = sd["FP_modules.0.mlp2.res_connect.weight"].shape[0]
                dec0 = self._buf(B * 16, n2 + 3)
                out_buf = dec0
            o, c = self._fp_module(j, feats[i - 1], chans[i - 1], feats[i], chans[i], out_buf, qctx=fp_q.get(j))
            feats[i - 1], chans[i - 1] = o, c
            self.levels["fp%d" % j] = (o, c)
        # output head fc_lyaer (pointnet2_with_pcld_condition.py:480-483): conv -> GN(32,128) -> ReLU -> conv
        c = chans[0]
        if self.fold_copies:
            self._prep_copies.append((dec0.data_ptr() + dec0.element_size() * c, dec0.shape[1], 1, 3))
        else:
            self.xyz_copy_idx.append(len(self.ops))
            self._emit(make_op(OP_COPY_COLS, i=(B * 16, 3, 3, dec0.shape[1], 0, int(self.prec == 1)),
                                    p=(self.xyz.data_ptr(), dec0.data_ptr() + dec0.element_size() * c)))
        hh = self._buf(B * 16, sd["fc_lyaer.0.weight"].shape[0])
        assert sd["fc_lyaer.0.weight"].shape[1] == c + 3
        head_i0 = len(self.ops)
        self._gemm(dec0, 4, [dict(w=self._w("fc_lyaer.0.weight"), bias=sd["fc_lyaer.0.bias"], mode=EPI_NORM,
                                  flags=F_POST_RELU, layout=gn_layout(sd["fc_lyaer.0.weight"].shape[0]), out=hh,
                                  gn=(sd["fc_lyaer.1.weight"], sd["fc_lyaer.1.bias"]))])
        self.eps_pad = self._buf(B * 16, self.out_dim, dtype=torch.float32)
        self._gemm(hh, 4, [dict(w=self._w("fc_lyaer.3.weight"), bias=sd["fc_lyaer.3.bias"], mode=EPI_RAW, out=self.eps_pad)])
        # what SLIDE_OP_HEAD_UPDATE needs (the samplers replace the two head GEMMs + their update launch by it, diffusion.py)
        self.head = None
        w0, w1 = self._w("fc_lyaer.0.weight"), self._w("fc_lyaer.3.weight")
        lay0 = gn_layout(w0.shape[0])
        if (self.prec == 1 and w0.shape[0] == 128 and np.array_equal(lay0[0], np.arange(128)) and lay0[3] == 4 and lay0[2] == 128
                and dec0.shape[1] <= 160 and w1.shape[1] == 128 and self.out_dim <= 64 and sd["fc_lyaer.1.weight"].shape[0] == 128):
            W0 = np.zeros((128, dec0.shape[1]), np.float32); W0[:, :w0.shape[1]] = w0
            n1c = ru(self.out_dim) // 32
            W1 = np.zeros((n1c * 32, 128), np.float32); W1[:w1.shape[0]] = w1
            b1 = np.zeros(n1c * 32, np.float32); b1[:w1.shape[0]] = sd["fc_lyaer.3.bias"]
            v0 = np.stack([sd["fc_lyaer.0.bias"], sd["fc_lyaer.1.weight"], sd["fc_lyaer.1.bias"]]).astype(np.float32)
            self.head = dict(idx=[head_i0, head_i0 + 1], X=dec0, k0=dec0.shape[1], n1c=n1c,
                             W0=self.A.put(W0, torch.float16), W1=self.A.put(W1, torch.float16), v0=self.A.put(v0), b1=self.A.put(b1))
        self.point_chain = self._point_chain(dec0, head_i0, w0, w1, lay0)
        self.eps = A.zeros(B, 16, self.out_dim)
        self.eps_copy_idx = len(self.ops)  # samplers read eps_pad directly and drop this op
        self._emit(make_op(OP_COPY_COLS, i=(B * 16, self.out_dim, self.eps_pad.shape[1], self.out_dim, 0, 0),
                                p=(self.eps_pad.data_ptr(), self.eps.data_ptr())))
        # t-embedding MLP + all .fc layers, class embedding + all .fc_condition layers (input-major weights)
        assert sum(w for _, w in self._tvec) == n_fc and sum(w for _, w in self._cvec) == n_fcc
        half = self.t_dim // 2
        cfreq = np.float32(np.log(10000) / (half - 1))
        freq = np.exp(np.arange(half, dtype=np.float32) * -cfreq).astype(np.float32)
        wfc = np.concatenate([sd[p + ".weight"] for p, _ in self._tvec], axis=0)      # (n_fc, 4*t_dim)
        bfc = np.concatenate([sd[p + ".bias"] for p, _ in self._tvec], axis=0)
        d = [A.put(a) for a in (sd["fc_t1.weight"].T, sd["fc_t1.bias"], sd["fc_t2.weight"].T, sd["fc_t2.bias"], wfc.T,
                                bfc, freq)]
        if self.per_sample_t:
            ts_src, nsamp = self.ts, B
        else:  # one-off table over every timestep (run once by `prepare()`), indexed by t_dev[0] in the epilogues
            ts_src, nsamp = A.put(np.arange(self.t_table, dtype=np.float32)), self.t_table
        temb = make_op(OP_TEMB, i=(nsamp, self.t_dim, n_fc),
                       p=(ts_src.data_ptr(), self.t_dev.data_ptr(), d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(),
                          d[3].data_ptr(), d[4].data_ptr(), d[5].data_ptr(), self.tvec.data_ptr(), d[6].data_ptr()))
        self.table_ops = None
        if self.per_sample_t:
            self.ops[temb_slot] = temb
        else:
            self.table_ops = (SlideOp * 1)(temb)
        wc = np.concatenate([sd[p + ".weight"] for p, _ in self._cvec], axis=0)
        bc = np.concatenate([sd[p + ".bias"] for p, _ in self._cvec], axis=0)
        dc = [A.put(a) for a in (sd["class_emb.weight"], wc.T, bc)]
        self.cond_op = make_op(OP_COND, i=(B, sd["class_emb.weight"].shape[1], n_fcc),
                               p=(self.label.data_ptr(), dc[0].data_ptr(), dc[1].data_ptr(), dc[2].data_ptr(),
                                  self.cvec.data_ptr()))
        if self._prep_copies:
            tab = (SlidePrepCopy * len(self._prep_copies))()
            for q, (dst, ld_, kind, n_) in enumerate(self._prep_copies):
                tab[q].dst, tab[q].ld, tab[q].kind, tab[q].n = dst, ld_, kind, n_
            self._prep_tab = A.put(np.frombuffer(bytes(tab), dtype=np.uint8).copy())
            prep = self.ops[self._prep_idx]
            prep.p[6], prep.i[4] = self._prep_tab.data_ptr(), len(self._prep_copies)
        self._merge_chains()
        self._merge_gx_pairs()
        self._merge_chain_query()
        self._merge_pp()
        self._w16 = {}
        # forward() runs the point chain too (round 6): what the samplers' step plans launch is what the golden forwards measure
        self.layer_ops = list(self.ops)  # (the four per-point GEMMs as launches of their own: parity tests of the layers)
        ops = self.layer_ops
        self._fwd_chain_args = self.point_chain_args()
        if self._fwd_chain_args is not None and all(self.ops[i] is not None and self.ops[i].kind == OP_GEMM for i in self.point_chain["idx"]):
            idx = self.point_chain["idx"]
            ops = [o if i != idx[0] else make_op(OP_POINT_CHAIN, p=(ctypes.addressof(self._fwd_chain_args),))
                   for i, o in enumerate(self.ops) if i not in idx[1:]]
        self.step_ops = (SlideOp * len(ops))(*ops)
        self.cond_ops = (SlideOp * 1)(self.cond_op)

    def _merge_chain_query(self):
        """SA blocks: [per-point query GEMM P | keys -> u GEMM (needs P) | fused Mlp chain (needs neither)] becomes
        [chain + query GEMM in ONE launch (SLIDE_OP_SA_CHAIN_P) | keys -> u GEMM]: the query GEMM's launch and gap disappear under
        the chain (SLIDE_CHAIN_P=0: three launches).  Op-index tables are re-keyed (the triple keeps its position)."""
        if os.environ.get("SLIDE_CHAIN_P", "1") == "0":
            return
        is_p = lambda o: (o is not None and o.kind == OP_GEMM and o.i[4] == 4 and o.i[6] == 1 and not (o.i[8] & 6) and o.i[9] != 3 and o.p[3] and o.p[6]
                          and not any(o.p[k] for k in (8, 9, 10, 11, 12, 13)))
        new_ops, remap, i = [], {}, 0
        self._chain_p_keep = []
        flops, nbytes, names = {}, {}, {}
        take = lambda src, dst, a_, b_: dst.__setitem__(b_, src[a_]) if a_ in src else None
        while i < len(self.ops):
            o0 = self.ops[i]
            o1 = self.ops[i + 1] if i + 1 < len(self.ops) else None
            o2 = self.ops[i + 2] if i + 2 < len(self.ops) else None
            k = len(new_ops)
            if (is_p(o0) and o1 is not None and o2 is not None and o1.kind == OP_GEMM_GX and o1.i[6] == 1 and o2.kind == OP_SA_CHAIN
                    and o0.i[0] == o2.i[0] * 16 and o0.i[10] == o1.i[10] == o2.i[10]):
                pair = (SlideOp * 2)(SlideOp.from_buffer_copy(bytes(o2)), SlideOp.from_buffer_copy(bytes(o0)))
                self._chain_p_keep.append(pair)
                op = make_op(OP_SA_CHAIN_P, i=(o2.i[0],), p=(ctypes.addressof(pair),))
                op.i[10] = o0.i[10]
                new_ops += [op, o1]
                remap[i], remap[i + 2], remap[i + 1] = k, k, k + 1
                flops[k] = self.gemm_flops.get(i, 0) + self.gemm_flops.get(i + 2, 0)
                nbytes[k] = tuple(self.gemm_bytes.get(i, (0, 0))[z] + self.gemm_bytes.get(i + 2, (0, 0))[z] for z in (0, 1))
                names[k] = "sa_chain_p_kernel<%d>" % (o2.i[3] // 32)
                for src, dst in ((self.gemm_flops, flops), (self.gemm_bytes, nbytes), (self.kernel_names, names)):
                    take(src, dst, i + 1, k + 1)
                i += 3
                continue
            new_ops.append(o0)
            remap[i] = k
            for src, dst in ((self.gemm_flops, flops), (self.gemm_bytes, nbytes), (self.kernel_names, names)):
                take(src, dst, i, k)
            i += 1
        self.ops = new_ops
        self.gemm_flops, self.gemm_bytes, self.kernel_names = flops, nbytes, names
        self.xyz_copy_idx = [remap[q] for q in self.xyz_copy_idx]
        self.eps_copy_idx = remap[self.eps_copy_idx]
        self._prep_idx = remap[self._prep_idx]
        self._remap_point_chain(remap)
        if self.head is not None:
            self.head["idx"] = [remap[q] for q in self.head["idx"]]
        self._body_args = {remap[q]: v for q, v in self._body_args.items()}
        self._tail_of = {k_: remap[v] for k_, v in self._tail_of.items()}

    def _merge_pp(self):
        """split plans (round 5): runs of consecutive per-point launches -- 16-row SLIDE_OP_GEMM in the split arithmetic and the float
        pair-table pass -- become ONE SLIDE_OP_PP_STAGE launch each (csrc/gemm_gxs.hip: one workgroup per sample walks the steps, the
        dense layers as exact fp32 FMA chains; 30 -> 17 launches per position step).  SLIDE_PP=0 keeps the launches apart.
        Op-index tables are re-keyed."""
        # OPT-IN (SLIDE_PP=1): measured SLOWER -- one workgroup per sample re-reads every layer's weights per sample and its K loop is a
        # chain of dependent L2 round trips on four waves: 55 - 131 us per stage launch against 27 - 45 us for the launches it replaces
        # (position chain alone 365 -> 652 us per step, bench 379 -> 227 shapes/s)
        if not self.use_gxs or os.environ.get("SLIDE_PP", "0") == "0":
            return
        B = self.B

        def step_of(o):
            if o is None:
                return None
            if o.kind == OP_GEMM and id(o) in self._w16 and o.i[0] == B * 16 and o.i[2] <= 192 and not o.i[10]:
                W = self._w16[id(o)][1]
                wt = self.A.put(np.ascontiguousarray(W.T), torch.float32)  # K-major [k_pad][n_cob*32]
                return [0, o.p[0], wt.data_ptr(), o.p[2] & ~1, o.p[3] or 0, o.p[4] or 0, o.i[1], o.i[2], o.i[3], o.i[5]] + [0] * 6
            if o.kind == OP_PAIR_NORM and o.i[3] == 2 and o.i[4] == 1 and o.i[0] == B and not o.i[10]:
                return [1] + [o.p[k] or 0 for k in range(13)] + [o.i[1], o.i[2]]
            return None
        new_ops, remap, i = [], {}, 0
        self._pp_keep = []
        flops, nbytes, names = {}, {}, {}
        while i < len(self.ops):
            run, j = [], i
            while j < len(self.ops) and len(run) < 8:
                st = step_of(self.ops[j])
                if st is None or (st[0] == 1 and sum(1 for r_ in run if r_[0] == 1) >= 2):
                    break
                run.append(st)
                j += 1
            k = len(new_ops)
            if len(run) >= 2:
                rec = np.array([len(run), B] + [v for st in run for v in st], np.int64)
                self._pp_keep.append(rec)
                op = make_op(OP_PP_STAGE, i=(B, len(run)), p=(rec.ctypes.data,))
                new_ops.append(op)
                for q in range(i, j):
                    remap[q] = k
                flops[k] = sum(self.gemm_flops.get(q, 0) for q in range(i, j))
                nbytes[k] = tuple(sum(self.gemm_bytes.get(q, (0, 0))[z] for q in range(i, j)) for z in (0, 1))
                names[k] = "pp_stage_kernel"
                i = j
                continue
            new_ops.append(self.ops[i])
            remap[i] = k
            for src, dst in ((self.gemm_flops, flops), (self.gemm_bytes, nbytes), (self.kernel_names, names)):
                if i in src:
                    dst[k] = src[i]
            i += 1
        self.ops = new_ops
        self.gemm_flops, self.gemm_bytes, self.kernel_names = flops, nbytes, names
        self.xyz_copy_idx = [remap[q] for q in self.xyz_copy_idx]
        self.eps_copy_idx = remap[self.eps_copy_idx]
        self._prep_idx = remap[self._prep_idx]
        self._remap_point_chain(remap)
        if self.head is not None:
            self.head["idx"] = [remap[q] for q in self.head["idx"]]
            if len(set(self.head["idx"])) != 2:
                self.head = None
        self._body_args = {remap[q]: v for q, v in self._body_args.items()}
        self._tail_of = {k_: remap[v] for k_, v in self._tail_of.items()}

    def _merge_gx_pairs(self):
        """the mode-1 (keys -> u) and mode-0 (first Mlp layer) generated-X GEMMs of an FP block are independent and adjacent in
        the plan: where both run 64-channel tiles at two workgroups per CU they become ONE SLIDE_OP_GEMM_GX_DUAL launch
        (SLIDE_GX_DUAL=0: two launches).  Op-index tables are re-keyed."""
        if os.environ.get("SLIDE_GX_DUAL", "1") == "0":
            return
        new_ops, remap, i = [], {}, 0
        self._dual_keep = []
        flops, nbytes, names = {}, {}, {}
        while i < len(self.ops):
            a_, b_ = self.ops[i], self.ops[i + 1] if i + 1 < len(self.ops) else None
            k = len(new_ops)
            if (a_ is not None and b_ is not None and a_.kind == OP_GEMM_GX and b_.kind == OP_GEMM_GX and a_.i[6] == 1 and b_.i[6] == 0
                    and ((a_.f[0] in (1.0, 2.0) and b_.f[0] == 2.0) or (a_.f[0] == 3.0 and b_.f[0] == 3.0))
                    and a_.i[0] == b_.i[0] and a_.i[4] == b_.i[4] and a_.i[10] == b_.i[10]):
                pair = (SlideOp * 2)(SlideOp.from_buffer_copy(bytes(a_)), SlideOp.from_buffer_copy(bytes(b_)))
                self._dual_keep.append(pair)
                op = make_op(OP_GEMM_GX_DUAL, i=(a_.i[0],), p=(ctypes.addressof(pair),))
                op.i[10] = a_.i[10]
                new_ops.append(op)
                remap[i] = remap[i + 1] = k
                flops[k] = self.gemm_flops.get(i, 0) + self.gemm_flops.get(i + 1, 0)
                nbytes[k] = tuple(self.gemm_bytes.get(i, (0, 0))[z] + self.gemm_bytes.get(i + 1, (0, 0))[z] for z in (0, 1))
                names[k] = ("gemm_gxs_dual_kernel<%d>" if a_.f[0] == 3.0 else "gemm_gx_dual_kernel<%d>") % a_.i[4]
                i += 2
                continue
            new_ops.append(a_)
            remap[i] = k
            for src, dst in ((self.gemm_flops, flops), (self.gemm_bytes, nbytes), (self.kernel_names, names)):
                if i in src:
                    dst[k] = src[i]
            i += 1
        self.ops = new_ops
        self.gemm_flops, self.gemm_bytes, self.kernel_names = flops, nbytes, names
        self.xyz_copy_idx = [remap[q] for q in self.xyz_copy_idx]
        self.eps_copy_idx = remap[self.eps_copy_idx]
        self._prep_idx = remap[self._prep_idx]
        self._remap_point_chain(remap)
        if self.head is not None:
            self.head["idx"] = [remap[q] for q in self.head["idx"]]
        self._body_args = {remap[q]: v for q, v in self._body_args.items()}
        self._tail_of = {k_: remap[v] for k_, v in self._tail_of.items()}

    def _merge_chains(self):
        """runs of consecutive per-point GEMM launches (16 rows per sample, fp16, plain input) become ONE SLIDE_OP_GEMM_CHAIN
        launch each (csrc/gemm_chain.hip): the FP blocks' second Mlp_plus_t_emb, the output head.  SLIDE_GEMM_CHAIN=0 keeps the
        launches apart.  Every op-index table of the plan is re-keyed."""
        # SLIDE_GEMM_CHAIN = largest chain in KB of weights (default 0: none).  OPT-IN -- measured in bench.py's arrangement (three
        # feature sub-batches + the position chain): 92 launches per step instead of 108, but 366 shapes/s with every chain
        # merged and 371 with the chains of <= 256 KB of weights against 374 without: a chain's 20-odd workgroups each stream
        # ALL its weights (660 KB for the feature net's FP1.mlp2) and pay every layer's descriptor -> operands -> epilogue ->
        # store round trips in sequence, where the wide-grid launches of the other chains fill the same CUs meanwhile
        max_kb = int(os.environ.get("SLIDE_GEMM_CHAIN", "0"))
        if self.prec != 1 or not self.use_glds or max_kb <= 0:
            return
        def eligible(o):
            return (o is not None and o.kind == OP_GEMM and o.i[4] == 4 and o.i[6] == 1 and o.i[8] == 1 and o.i[9] != 3
                    and not any(o.p[k] for k in (3, 4, 6, 7, 8, 9, 10, 11, 12, 13)) and o.i[2] <= 1024)
        new_ops, remap, i = [], {}, 0
        self._chain_keep = []
        flops, nbytes, names = {}, {}, {}
        while i < len(self.ops):
            j = i
            while (j < len(self.ops) and eligible(self.ops[j]) and self.ops[j].i[0] == self.ops[i].i[0]
                   and self.ops[j].i[10] == self.ops[i].i[10] and j - i < 6):
                j += 1
            k = len(new_ops)
            while j - i >= 2 and sum(o.i[2] * o.i[3] * 64 for o in self.ops[i:j]) > max_kb * 1024:
                j -= 1  # (drop layers from the end until the chain's weights fit)
            if j - i >= 2:
                tab = (SlideChainLayer * (j - i))()
                for q, o in enumerate(self.ops[i:j]):
                    tab[q].X, tab[q].W, tab[q].epi = o.p[0], o.p[1], o.p[2] & ~1  # (strip SLIDE_EPI_PACKED_VECS: this kernel reads the descriptors' pointers)
                    tab[q].x_ld, tab[q].k_pad, tab[q].n_cob = o.i[1], o.i[2], o.i[3]
                self._chain_keep.append(tab)
                op = make_op(OP_GEMM_CHAIN, i=(self.ops[i].i[0], j - i), p=(ctypes.addressof(tab),))
                op.i[10] = self.ops[i].i[10]
                new_ops.append(op)
                for q in range(i, j):
                    remap[q] = k
                flops[k] = sum(self.gemm_flops.get(q, 0) for q in range(i, j))
                nbytes[k] = tuple(sum(self.gemm_bytes.get(q, (0, 0))[z] for q in range(i, j)) for z in (0, 1))
                names[k] = "gemm_chain_kernel"
                i = j
                continue
            new_ops.append(self.ops[i])
            remap[i] = k
            for src, dst in ((self.gemm_flops, flops), (self.gemm_bytes, nbytes), (self.kernel_names, names)):
                if i in src:
                    dst[k] = src[i]
            i += 1
        self.ops = new_ops
        self.gemm_flops, self.gemm_bytes, self.kernel_names = flops, nbytes, names
        self.xyz_copy_idx = [remap[q] for q in self.xyz_copy_idx]
        self.eps_copy_idx = remap[self.eps_copy_idx]
        if self.head is not None:
            self.head["idx"] = [remap[q] for q in self.head["idx"]]
            if len(set(self.head["idx"])) != 2:
                self.head = None  # (the head GEMMs went into a chain launch)
        self._prep_idx = remap[self._prep_idx]
        self._remap_point_chain(remap)
        self._body_args = {remap[q]: v for q, v in self._body_args.items()}
        self._tail_of = {k_: remap[v] for k_, v in self._tail_of.items()}

    # ------------------------------------------------------------------ execution
    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(self, ops_array, n=None):
        if self.x.device.type != "cuda":
            raise SlideHipError("DenoiserEngine plans only run on a GPU; there is no CPU fallback")
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.x.device)
        check(lib().slide_run_ops2(ops_array, len(ops_array) if n is None else n, self._stream(),
                                   ctypes.c_void_p(self._side.cuda_stream)), "slide_run_ops2")

    def prepare(self):
        """sampler mode: fill the per-timestep t-embedding table (once)"""
        if self.table_ops is not None:
            self.run(self.table_ops)
            self.table_ops = None

    def set_label(self, label):
        self.label.copy_(torch.as_tensor(label).to(self.device, torch.int64).reshape(self.B))
        self.run(self.cond_ops)

    def forward(self, x, ts, label):
        """PointNet2CloudCondition.forward(pointcloud=x, ts=ts, label=label) -> (B,16,out_dim)."""
        assert self.per_sample_t, "engine was built for a shared device-side timestep"
        self.x.copy_(torch.as_tensor(x).to(self.device, torch.float32).reshape(self.x.shape))
        self.ts.copy_(torch.as_tensor(ts).to(self.device, torch.float32).reshape(self.B))
        self.set_label(label)
        self.run(self.step_ops)
        return self.eps.clone()
