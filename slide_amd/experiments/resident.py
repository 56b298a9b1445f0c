"""Host side of the LDS-resident denoiser kernel (include/experiments/slide_resident.h, csrc/experiments/resident.hip).

`ResidentPlan` turns a reference `pointnet_config` + reference-named state dict into the kernel's inputs for the network
family whose per-sample working set fits one compute unit's LDS (the shipped position-DDPM configs: 16 points, 'nn'
grouping, kNN feature propagation, channel widths <= 128):
  * a per-step PROGRAM of `ROp` records (one op = whole layers of PointNet2CloudCondition.forward,
    pointnet2/models/pointnet2_with_pcld_condition.py:286-489; block internals pointnet2_ops/pointnet2_modules.py:119-176,
    :222-292, :771-873, pointnet2_ops/attention.py:70-96),
  * the fp16 weights packed in MFMA A-fragment order (1 KB per wave load) and the fp32 epilogue vectors,
  * the LDS arena map (every activation of a sample is a row-major fp16 matrix in LDS).
`ResidentPositionSampler` is the drop-in for `diffusion.PositionSampler`: sampling() (pointnet2/util.py:197-259) with ALL
requested reverse steps in ONE kernel launch (the engine plan: 43 launches per step).

torch is used for device memory and streams only; there is no CPU fallback.
"""
import ctypes

import numpy as np
import torch

from .._lib import LIB_EXP_PATH, SlideHipError, _load, check


def lib():
    """the resident kernel only exists in the EXPERIMENTS build of the library"""
    return _load(LIB_EXP_PATH)

from ..engine import DenoiserEngine

R_PREP, R_ASSEMBLE, R_GEMM, R_FINALIZE, R_AFFINE, R_TAIL, R_ZFILL = range(1, 8)
RS_RAW, RS_NORM, RS_STATS = 0, 1, 2
RF_PRE_RELU, RF_POST_RELU, RF_OUT_F32 = 1, 2, 4
RO_BARRIER_BEFORE_STORE = 1
LDS_LIMIT = 160 * 1024


class RIn(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("gat_off", "gat_ld", "nks_gat", "x_off", "x_ld", "nks_x")]


class ROp(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("type", "rows_log2", "kshift", "n_strips", "parts", "strip0", "flags", "pad0")] + [
        ("a", RIn), ("b", RIn), ("p", ctypes.c_int32 * 8), ("f", ctypes.c_float * 4)]


class RStrip(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("mode", "flags", "gs", "n_valid", "n_store", "out_off", "out_ld", "out_col", "wfrag",
                                               "vec_off", "addvec_kind", "addvec_off", "preadd_off", "preadd_ld", "stats_off")] + [
        ("inv_count", ctypes.c_float)]


class RArgs(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("ops", "strips", "wpool", "vpool", "tvec", "cvec", "x", "eps_out", "t_dev", "c_eps",
                                                "sqrt_alpha", "sigma", "noise", "dbg", "timeline")] + [
        (n, ctypes.c_int32) for n in ("n_ops", "n_steps", "B", "cx", "out_dim", "per_sample_t", "tvec_ld", "cvec_ld", "xstate_off",
                                      "xyz_off", "knn_off", "kd2_off", "eps_off", "lds_bytes")] + [
        ("seed_lo", ctypes.c_uint32), ("seed_hi", ctypes.c_uint32)]


def pad16(n):
    return (n + 15) // 16 * 16


class Buf:
    """a row-major fp16 matrix in the LDS arena; ld = padded width + 8 halfs (row stride = odd multiple of 16 bytes)"""

    def __init__(self, off, rows, width, alloc_rows=None, dtype_bytes=2, ld=None):
        self.off, self.rows, self.width = off, rows, width
        self.kp = pad16(width)
        self.ld = self.kp + 8 if ld is None else ld
        self.nbytes = (alloc_rows or rows) * self.ld * dtype_bytes

    @staticmethod
    def size(rows, width):
        return rows * (pad16(width) + 8) * 2


class ResidentPlan:
    NP = 16

    def __init__(self, engine):
        lib()
        self.e = e = engine
        self.hp, self.sd = e.hp, e.sd
        hp = self.hp
        arch = hp["architecture"]
        if len(arch["npoint"]) != 2 or len(arch["decoder_feature_dim"]) != 3 or arch["K"] != 8:
            raise SlideHipError("resident kernel: 2 SA + 2 FP levels with K = 8 expected")
        self.cx, self.out_dim = e.cx, e.out_dim
        self.toff = self._offsets(e._tvec)
        self.coff = self._offsets(e._cvec)
        self.ops, self.strips, self.frags, self.vecs = [], [], [], []
        self.names = {}  # debug: buffer name -> Buf (last definition)
        self._top = 0
        self._build()

    @staticmethod
    def _offsets(lst):
        out, o = {}, 0
        for name, w in lst:
            out[name] = o
            o += w
        return out

    # ------------------------------------------------------------------ arena
    def _alloc(self, nbytes):
        off = self._top
        self._top = (off + nbytes + 15) // 16 * 16
        self._peak = max(getattr(self, "_peak", 0), self._top)
        return off

    def _buf(self, name, rows, width, alloc_rows=None, at=None):
        ar = alloc_rows or rows
        off = self._alloc(Buf.size(ar, width)) if at is None else at
        b = Buf(off, rows, width, ar)
        self.names[name] = b
        return b

    def _w(self, name):
        w = self.sd[name]
        return w.reshape(w.shape[0], -1)

    # ------------------------------------------------------------------ emitters
    def _emit(self, type_, rows_log2=0, kshift=0, n_strips=0, parts=1, strip0=0, flags=0, a=None, b=None, p=(), f=()):
        o = ROp()
        o.type, o.rows_log2, o.kshift, o.n_strips, o.parts, o.strip0, o.flags = type_, rows_log2, kshift, n_strips, parts, strip0, flags
        for dst, src in ((o.a, a), (o.b, b)):
            if src is not None:
                for k, v in src.items():
                    setattr(dst, k, int(v))
        for k, v in enumerate(p):
            o.p[k] = int(v)
        for k, v in enumerate(f):
            o.f[k] = float(v)
        self.ops.append(o)
        return o

    def _in(self, x=None, gat=None):
        """x: Buf read as rows (all its padded width); gat: (table Buf) gathered through the neighbour table"""
        if x is not None and x.rows == 16:
            # a 16-row tensor is read as a 32-row MFMA operand: rows 16..31 are whatever follows it in the arena (they only
            # feed accumulator rows that are never stored); the arena must merely extend that far
            self._reach = max(getattr(self, "_reach", 0), x.off + 32 * x.ld * 2)
        d = dict(gat_off=0, gat_ld=0, nks_gat=0, x_off=0, x_ld=0, nks_x=0)
        if gat is not None:
            d.update(gat_off=gat.off, gat_ld=gat.ld, nks_gat=gat.kp // 16)
        if x is not None:
            d.update(x_off=x.off, x_ld=x.ld, nks_x=x.kp // 16)
        return d

    @staticmethod
    def _kphys(inp):
        return 16 * (inp["nks_gat"] + inp["nks_x"])

    def _pack(self, wp):
        """wp: (32, Kphys) float32 -> Kphys/16 fragments [64 lanes][8] fp16 (A operand of v_mfma_f32_32x32x16_f16)"""
        K = wp.shape[1]
        lane = np.arange(64)
        for f in range(K // 16):
            cols = 16 * f + 8 * (lane[:, None] >> 5) + np.arange(8)[None, :]
            self.frags.append(wp[(lane & 31)[:, None], cols].astype(np.float16))

    def _gemm(self, rows_log2, kshift, ina, segs, inb=None, flags=0, tail=False):
        """segs: dicts  w (O, Ka_logical) + cols (physical column of each logical input channel)  bias  mode  flags  gn=(g, b)
        gs  addvec=(kind, off)  preadd=Buf  out=(Buf, col0)  stats_off  wb / cols_b / bias_b (phase B)  f32_out=(off, ld, n)"""
        rows = 1 << rows_log2
        ka, kb = self._kphys(ina), (self._kphys(inb) if inb is not None else 0)
        assert (ka + kb) // 16 <= 16, "too many K steps for one unit (MAXF)"
        strip0 = len(self.strips)
        n_strips = 0
        any_norm = 0
        for sg in segs:
            O = sg["w"].shape[0]
            ns = (O + 31) // 32
            mode = sg.get("mode", RS_RAW)
            if mode == RS_NORM:
                assert O % 32 == 0 and sg["gs"] in (1, 2, 4), (O, sg.get("gs"))
            any_norm |= int(mode != RS_RAW)
            wa = np.zeros((ns * 32, ka), np.float32)
            wa[:O][:, sg["cols"]] = sg["w"]
            wb = np.zeros((ns * 32, kb), np.float32)
            if sg.get("wb") is not None:
                wb[:O][:, sg["cols_b"]] = sg["wb"]
            for s in range(ns):
                st = RStrip()
                st.mode, st.flags, st.gs = mode, sg.get("flags", 0), sg.get("gs", 1)
                nv = min(32, O - 32 * s)
                st.n_valid = nv
                st.wfrag = len(self.frags)
                self._pack(wa[32 * s:32 * s + 32])
                if kb:
                    self._pack(wb[32 * s:32 * s + 32])
                vec = np.zeros((4, 32), np.float32)
                for k, key in enumerate(("bias", None, None, "bias_b")):
                    if key and sg.get(key) is not None:
                        vec[k, :nv] = sg[key][32 * s:32 * s + nv]
                if sg.get("gn") is not None:
                    vec[1, :nv] = sg["gn"][0][32 * s:32 * s + nv]
                    vec[2, :nv] = sg["gn"][1][32 * s:32 * s + nv]
                st.vec_off = 128 * len(self.vecs)
                self.vecs.append(vec)
                if sg.get("f32_out") is not None:
                    off, ld, n = sg["f32_out"]
                    st.flags |= RF_OUT_F32
                    st.out_off, st.out_ld, st.out_col, st.n_store = off, ld, 0, n
                else:
                    ob, col0 = sg["out"]
                    st.out_off, st.out_ld, st.out_col = ob.off, ob.ld, col0 + 32 * s
                    # every column below the padded width is written (pads = 0: they are K columns of the next layer)
                    st.n_store = max(0, min(32, pad16(O) - 32 * s))
                if sg.get("addvec") is not None:
                    st.addvec_kind, st.addvec_off = sg["addvec"][0], sg["addvec"][1] + 32 * s
                    assert st.addvec_off % 4 == 0
                st.preadd_off, st.preadd_ld = -1, 0
                if sg.get("preadd") is not None:
                    pb = sg["preadd"]
                    st.preadd_off, st.preadd_ld = pb.off + 2 * 32 * s, pb.ld
                st.stats_off = sg["stats_off"] + 32 * s * 8 if sg.get("stats_off") is not None else 0
                st.inv_count = 1.0 / (sg.get("gs", 1) * rows)
                self.strips.append(st)
            n_strips += ns
        if rows_log2 == 4:
            parts = 1
        else:
            rb = rows // 32
            parts = 1
            while n_strips * parts * 2 <= 4 and parts * 2 <= rb:
                parts *= 2
            while tail and rb // parts > 4:
                parts *= 2
            assert parts in (1, 2, 4) and rb % parts == 0
        assert n_strips * parts * 256 <= self.xch_bytes, "statistics exchange scratch too small"
        self._emit(R_TAIL if tail else R_GEMM, rows_log2, kshift, n_strips, parts, strip0, flags, ina, inb, p=(self.xch, any_norm))

    # ------------------------------------------------------------------ blocks
    def _gn(self, pfx):
        return self.sd[pfx + ".group_norm.weight"], self.sd[pfx + ".group_norm.bias"]

    @staticmethod
    def _gs(C):
        G = min(32, C)
        assert C % G == 0
        return C // G

    def _grouped_in(self, table, C, ncoord):
        """input descriptor + physical column map of the grouped input [feat(C) | coords(ncoord)] of an SA / FP block"""
        if C + ncoord <= 16:  # everything fits the assembled chunk
            return None, C, np.arange(C + ncoord)
        kp = table.kp
        return table, 0, np.concatenate([np.arange(C), kp + np.arange(ncoord)])

    def _attention_block(self, rows_log2, kshift, T_in, C, q_tab, Cq, mp, ap, fp_mode, out, base):
        """one SA / KnnFP grouping block up to the attention output.  T_in: neighbour feature table (C channels),
        q_tab: query-point feature table (Cq channels), mp / ap: state-dict prefixes of the Mlp and the attention module,
        out = (Buf, col0) receiving the (16, c_last) attention output.  Returns c_last."""
        sd = self.sd
        K = 1 << kshift
        rows = 16 * K
        ncoord = 11 if fp_mode else 9
        c1 = sd[mp + ".first_mlp.0.weight"].shape[0]
        c_last = sd[mp + ".res_connect.weight"].shape[0]
        C1 = sd[ap + ".feat_conv.weight"].shape[0]
        C2 = sd[ap + ".grouped_feat_conv.weight"].shape[0]
        inter = sd[ap + ".weight_conv.2.weight"].shape[0]
        cout = sd[ap + ".weight_conv.5.weight"].shape[0]
        assert sd[mp + ".first_mlp.0.weight"].shape[1] == C + ncoord and cout == c_last
        assert sd[ap + ".grouped_feat_conv.weight"].shape[1] == C + ncoord and sd[ap + ".feat_conv.weight"].shape[1] == Cq
        has_rest = (mp + ".rest_mlp.0.weight") in sd
        gat, nfc, cols = self._grouped_in(T_in, C, ncoord)
        # ---- arena: R1 = [gc | h1 | Tq | P] later overwritten by mo; R2 = Tk later overwritten by u
        self._top = base
        r1 = self._top
        gc = Buf(self._alloc(rows * 24 * 2), rows, 16, ld=24)
        self.names[mp + ".gc"] = gc
        h1 = self._buf(mp + ".h1", rows, c1)
        Tq = self._buf(ap + ".Tq", 16, C1)
        P = self._buf(ap + ".P", 16, inter)
        c_mid = sd[mp + ".second_mlp.0.weight"].shape[0]
        h2 = h1
        if has_rest and c_mid != c1:
            h2 = self._buf(mp + ".h2", rows, c_mid)
        r1_end = self._top
        mo = self._buf(mp + ".mo", rows, c_last, at=r1)
        self._top = max(r1_end, r1 + mo.nbytes)
        r2 = self._top
        Tk = self._buf(ap + ".Tk", rows, C2, at=r2)
        u = self._buf(ap + ".u", rows, inter, at=r2)
        self._top = r2 + max(Tk.nbytes, u.nbytes)
        self._peak = max(self._peak, self._top)
        # ---- ops
        self._emit(R_ASSEMBLE, rows_log2, kshift, p=(gc.off, gc.ld, int(fp_mode), T_in.off, T_in.ld, nfc))
        g_in = self._in(x=gc, gat=gat)
        first = dict(w=self._w(mp + ".first_mlp.0.weight"), cols=cols, bias=sd[mp + ".first_mlp.0.bias"], mode=RS_NORM,
                     flags=RF_POST_RELU, gs=self._gs(c1), gn=self._gn(mp + ".first_mlp.1"), out=(h1, 0))
        if (mp + ".fc.weight") in sd:
            first["addvec"] = (1, self.toff[mp + ".fc"])
        kseg = dict(w=self._w(ap + ".grouped_feat_conv.weight"), cols=cols, bias=sd[ap + ".grouped_feat_conv.bias"],
                    mode=RS_STATS, out=(Tk, 0), stats_off=self.kstat)
        self._gemm(rows_log2, kshift, g_in, [first, kseg])
        self._gemm(4, 0, self._in(x=q_tab), [dict(w=self._w(ap + ".feat_conv.weight"), cols=np.arange(Cq),
                                                  bias=sd[ap + ".feat_conv.bias"], mode=RS_STATS, out=(Tq, 0),
                                                  stats_off=self.qstat)])
        Ct = C1 + C2
        G = min(32, Ct)
        gsc = (Ct - Ct % G) // G
        gam, bet = self._gn(ap + ".weight_conv.1")
        nn = gam.shape[0]
        gvec = np.zeros((2, Ct), np.float32)
        gvec[0, :nn], gvec[1, :nn] = gam, bet
        voff = 128 * len(self.vecs)
        nrow = (2 * Ct + 127) // 128
        padv = np.zeros(nrow * 128, np.float32)
        padv[:2 * Ct] = gvec.reshape(-1)
        for r in range(nrow):
            self.vecs.append(padv[128 * r:128 * r + 128].reshape(4, 32))
        assert pad16(C1) <= 1023 and pad16(C2) <= 1023 and pad16(C1) * 8 <= self.aff_bytes and pad16(C2) * 8 <= self.aff_bytes
        self._emit(R_FINALIZE, p=(self.qstat, self.kstat, C1, C2, voff, self.qaff, self.kaff,
                                   pad16(C1) | (pad16(C2) << 10) | (1 << 20)), f=(1.0 / (gsc * rows), float(K)))
        self._emit(R_AFFINE, p=(Tq.off, Tq.ld, 16, Tq.kp, self.qaff))
        self._emit(R_AFFINE, p=(Tk.off, Tk.ld, rows, Tk.kp, self.kaff))
        w2 = self._w(ap + ".weight_conv.2.weight")
        self._gemm(4, 0, self._in(x=Tq), [dict(w=w2[:, :C1], cols=np.arange(C1), mode=RS_RAW, out=(P, 0))])
        self._gemm(rows_log2, kshift, self._in(x=Tk),
                   [dict(w=w2[:, C1:], cols=np.arange(C2), bias=sd[ap + ".weight_conv.2.bias"], mode=RS_NORM, flags=RF_PRE_RELU,
                         gs=self._gs(inter), gn=self._gn(ap + ".weight_conv.4"), preadd=P, out=(u, 0))],
                   flags=RO_BARRIER_BEFORE_STORE)
        # Mlp tail: second (+ class embedding) [-> rest] + res_connect over the grouped input as a second phase
        second = dict(w=self._w(mp + ".second_mlp.0.weight"), cols=np.arange(c1), bias=sd[mp + ".second_mlp.0.bias"],
                      mode=RS_NORM, flags=RF_POST_RELU, gs=self._gs(c_mid), gn=self._gn(mp + ".second_mlp.1"))
        if (mp + ".fc_condition.weight") in sd:
            second["addvec"] = (2, self.coff[mp + ".fc_condition"])
        res = dict(wb=self._w(mp + ".res_connect.weight"), cols_b=cols, bias_b=sd[mp + ".res_connect.bias"])
        if has_rest:
            second["out"] = (h2, 0)
            self._gemm(rows_log2, kshift, self._in(x=h1), [second], flags=RO_BARRIER_BEFORE_STORE)
            c3 = sd[mp + ".rest_mlp.0.weight"].shape[0]
            rest = dict(w=self._w(mp + ".rest_mlp.0.weight"), cols=np.arange(c_mid), bias=sd[mp + ".rest_mlp.0.bias"], mode=RS_NORM,
                        flags=RF_POST_RELU, gs=self._gs(c3), gn=self._gn(mp + ".rest_mlp.1"), out=(mo, 0), **res)
            self._gemm(rows_log2, kshift, self._in(x=h2), [rest], inb=g_in, flags=RO_BARRIER_BEFORE_STORE)
        else:
            second.update(out=(mo, 0), **res)
            self._gemm(rows_log2, kshift, self._in(x=h1), [second], inb=g_in, flags=RO_BARRIER_BEFORE_STORE)
        # attention tail
        tail = dict(w=self._w(ap + ".weight_conv.5.weight"), cols=np.arange(inter), bias=sd[ap + ".weight_conv.5.bias"],
                    wb=self._w(ap + ".feat_out_conv.0.weight"), cols_b=np.arange(c_last), bias_b=sd[ap + ".feat_out_conv.0.bias"],
                    mode=RS_NORM, gs=self._gs(cout), gn=self._gn(ap + ".feat_out_conv.1"), out=out)
        self._gemm(rows_log2, kshift, self._in(x=u), [tail], inb=self._in(x=mo), tail=True)
        return c_last

    def _mlp16(self, pfx, Z, zin, out, base_after):
        """Mlp_plus_t_emb on 16 rows (the mlp2 of a KnnFP module): first (+t) -> second (+cond) + res_connect(Z)"""
        sd = self.sd
        n1 = sd[pfx + ".first_mlp.0.weight"].shape[0]
        n2 = sd[pfx + ".second_mlp.0.weight"].shape[0]
        assert (pfx + ".rest_mlp.0.weight") not in sd and sd[pfx + ".first_mlp.0.weight"].shape[1] == zin
        self._top = base_after
        hz = self._buf(pfx + ".hz", 16, n1)
        self._peak = max(self._peak, self._top)
        first = dict(w=self._w(pfx + ".first_mlp.0.weight"), cols=np.arange(zin), bias=sd[pfx + ".first_mlp.0.bias"], mode=RS_NORM,
                     flags=RF_POST_RELU, gs=self._gs(n1), gn=self._gn(pfx + ".first_mlp.1"), out=(hz, 0))
        if (pfx + ".fc.weight") in sd:
            first["addvec"] = (1, self.toff[pfx + ".fc"])
        self._gemm(4, 0, self._in(x=Z), [first])
        second = dict(w=self._w(pfx + ".second_mlp.0.weight"), cols=np.arange(n1), bias=sd[pfx + ".second_mlp.0.bias"],
                      mode=RS_NORM, flags=RF_POST_RELU, gs=self._gs(n2), gn=self._gn(pfx + ".second_mlp.1"), out=out,
                      wb=self._w(pfx + ".res_connect.weight"), cols_b=np.arange(zin), bias_b=sd[pfx + ".res_connect.bias"])
        if (pfx + ".fc_condition.weight") in sd:
            second["addvec"] = (2, self.coff[pfx + ".fc_condition"])
        self._gemm(4, 0, self._in(x=hz), [second], inb=self._in(x=Z))
        return n2

    # ------------------------------------------------------------------ whole network
    def _build(self):
        sd, cx = self.sd, self.cx
        C0 = cx
        self._peak = 0
        # fixed small areas
        self.xstate = self._alloc(16 * cx * 4)
        self.xyz = self._alloc(16 * 3 * 4)
        self.knn = self._alloc(256)
        self.kd2 = self._alloc(16 * 16 * 4)
        self.eps = self._alloc(16 * 4 * 4)
        self.xch_bytes = 2048
        self.xch = self._alloc(self.xch_bytes)
        self.aff_bytes = 160 * 8
        self.qstat, self.kstat = self._alloc(self.aff_bytes), self._alloc(self.aff_bytes)
        self.qaff, self.kaff = self._alloc(self.aff_bytes), self._alloc(self.aff_bytes)
        # per-level point-feature tables (32 rows allocated: they are read as 32-row MFMA operands)
        c_sa = [sd["SA_modules.%d.mlps.0.res_connect.weight" % i].shape[0] for i in range(2)]
        T = [self._buf("feat0", 16, C0), self._buf("feat1", 16, c_sa[0]),
             self._buf("feat2", 16, c_sa[1])]
        n_fp1 = sd["FP_modules.1.mlp2.second_mlp.0.weight"].shape[0]
        n_fp0 = sd["FP_modules.0.mlp2.second_mlp.0.weight"].shape[0]
        T1b = self._buf("feat1b", 16, n_fp1)
        headin = self._buf("headin", 16, n_fp0 + 3)
        base = self._top
        self._emit(R_PREP, p=(T[0].off, T[0].ld, T[0].kp))
        chans = [C0] + c_sa
        for i in range(2):
            pfx = "SA_modules.%d" % i
            c = self._attention_block(8, 4, T[i], chans[i], T[i], chans[i], pfx + ".mlps.0", pfx + ".attention_modules.0", False,
                                      (T[i + 1], 0), base)
            assert c == chans[i + 1]
        # FP1: unknown = level 1 (feat1), known = level 2 (feat2);  FP0: unknown = level 0 (feat0), known = level 1 (FP1 output)
        for j, U, CU, Kf, CK, out in ((1, T[1], chans[1], T[2], chans[2], (T1b, 0)), (0, T[0], chans[0], T1b, n_fp1, (headin, 0))):
            pfx = "FP_modules.%d" % j
            c_last = sd[pfx + ".mlp1.res_connect.weight"].shape[0]
            zin = c_last + CU + 3
            self._top = base
            Z = self._buf(pfx + ".Z", 16, zin)
            after_z = self._top
            self._attention_block(7, 3, Kf, CK, U, CU, pfx + ".mlp1", pfx + ".attention_module", True, (Z, 0), after_z)
            blk_end = self._top
            self._emit(R_ZFILL, p=(Z.off, Z.ld, c_last, U.off, U.ld, CU, Z.kp))
            self._mlp16(pfx + ".mlp2", Z, zin, out, blk_end)
        # head fc_lyaer (pointnet2_with_pcld_condition.py:480-483): conv -> GN(32, 128) -> ReLU -> conv
        self._emit(R_ZFILL, p=(headin.off, headin.ld, n_fp0, -1, 0, 0, headin.kp))
        ch = sd["fc_lyaer.0.weight"].shape[0]
        self._top = base
        hh = self._buf("hh", 16, ch)
        self._peak = max(self._peak, self._top)
        assert sd["fc_lyaer.0.weight"].shape[1] == n_fp0 + 3
        self._gemm(4, 0, self._in(x=headin), [dict(w=self._w("fc_lyaer.0.weight"), cols=np.arange(n_fp0 + 3), bias=sd["fc_lyaer.0.bias"],
                                                    mode=RS_NORM, flags=RF_POST_RELU, gs=self._gs(ch),
                                                    gn=(sd["fc_lyaer.1.weight"], sd["fc_lyaer.1.bias"]), out=(hh, 0))])
        assert self.out_dim <= 4
        self._gemm(4, 0, self._in(x=hh), [dict(w=self._w("fc_lyaer.3.weight"), cols=np.arange(ch), bias=sd["fc_lyaer.3.bias"],
                                                mode=RS_RAW, f32_out=(self.eps, 4, self.out_dim))])
        self.lds_bytes = (max(self._peak, getattr(self, "_reach", 0)) + 15) // 16 * 16
        if self.lds_bytes > LDS_LIMIT:
            raise SlideHipError("resident kernel: the network's working set (%d bytes) does not fit one CU's LDS" % self.lds_bytes)
        # ---- device copies
        A = self.e.A
        self.d_ops = A.put(np.frombuffer(b"".join(bytes(o) for o in self.ops), dtype=np.uint8).copy())
        self.d_strips = A.put(np.frombuffer(b"".join(bytes(s) for s in self.strips), dtype=np.uint8).copy())
        self.d_w = A.put(np.stack(self.frags).reshape(-1), torch.float16)
        self.d_v = A.put(np.stack(self.vecs).reshape(-1))
        self.weight_bytes = 2 * self.d_w.numel()

    def args(self, n_steps, x, eps_out=None, t_dev=None, tabs=None, noise=None, seed=0, per_sample_t=False, dbg=None, timeline=None):
        e = self.e
        a = RArgs()
        a.ops, a.strips, a.wpool, a.vpool = self.d_ops.data_ptr(), self.d_strips.data_ptr(), self.d_w.data_ptr(), self.d_v.data_ptr()
        a.tvec, a.cvec, a.x = e.tvec.data_ptr(), e.cvec.data_ptr(), x.data_ptr()
        a.eps_out = None if eps_out is None else eps_out.data_ptr()
        a.t_dev = None if t_dev is None else t_dev.data_ptr()
        if tabs is not None:
            a.c_eps, a.sqrt_alpha, a.sigma = (t.data_ptr() for t in tabs)
        a.noise = None if noise is None else noise.data_ptr()
        a.dbg = None if dbg is None else dbg.data_ptr()
        a.timeline = None if timeline is None else timeline.data_ptr()
        a.n_ops, a.n_steps, a.B, a.cx, a.out_dim = len(self.ops), int(n_steps), e.B, self.cx, self.out_dim
        a.per_sample_t, a.tvec_ld, a.cvec_ld = int(per_sample_t), e.tvec.shape[1], e.cvec.shape[1]
        a.xstate_off, a.xyz_off, a.knn_off, a.kd2_off, a.eps_off, a.lds_bytes = self.xstate, self.xyz, self.knn, self.kd2, self.eps, self.lds_bytes
        a.seed_lo, a.seed_hi = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
        return a


def _run(a, stream):
    check(lib().slide_resident_run(ctypes.byref(a), ctypes.c_void_p(stream.cuda_stream)), "slide_resident_run")


class ResidentDenoiser:
    """forward(x, ts, label) API on the resident kernel (tests / module-level use): the t-embedding and class-embedding
    vectors come from the engine's own kernels (per-sample timesteps), the network runs in one launch."""

    def __init__(self, hp, state_dict, batch, device):
        self.engine = DenoiserEngine(hp, state_dict, batch, device, prec="fp16", per_sample_t=True)
        self.plan = ResidentPlan(self.engine)
        self.eps = torch.zeros(batch, 16, self.engine.out_dim, device=device)

    def forward(self, x, ts, label, dbg=False):
        e = self.engine
        if e.x.device.type != "cuda":
            raise SlideHipError("the resident denoiser only runs on a GPU; there is no CPU fallback")
        e.x.copy_(torch.as_tensor(x).to(e.device, torch.float32).reshape(e.x.shape))
        e.ts.copy_(torch.as_tensor(ts).to(e.device, torch.float32).reshape(e.B))
        e.set_label(label)
        from ..engine import OP_TEMB, SlideOp
        temb = [o for o in e.ops if o.kind == OP_TEMB]
        e.run((SlideOp * 1)(temb[0]))
        d = torch.zeros(e.B, self.plan.lds_bytes, dtype=torch.uint8, device=e.device) if dbg else None
        _run(self.plan.args(1, e.x, eps_out=self.eps, per_sample_t=True, dbg=d), torch.cuda.current_stream())
        return (self.eps.clone(), d) if dbg else self.eps.clone()


class ResidentPositionSampler:
    """sampling(net, (B,16,3), diffusion_hyperparams, label=...) -- pointnet2/util.py:197-259 -- with every requested reverse
    step inside ONE launch of the LDS-resident kernel.  Same interface and the same in-kernel noise stream as
    diffusion.PositionSampler (seed, chain nonce, step, element)."""

    def __init__(self, hp, state_dict, batch, device, diffusion_config, noise=None, seed=0):
        from ..diffusion import F32, calc_diffusion_hyperparams
        self.engine = e = DenoiserEngine(hp, state_dict, batch, device, prec="fp16", per_sample_t=False, t_table=diffusion_config["T"])
        self.plan = ResidentPlan(e)
        self.B, self.device, self.seed = int(batch), device, int(seed)
        self.stream = torch.cuda.Stream(device=device)
        dh = calc_diffusion_hyperparams(**diffusion_config)
        self.dh, self.T = dh, dh["T"]
        c_eps = (F32(1) - dh["Alpha"]) / np.sqrt(F32(1) - dh["Alpha_bar"]).astype(F32)
        self.tabs = [e.A.put(a.astype(F32)) for a in (c_eps, np.sqrt(dh["Alpha"]), dh["Sigma"])]
        self.noise = None if noise is None else e.A.put(np.asarray(noise, F32).reshape(len(noise), -1))
        self.n_launches = 1
        self._t_init = {}
        if device.type == "cuda":
            e.prepare()
            torch.cuda.synchronize(device)

    def begin(self, label, x_T, t_start=None):
        e = self.engine
        t_start = self.T - 1 if t_start is None else int(t_start)
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        for t_ in (label, x_T):  # (the allocator must not recycle a caller-stream input before the copies below have run)
            if torch.is_tensor(t_) and t_.is_cuda:
                t_.record_stream(self.stream)
        with torch.cuda.stream(self.stream):
            e.set_label(label)
            e.x.copy_(torch.as_tensor(x_T).to(self.device, torch.float32).reshape(e.x.shape))
            t0 = self._t_init.get(t_start)
            if t0 is None:
                t0 = self._t_init[t_start] = torch.tensor([t_start, 0, 0], dtype=torch.int32).to(self.device)
            e.t_dev[:3].copy_(t0)
            e.t_dev[3:4].add_(1)  # chain nonce

    def advance(self, n_steps):
        if n_steps <= 0:
            return
        with torch.cuda.stream(self.stream):
            _run(self.plan.args(n_steps, self.engine.x, t_dev=self.engine.t_dev, tabs=self.tabs, noise=self.noise, seed=self.seed),
                 self.stream)

    def state(self):
        with torch.cuda.stream(self.stream):
            out = self.engine.x.clone()
        self.stream.synchronize()
        return out

    def sample(self, label, x_T, t_start=None, n_steps=None):
        t_start = self.T - 1 if t_start is None else t_start
        n_steps = t_start + 1 if n_steps is None else n_steps
        self.begin(label, x_T, t_start)
        self.advance(n_steps)
        return self.state()
