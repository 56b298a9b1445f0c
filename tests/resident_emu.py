"""TEST INFRASTRUCTURE: a numpy interpreter of the LDS-resident kernel's op program (include/experiments/slide_resident.h).

It executes the `ROp` / `RStrip` records and the packed weight / vector pools that `slide_amd.resident.ResidentPlan` hands
to `resident_kernel` with the kernel's own data layout (a byte arena standing for one workgroup's LDS, fp16 activations,
fp32 accumulation and statistics), so the CPU tests can check the host-side plan -- op order, arena aliasing, weight
packing, column maps -- against the oracle without a GPU.  The arena starts as fp16 NaNs: any read of a column / row the
program never wrote shows up as a NaN in the output.
"""
import numpy as np

from slide_amd.experiments import resident as R

F32 = np.float32


class Emu:
    def __init__(self, plan):
        self.p = plan
        self.frags = np.stack(plan.frags).astype(np.float16)  # [n][64][8]
        self.vecs = np.stack(plan.vecs).reshape(-1).astype(F32)

    # ---- arena views
    def h(self, off, ld, rows, cols):
        v = self.lds[off:off + rows * ld * 2].view(np.float16).reshape(rows, ld)
        return v[:, :cols]

    def f(self, off, n):
        return self.lds[off:off + 4 * n].view(F32)

    def wstrip(self, wfrag, f0, nf):
        """(32, 16 nf) weight block of fragments wfrag + f0 .. + f0 + nf"""
        lane = np.arange(64)
        W = np.zeros((32, 16 * nf), F32)
        for f in range(nf):
            fr = self.frags[wfrag + f0 + f].astype(F32)
            for i in range(8):
                W[lane & 31, 16 * f + 8 * (lane >> 5) + i] = fr[lane, i]
        return W

    def xin(self, op, inp, rows):
        """the (rows, K) fp16 operand matrix of one accumulation phase, as float32"""
        parts = []
        if inp.nks_gat:
            K = 1 << op.kshift
            r = np.arange(rows)
            nb = self.lds[self.p.knn:self.p.knn + 256].reshape(16, 16)[(r >> op.kshift) & 15, r & (K - 1)]
            tab = self.h(inp.gat_off, inp.gat_ld, 16, 16 * inp.nks_gat)
            parts.append(tab[nb])
        if inp.nks_x:
            parts.append(self.h(inp.x_off, inp.x_ld, rows, 16 * inp.nks_x))
        return np.concatenate(parts, axis=1).astype(F32)

    @staticmethod
    def gn(y, gs, inv_count, gamma, beta):
        """y (rows, 32) one strip; groups of gs consecutive channels"""
        S = y.sum(0, dtype=F32).reshape(-1, gs).sum(1)
        Q = (y * y).sum(0, dtype=F32).reshape(-1, gs).sum(1)
        mean = S * F32(inv_count)
        var = np.maximum(Q * F32(inv_count) - mean * mean, 0)
        rstd = F32(1) / np.sqrt(var + F32(1e-5))
        sc = gamma * np.repeat(rstd, gs)
        sh = beta - np.repeat(mean, gs) * sc
        return (y * sc + sh).astype(F32)

    def store(self, st, y, rows):
        if st.flags & R.RF_OUT_F32:
            o = self.f(st.out_off, rows * st.out_ld).reshape(rows, st.out_ld)
            o[:, st.out_col:st.out_col + st.n_store] = y[:, :st.n_store]
        elif st.n_store > 0:
            o = self.h(st.out_off, st.out_ld, rows, st.out_col + st.n_store)
            o[:, st.out_col:] = y[:, :st.n_store].astype(np.float16)

    def gemm(self, op, trow, crow, tail):
        rows = 1 << op.rows_log2
        nfa, nfb = op.a.nks_gat + op.a.nks_x, op.b.nks_gat + op.b.nks_x
        xa = self.xin(op, op.a, rows)
        xb = self.xin(op, op.b, rows) if nfb else None
        outs = []
        for s in range(op.n_strips):
            st = self.p.strips[op.strip0 + s]
            v = self.vecs[st.vec_off:st.vec_off + 128].reshape(4, 32)
            y = xa @ self.wstrip(st.wfrag, 0, nfa).T + v[0]
            if tail:
                val = xb @ self.wstrip(st.wfrag, nfa, nfb).T + v[3]
                val = np.maximum(self.gn(val, st.gs, st.inv_count, v[1], v[2]), 0)
                K = 1 << op.kshift
                sc = y.reshape(16, K, 32)
                e = np.exp(sc - sc.max(1, keepdims=True))
                o = (e * val.reshape(16, K, 32)).sum(1) / e.sum(1)
                outs.append((st, o.astype(F32), 16))
                continue
            if st.preadd_off >= 0:
                P = self.h(st.preadd_off, st.preadd_ld, 16, 32).astype(F32)
                y = y + P[(np.arange(rows) >> op.kshift) & 15]
            if st.mode == R.RS_STATS:
                r = np.maximum(y, 0)
                d = self.f(st.stats_off, 64).reshape(32, 2)
                d[:, 0], d[:, 1] = r.sum(0, dtype=F32), (r * r).sum(0, dtype=F32)
            else:
                if st.flags & R.RF_PRE_RELU:
                    y = np.maximum(y, 0)
                if st.mode == R.RS_NORM:
                    y = self.gn(y, st.gs, st.inv_count, v[1], v[2])
                if st.flags & R.RF_POST_RELU:
                    y = np.maximum(y, 0)
            if st.addvec_kind:
                y = y + (trow if st.addvec_kind == 1 else crow)[st.addvec_off:st.addvec_off + 32]
            if nfb:
                y = y + xb @ self.wstrip(st.wfrag, nfa, nfb).T + v[3]
            outs.append((st, y.astype(F32), rows))
        for st, y, r in outs:  # all strips read their inputs before any output lands (outputs may alias inputs)
            self.store(st, y, r)

    def run(self, x, trow, crow, n_ops=None):
        """one forward for one sample: x (16, cx) -> eps (16, out_dim); n_ops: stop after that many ops (debugging)"""
        p = self.p
        self.lds = np.full(p.lds_bytes, 0, np.uint8)
        self.lds.view(np.uint16)[:] = 0x7E00  # fp16 NaN (and an fp32 NaN pattern, 0x7E007E00)
        cx = p.cx
        self.f(p.xstate, 16 * cx)[:] = x.reshape(-1)
        for op in p.ops[:n_ops]:
            q = list(op.p)
            if op.type == R.R_PREP:
                xs = self.f(p.xstate, 16 * cx).reshape(16, cx)
                xyz = xs[:, :3].copy()
                self.f(p.xyz, 48)[:] = xyz.reshape(-1)
                ft = self.h(q[0], q[1], 16, q[2])
                ft[:] = 0
                ft[:, :cx - 3] = xs[:, 3:]
                ft[:, cx - 3:cx] = xyz
                from oracle import ops as oracle_ops  # the C restatement: sorted squared distances, ties -> lower index
                d2, idx = oracle_ops.knn_points(xyz[None], xyz[None], 16)
                order, dsort = idx[0], d2[0].astype(F32)
                self.lds[p.knn:p.knn + 256] = order.astype(np.uint8).reshape(-1)
                self.f(p.kd2, 256)[:] = dsort.reshape(-1)
            elif op.type == R.R_ASSEMBLE:
                K = 1 << op.kshift
                xyz = self.f(p.xyz, 48).reshape(16, 3)
                knn = self.lds[p.knn:p.knn + 256].reshape(16, 16)
                kd2 = self.f(p.kd2, 256).reshape(16, 16)
                ft = self.h(q[3], q[4], 16, max(q[5], 1)).astype(F32)
                out = self.h(q[0], q[1], 16 * K, 16)
                for r in range(16 * K):
                    pt, k = r >> op.kshift, r & (K - 1)
                    nb = knn[pt, k]
                    v = list(ft[nb, :q[5]])
                    if q[2]:
                        rec = F32(1) / (kd2[pt, :K] + F32(1e-8))
                        v += [kd2[pt, k], rec[k] / rec.sum(dtype=F32)] + list(xyz[nb]) + list(xyz[nb] - xyz[pt]) + list(xyz[pt])
                    else:
                        v += list(xyz[nb] - xyz[pt]) + list(xyz[nb]) + list(xyz[pt])
                    out[r] = np.array(v + [0] * (16 - len(v)), F32).astype(np.float16)
            elif op.type == R.R_GEMM:
                self.gemm(op, trow, crow, False)
            elif op.type == R.R_TAIL:
                self.gemm(op, trow, crow, True)
            elif op.type == R.R_FINALIZE:
                C1, C2 = q[2], q[3]
                C1p, C2p = q[7] & 1023, (q[7] >> 10) & 1023
                Ct = C1 + C2
                G = min(32, Ct)
                nn = Ct - Ct % G
                gs = nn // G
                qs = self.f(q[0], 2 * C1p).reshape(C1p, 2)[:C1] * F32(op.f[1])
                ks = self.f(q[1], 2 * C2p).reshape(C2p, 2)[:C2]
                st = np.concatenate([qs, ks], 0)
                gam, bet = self.vecs[q[4]:q[4] + Ct], self.vecs[q[4] + Ct:q[4] + 2 * Ct]
                sc, sh = np.ones(Ct, F32), np.zeros(Ct, F32)
                S = st[:nn, 0].reshape(G, gs).sum(1)
                Q = st[:nn, 1].reshape(G, gs).sum(1)
                mean = S * F32(op.f[0])
                var = np.maximum(Q * F32(op.f[0]) - mean * mean, 0)
                rstd = F32(1) / np.sqrt(var + F32(1e-5))
                sc[:nn] = gam[:nn] * np.repeat(rstd, gs)
                sh[:nn] = bet[:nn] - np.repeat(mean, gs) * sc[:nn]
                qo = self.f(q[5], 2 * C1p).reshape(C1p, 2)
                ko = self.f(q[6], 2 * C2p).reshape(C2p, 2)
                qo[:] = 0; ko[:] = 0
                qo[:C1, 0], qo[:C1, 1] = sc[:C1], sh[:C1]
                ko[:C2, 0], ko[:C2, 1] = sc[C1:], sh[C1:]
            elif op.type == R.R_AFFINE:
                x_ = self.h(q[0], q[1], q[2], q[3])
                ss = self.f(q[4], 2 * q[3]).reshape(q[3], 2)
                x_[:] = (np.maximum(x_.astype(F32), 0) * ss[:, 0] + ss[:, 1]).astype(np.float16)
            elif op.type == R.R_ZFILL:
                xyz = self.f(p.xyz, 48).reshape(16, 3)
                dst = self.h(q[0], q[1], 16, q[6])
                n = q[5]
                if n:
                    dst[:, q[2]:q[2] + n] = self.h(q[3], q[4], 16, n)
                dst[:, q[2] + n:q[2] + n + 3] = xyz.astype(np.float16)
                dst[:, q[2] + n + 3:] = 0
            else:
                raise ValueError(op.type)
        return self.f(p.eps, 64).reshape(16, 4)[:, :p.out_dim].copy()
