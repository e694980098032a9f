"""numpy interpreter of an eabnet_amd.program.Program  (TEST INFRASTRUCTURE).

Executes the SAME op list, packed-weight arena and workspace plan that
libeabnet_hip.so receives, following the semantics documented in
include/eabnet_hip.h, so the host-side lowering (weight packing, taps/phases of
the transposed convolutions, channel permutations, schedule, buffer plan) can
be verified against the oracle on a machine without a GPU.  It deliberately
mirrors the kernels' tiling of the InstanceNorm partial sums.
"""
from __future__ import annotations

import numpy as np

from eabnet_amd import program as prg


def _sig(x):
    return 1.0 / (1.0 + np.exp(-x))


def _prelu(x, a):
    return np.where(x > 0, x, a * x)


def _merge_welford(st):
    """st: (B, tiles, C, 4) float64 triples (n, mean, M2, -) -> (mean, biased variance), (B, C)."""
    n, mu, m2 = st[..., 0], st[..., 1], st[..., 2]
    N = n.sum(1)
    mean = (n * mu).sum(1) / N
    M2 = m2.sum(1) + (n * (mu - mean[:, None]) ** 2).sum(1)
    return mean, M2 / N


def f16_rtz(x):
    """fp32 -> fp16 rounding toward zero (v_cvt_pkrtz_f16_f32), returned as fp32."""
    x = np.asarray(x, dtype=np.float32)
    h = x.astype(np.float16)
    over = np.abs(h.astype(np.float32)) > np.abs(x)
    h = np.where(over, np.nextafter(h, np.float16(0)), h)
    return h.astype(np.float32)


def bf16_round(x):
    """fp32 -> bf16 (round to nearest even) -> fp32, what v_cvt_pk_bf16_f32 does"""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32).reshape(np.shape(x))


def split_f16x3(x):
    hi = f16_rtz(x)
    return hi, f16_rtz(np.asarray(x, np.float32) - hi)


def unpack_f16x3(w_store, N, K):
    """inverse of eabnet_amd.program.pack_f16x3 -> (hi, lo) as fp32 [N][K]."""
    h = np.ascontiguousarray(w_store, dtype=np.float32).view(np.float16).reshape(N, K // 4, 8)
    return (h[:, :, :4].reshape(N, K).astype(np.float32), h[:, :, 4:].reshape(N, K).astype(np.float32))


class Emulator:
    def __init__(self, prog: prg.Program, x_in: np.ndarray, x_in2: np.ndarray = None):
        self.p = prog
        B, T, F = prog.B, prog.T, prog.F
        self.stages = getattr(prog.cfg, "q", 1) if x_in2 is not None else 1     # GaGNet: one planar output per stage
        self.arena = {
            "w": prog.weights.astype(np.float32),
            "a": np.full(prog.act_floats, np.nan, dtype=np.float32),   # NaN-poisoned: reads of unwritten memory show
            "in": np.ascontiguousarray(x_in, dtype=np.float32).reshape(-1),
            "out": np.full(self.stages * B * 2 * T * F, np.nan, dtype=np.float32),
        }
        if x_in2 is not None:
            self.arena["in2"] = np.ascontiguousarray(x_in2, dtype=np.float32).reshape(-1)
        for ref, n in getattr(prog, "zero_init", ()):           # arrival counters of the fused finalisation (zeroed at bind)
            self.arena[ref.arena][ref.off:ref.off + n] = 0.0

    def v(self, ref, shape):
        if ref is None:
            return None
        n = int(np.prod(shape))
        return self.arena[ref.arena][ref.off:ref.off + n].reshape(shape)

    # ------------------------------------------------------------------ ops
    def conv(self, op: prg.ConvOp):
        if op.korder == prg.KORDER_FRAG:
            # small-tile kernel: weights in MFMA-fragment order, optionally two output-column phases in one launch.
            # Interpreted as one or two launches of the tap-ordered form (what the fragment layout encodes).
            import dataclasses
            dual = op.epi in (prg.EPI_DUALGATE, prg.EPI_GLU)
            order = prg.glu_row_order(op.N) if dual else np.arange(op.N)
            bias = op.bias
            if op.epi == prg.EPI_GLU and op.bias is not None:
                # the small-tile kernel reads the bias in the convolution's own row order; the tap-ordered form interleaves
                self.arena["tmpb"] = self.v(op.bias, (op.N,))[order].copy()
                bias = prg.Ref("tmpb", 0)
            passes = [(op.w, op.Kpad, op.No, op.ophase, op.dt, op.ioff, op.stat_tile0)]
            if op.ph1_No > 0:
                passes.append((op.ph1_w, op.ph1_Kpad, op.ph1_No, op.ph1_ophase, op.ph1_dt, op.ph1_ioff,
                               op.stat_tile0 + prg.conv_tiles(op.T, op.No, op.bm)))
            for wref, K, No_, oph, dt, ioff, st0 in passes:
                self.arena["tmp"] = prg.unpack_frag(self.v(wref, (op.N * K,)), op.N, K, dual)[order].reshape(-1)
                self.conv(dataclasses.replace(op, korder=prg.KORDER_TAP, w=prg.Ref("tmp", 0), bias=bias, Kpad=K, No=No_, ophase=oph, dt=list(dt),
                                              ioff=list(ioff), stat_tile0=st0, ph1_w=None, ph1_No=0, ph1_dt=[], ph1_ioff=[], f2_w=None))
            if op.f2_w is not None:
                # fused second 1x1 convolution on the rows just written: as its own launch on this launch's output
                self.arena["tmp"] = prg.unpack_frag(self.v(op.f2_w, (op.f2_N * op.N,)), op.f2_N, op.N).reshape(-1)
                self.conv(prg.ConvOp(src0=op.dst, src1=None, xf0=None, xf1=None, slope0=None, slope1=None, C0=op.N, C1=0,
                                     xf_mode=prg.XF_NONE, w=prg.Ref("tmp", 0), bias=None, N=op.f2_N, Kpad=op.N, B=op.B, T=op.T, Fin=1,
                                     Fout=1, No=1, ostride=1, ophase=0, istride=1, dt=[0], ioff=[0], epi=prg.EPI_LINEAR, aux=None,
                                     dst=op.f2_dst, dst_acc=None, Cout=op.f2_N, stats=op.f2_stats, nsets=op.f2_nsets,
                                     stat_slope0=op.f2_stat_slope0, stat_slope1=op.f2_stat_slope1, stat_tiles=op.f2_stat_tiles,
                                     stat_tile0=0, bm=op.bm, win=op.win, name=op.name + ".f2"))
            return
        B, T, Fin, No = op.B, op.T, op.Fin, op.No
        def xform(x, tab, a):
            s, h = tab[:, None, None, :, 0], tab[:, None, None, :, 1]
            return _prelu(x * s + h, a) if op.xf_mode == prg.XF_NORM_PRELU else _prelu(x, a) * s + h

        dual = op.epi == prg.EPI_DUALGATE
        fin_tabs = None
        if op.fin_stats is not None:         # in-kernel InstanceNorm finalisation
            st = self.v(op.fin_stats, (B, op.fin_tiles, op.fin_nsets, op.C0, 4)).astype(np.float64)
            assert not np.isnan(st).any(), f"{op.name}: producer partials not fully written"
            fin_tabs = []
            for k, (g, bb) in enumerate(((op.fin_gamma0, op.fin_beta0), (op.fin_gamma1, op.fin_beta1))):
                if g is None:
                    break
                mean, var = _merge_welford(st[:, :, k])
                scale = self.v(g, (op.C0,)) / np.sqrt(var + op.fin_eps)
                fin_tabs.append(np.stack([scale, self.v(bb, (op.C0,)) - mean * scale], -1).astype(np.float32))
        srcs, X2 = [], None
        for i, (ref, xf, sl, Cs) in enumerate(((op.src0, op.xf0, op.slope0, op.C0), (op.src1, op.xf1, op.slope1, op.C1))):
            if ref is None:
                continue
            x = self.v(ref, (B, T, Fin, Cs)).astype(np.float32)
            tab = fin_tabs[0] if (fin_tabs is not None and i == 0) else (self.v(xf, (B, Cs, 2)) if xf is not None else None)
            if dual:
                assert i == 0 and op.src1 is None
                tab1 = fin_tabs[1] if fin_tabs is not None else self.v(op.xf1, (B, Cs, 2))
                X2 = xform(x, tab1, self.v(op.slope1, (Cs,)))
            if tab is not None and op.xf_mode != prg.XF_NONE:
                x = xform(x, tab, self.v(sl, (Cs,)))
            srcs.append(x)
        X = np.concatenate(srcs, axis=-1)
        Ct = X.shape[-1]
        upt = (Ct + 15) // 16
        ntaps = len(op.dt)
        assert op.Kpad == ntaps * upt * 16
        h3 = op.precision == prg.PREC_F16X3

        def by_tap(w2d):     # packed [N][Kpad] -> [N][tap][channel], whatever the unit order
            if op.korder == prg.KORDER_CHUNK:
                return w2d.reshape(op.N, upt, ntaps, 16).transpose(0, 2, 1, 3).reshape(op.N, ntaps, upt * 16)
            return w2d.reshape(op.N, ntaps, upt * 16)

        if h3:
            Wh, Wl = unpack_f16x3(self.v(op.w, (op.N, op.Kpad)), op.N, op.Kpad)
            Wh, Wl = by_tap(Wh), by_tap(Wl)
            W = Wh + Wl
        else:
            W = by_tap(self.v(op.w, (op.N, op.Kpad)))
        assert not np.any(W[:, :, Ct:]), "padding columns of the packed weights must be zero"

        bf = op.precision == prg.PREC_BF16

        def mm(G, rows, j):
            if bf:               # both operands rounded to bf16, fp32 accumulate
                return bf16_round(G) @ bf16_round(W[rows, j, :Ct]).T
            if not h3:
                return G @ W[rows, j, :Ct].T
            gh, gl = split_f16x3(G)           # the kernel's three MFMAs: lo*hi + hi*lo + hi*hi
            return (gl @ Wh[rows, j, :Ct].T + gh @ Wl[rows, j, :Ct].T) + gh @ Wh[rows, j, :Ct].T

        allrows = np.arange(op.N)
        acc = np.zeros((B, T, No, op.N), dtype=np.float32)
        if op.bias is not None:
            acc += self.v(op.bias, (op.N,))
        o = np.arange(No)
        for j in range(ntaps):
            dt, fi = op.dt[j], o * op.istride + op.ioff[j]
            ok = (fi >= 0) & (fi < Fin)
            G = np.zeros((B, T, No, Ct), dtype=np.float32)
            tv = T - abs(dt)                 # rows whose source row t+dt lies inside [0, T)
            dst_t = slice(-dt, T) if dt <= 0 else slice(0, tv)
            src_t = slice(0, tv) if dt <= 0 else slice(dt, T)
            if tv > 0:
                G[:, dst_t, ok] = X[:, src_t][:, :, fi[ok]]
            if not dual:
                acc += mm(G, allrows, j)
            else:                            # packed rows with (r % 64) < 32 are "left", the others "right"
                G2 = np.zeros_like(G)
                if tv > 0:
                    G2[:, dst_t, ok] = X2[:, src_t][:, :, fi[ok]]
                left = (np.arange(op.N) % 64) < 32
                acc[..., left] += mm(G, np.nonzero(left)[0], j)
                acc[..., ~left] += mm(G2, np.nonzero(~left)[0], j)
        if op.epi == prg.EPI_PHASE2:
            # both output-column phases of a stride-2 transposed convolution from one launch: row (t, o) writes
            # out[t][2o] (packed "value" columns) and out[t][2o+1] (packed "gate" columns, while 2o+1 < Fout); phase-1 rows of
            # the weights are zero for the taps outside p2_mask1 (checked: the kernel skips those products)
            c = np.arange(op.N // 2)
            rv = (c // 32) * 64 + c % 32
            for j in range(ntaps):
                if not (op.p2_mask1 >> j) & 1:
                    assert not np.any(W[rv + 32, j]), "phase-1 weights of a masked tap must be zero"
            assert op.ostride == 2 and op.ophase == 0 and op.aux is None and op.dst_acc is None and op.nsets <= 1
            v0, v1 = acc[..., rv], acc[..., rv + 32]
            Cout = op.N // 2
            assert Cout == op.Cout
            dst = self.v(op.dst, (B, T, op.Fout, Cout))
            n1 = op.Fout // 2
            dst[:, :, 0:2 * No:2] = v0
            dst[:, :, 1:2 * n1:2] = v1[:, :, :n1]
            if op.stats is not None:
                st = self.v(op.stats, (B, op.stat_tiles, op.nsets, Cout, 4))
                ok1 = np.tile(2 * np.arange(No) + 1 < op.Fout, T)                 # per row q = t*No + o
                r0, r1 = v0.reshape(B, T * No, Cout), v1.reshape(B, T * No, Cout)
                if op.stat_slope0 is not None:
                    a = self.v(op.stat_slope0, (Cout,))
                    r0, r1 = _prelu(r0, a), _prelu(r1, a)
                for t in range(prg.conv_tiles(T, No, op.bm)):
                    sl = slice(t * op.bm, (t + 1) * op.bm)
                    blk = np.concatenate([r0[:, sl], r1[:, sl][:, ok1[sl]]], axis=1).astype(np.float64)
                    mu = blk.mean(1)
                    st[:, op.stat_tile0 + t, 0, :, 0] = blk.shape[1]
                    st[:, op.stat_tile0 + t, 0, :, 1] = mu
                    st[:, op.stat_tile0 + t, 0, :, 2] = ((blk - mu[:, None]) ** 2).sum(1)
                    st[:, op.stat_tile0 + t, 0, :, 3] = 0.0
            return
        if op.epi in (prg.EPI_GLU, prg.EPI_DUALGATE):
            c = np.arange(op.N // 2)
            rv = (c // 32) * 64 + c % 32
            out = acc[..., rv] * _sig(acc[..., rv + 32])
        else:
            out = acc
        Cout = out.shape[-1]
        assert Cout == op.Cout
        dst = self.v(op.dst, (B, T, op.Fout, Cout))
        fo = o * op.ostride + op.ophase
        if op.epi == prg.EPI_RELU:
            out = np.maximum(out, 0)
        elif op.epi == prg.EPI_MULSIG:
            out = self.v(op.aux, (B, T, op.Fout, Cout))[:, :, fo] * _sig(out)
        elif op.epi == prg.EPI_ADD:
            out = out + self.v(op.aux, (B, T, op.Fout, Cout))[:, :, fo]
        dst[:, :, fo] = out
        if op.dst_acc is not None:
            self.v(op.dst_acc, (B, T, op.Fout, Cout))[:, :, fo] += out
        if op.stats is not None:
            st = self.v(op.stats, (B, op.stat_tiles, op.nsets, Cout, 4))
            rows = out.reshape(B, T * No, Cout)
            tiles = prg.conv_tiles(T, No, op.bm)
            for s, slr in enumerate((op.stat_slope0, op.stat_slope1)[:op.nsets]):
                g = rows if slr is None else _prelu(rows, self.v(slr, (Cout,)))
                for t in range(tiles):       # Welford triple (n, mean, M2, 0) of the tile's valid rows
                    blk = g[:, t * op.bm:(t + 1) * op.bm].astype(np.float64)
                    mu = blk.mean(1)
                    st[:, op.stat_tile0 + t, s, :, 0] = blk.shape[1]
                    st[:, op.stat_tile0 + t, s, :, 1] = mu
                    st[:, op.stat_tile0 + t, s, :, 2] = ((blk - mu[:, None]) ** 2).sum(1)
                    st[:, op.stat_tile0 + t, s, :, 3] = 0.0
            if op.fz_counter is not None:
                # fused finalisation: the tile that completes the count of a batch element merges all partials (the
                # counter is an int32 in the kernel; its bit pattern is kept in the fp32 arena here) and re-arms it
                cnt = self.v(op.fz_counter, (B,)).view(np.int32)
                cnt += tiles
                assert np.all(cnt <= op.stat_tiles)
                if np.all(cnt == op.stat_tiles):
                    full = st.astype(np.float64)
                    assert not np.isnan(full).any(), f"{op.name}: statistics partials not fully written"
                    for s, (g, b, xf) in enumerate(((op.fz_gamma0, op.fz_beta0, op.fz_xf0), (op.fz_gamma1, op.fz_beta1, op.fz_xf1))[:op.nsets]):
                        mean, var = _merge_welford(full[:, :, s])
                        scale = self.v(g, (Cout,)) / np.sqrt(var + op.fz_eps)
                        o = self.v(xf, (B, Cout, 2))
                        o[..., 0], o[..., 1] = scale, self.v(b, (Cout,)) - mean * scale
                    cnt[:] = 0

    def finalize(self, op: prg.FinalizeOp):
        st = self.v(op.stats, (op.B, op.stat_tiles, op.nsets, op.C, 4)).astype(np.float64)
        assert not np.isnan(st).any(), f"{op.name}: statistics partials not fully written"
        for s, (g, b, xf) in enumerate(((op.gamma0, op.beta0, op.xf0), (op.gamma1, op.beta1, op.xf1))[:op.nsets]):
            mean, var = _merge_welford(st[:, :, s])
            assert np.all(st[:, :, s, :, 0].sum(1) == op.count), f"{op.name}: partial counts do not add up"
            scale = self.v(g, (op.C,)) / np.sqrt(var + op.eps)
            shift = self.v(b, (op.C,)) - mean * scale
            out = self.v(xf, (op.B, op.C, 2))
            out[..., 0], out[..., 1] = scale, shift

    def cln_stats(self, op):
        x = self.v(op.x, (op.B, op.T, op.P)).astype(np.float64)
        if op.slope is not None:
            a = np.tile(self.v(op.slope, (op.C,)).astype(np.float64), op.P // op.C)
            x = np.where(x > 0, x, a * x)
        lo, hi = getattr(self, "_win", (0, op.T))
        sums = self.v(op.sums, (op.B, op.T, 4)).view(np.float64).reshape(op.B, op.T, 2)
        sums[:, lo:hi, 0], sums[:, lo:hi, 1] = x[:, lo:hi].sum(-1), (x[:, lo:hi] ** 2).sum(-1)
        st = self.v(op.state, (op.B, 4)).view(np.float64).reshape(op.B, 2) if op.state is not None else None
        cs = st[:, 0].copy() if (st is not None and lo > 0) else np.zeros(op.B)
        cq = st[:, 1].copy() if (st is not None and lo > 0) else np.zeros(op.B)
        mr = self.v(op.mr, (op.B, op.T, 2))
        for t in range(lo, hi):
            cs += sums[:, t, 0]
            cq += sums[:, t, 1]
            cnt = op.P * (t + 1.0)
            mean = cs / cnt
            var = (cq - 2.0 * mean * cs) / cnt + mean * mean
            mr[:, t, 0], mr[:, t, 1] = mean, 1.0 / np.sqrt(var + op.eps)
        if st is not None:
            st[:, 0], st[:, 1] = cs, cq

    def cln_apply(self, op):
        lo, hi = getattr(self, "_win", (0, op.T))
        x = self.v(op.x, (op.B, op.T, op.P // op.C, op.C))[:, lo:hi]
        mr = self.v(op.mr, (op.B, op.T, 2))[:, lo:hi]
        g, b, a = (self.v(r, (op.C,)) for r in (op.gain, op.bias, op.slope))
        mean, rstd = mr[..., 0][:, :, None, None], mr[..., 1][:, :, None, None]
        if op.mode == prg.XF_NORM_PRELU:
            y = _prelu((x - mean) * rstd * g + b, a)
        else:
            y = (_prelu(x, a) - mean) * rstd * g + b
        if op.add is not None:
            y = y + self.v(op.add, (op.B, op.T, op.P // op.C, op.C))[:, lo:hi]
        self.v(op.out, (op.B, op.T, op.P // op.C, op.C))[:, lo:hi] = y.astype(np.float32)

    def gate_rows(self, op):
        lo, hi = getattr(self, "_win", (0, op.T))
        a, r = (self.v(q, (op.B, op.T, op.row))[:, lo:hi] for q in (op.a, op.r))
        self.v(op.z, (op.B, op.T, op.row))[:, lo:hi] = (a * _sig(r)).astype(np.float32)

    def norm_act(self, op: prg.NormActOp):
        def f(ref, xf, sl):
            tab = self.v(xf, (op.B, op.C, 2))
            return _prelu(self.v(ref, (op.B, op.P, op.C)) * tab[:, None, :, 0] + tab[:, None, :, 1], self.v(sl, (op.C,)))
        r = f(op.a, op.xfa, op.slopea)
        if op.b is not None:
            r = r + f(op.b, op.xfb, op.slopeb)
        self.v(op.out, (op.B, op.P, op.C))[:] = r

    def lstm(self, op: prg.LstmOp):
        B, T, F = op.B, op.T, op.F
        x = self.v(op.x, (B, T, F, 64)).astype(np.float32)
        if op.ln_g is not None:
            mu = x.mean(-1, keepdims=True)
            var = ((x - mu) ** 2).mean(-1, keepdims=True)
            x = (x - mu) / np.sqrt(var + op.ln_eps) * self.v(op.ln_g, (64,)) + self.v(op.ln_b, (64,))
        W, bias = self.v(op.wcat, (256, 128)), self.v(op.bias, (256,))
        h = np.zeros((B, F, 64), np.float32)
        c = np.zeros((B, F, 64), np.float32)
        out = self.v(op.h_out, (B, T, F, 64))
        h3 = op.precision == prg.PREC_F16X3
        if h3:      # weights split with round-to-nearest in the kernel, activations with round-toward-zero
            Wh = W.astype(np.float16).astype(np.float32)
            Wl = (W - Wh).astype(np.float16).astype(np.float32)
        bf = op.precision == prg.PREC_BF16
        if bf:
            Wb = bf16_round(W)
        for t in range(T):
            a = np.concatenate([x[:, t], h], -1)
            if bf:
                g = bf16_round(a) @ Wb.T + bias
            elif h3:
                ah, al = split_f16x3(a)
                g = ((al @ Wh.T + ah @ Wl.T) + ah @ Wh.T) + bias
            else:
                g = a @ W.T + bias
            i, f, gg, o = np.split(g, 4, -1)
            c = _sig(f) * c + _sig(i) * np.tanh(gg)
            h = (_sig(o) * np.tanh(c)).astype(np.float32)
            out[:, t] = h
            if h3:  # the kernel carries h as fp16 hi + lo (both round-to-nearest)
                hh = h.astype(np.float16).astype(np.float32)
                h = hh + (h - hh).astype(np.float16).astype(np.float32)
                out[:, t] = h
            # bf16: h_t leaves as exact fp32; the recurrence multiplies its bf16 rounding (applied to `a` above)

    def bfw(self, op: prg.BfwOp):
        B, T, F, M = op.B, op.T, op.F, op.M
        y1 = self.v(op.y1, (B, T, F, 64))
        if op.w1 is not None:           # fused first Linear + ReLU (exact fp32 in every precision mode)
            y1 = np.maximum(y1 @ self.v(op.w1, (64, 64)).T + self.v(op.b1, (64,)), 0.0).astype(np.float32)
        w = y1 @ self.v(op.w2, (2 * M, 64)).T + self.v(op.b2, (2 * M,))
        w = w.reshape(B, T, F, M, 2)
        if op.bfw is not None:
            self.v(op.bfw, (B, T, F, M, 2))[:] = w
        x = self.v(op.x, (B, T, F, M, 2))
        out = self.v(op.out, (B, 2, T, F))
        out[:, 0] = (w[..., 0] * x[..., 0] - w[..., 1] * x[..., 1]).sum(-1)
        out[:, 1] = (w[..., 0] * x[..., 1] + w[..., 1] * x[..., 0]).sum(-1)

    def gag_pack(self, op: prg.GagPackOp):
        B, T, F = op.B, op.T, op.F
        a, b = self.v(op.inpt, (B, 2, T, F)), self.v(op.pre_x, (B, 2, T, F))
        enc = self.v(op.enc_in, (B, T, F, 4))
        enc[..., 0], enc[..., 1], enc[..., 2], enc[..., 3] = a[:, 0], a[:, 1], b[:, 0], b[:, 1]
        pre = self.v(op.pre, (B, T, prg.GAG_PRE_LD))
        pre[:] = 0.0
        pre[:, :, 0:2 * F:2], pre[:, :, 1:2 * F:2] = b[:, 0], b[:, 1]

    def gag_crm(self, op: prg.GagCrmOp):
        B, T, F = op.B, op.T, op.F
        pre = self.v(op.pre, (B, T, prg.GAG_PRE_LD))[:, :, :2 * F].reshape(B, T, F, 2)
        g, r, i = (self.v(x, (B, T, prg.GAG_LIN_LD))[:, :, :F] for x in (op.g, op.r, op.i))
        gain = {prg.ACT_SIGMOID: _sig, prg.ACT_TANH: np.tanh, prg.ACT_RELU: lambda v: np.maximum(v, 0)}[op.act](g)
        yr, yi = pre[..., 0] * gain + r, pre[..., 1] * gain + i
        nxt = self.v(op.pre_out, (B, T, prg.GAG_PRE_LD))
        nxt[:] = 0.0
        nxt[:, :, 0:2 * F:2], nxt[:, :, 1:2 * F:2] = yr, yi
        out = self.v(op.planar, (B, 2, T, F))
        out[:, 0], out[:, 1] = yr, yi

    def step(self, op):
        with np.errstate(over="ignore"):
            {prg.OP_CONV: self.conv, prg.OP_IN_FINALIZE: self.finalize, prg.OP_NORM_ACT: self.norm_act,
             prg.OP_LSTM64: self.lstm, prg.OP_BFW_FS: self.bfw, prg.OP_GAG_PACK: self.gag_pack,
             prg.OP_GAG_CRM: self.gag_crm,
             prg.OP_CLN_STATS: self.cln_stats, prg.OP_CLN_APPLY: self.cln_apply, prg.OP_GATE_ROWS: self.gate_rows,
             prg.OP_MEMSET0: lambda o: self.v(o.ptr, (o.nfloats,)).fill(0)}[op.kind](op)

    def _outputs(self, op):
        """(ref, shape, time axis) of everything an op writes, for the streaming window emulation"""
        p = self.p
        if op.kind == prg.OP_CONV:
            o = [(op.dst, (op.B, op.T, op.Fout * op.Cout), 1)]
            return o + ([(op.dst_acc, (op.B, op.T, op.Fout * op.Cout), 1)] if op.dst_acc is not None else [])
        if op.kind == prg.OP_NORM_ACT:
            return [(op.out, (op.B, op.T, (op.P // op.T) * op.C), 1)]
        if op.kind == prg.OP_LSTM64:
            return [(op.h_out, (op.B, op.T, op.F * 64), 1)]
        if op.kind == prg.OP_BFW_FS:
            o = [(op.out, (op.B, 2, op.T, op.F), 2)]
            return o + ([(op.bfw, (op.B, op.T, op.F * op.M * 2), 1)] if op.bfw is not None else [])
        if op.kind == prg.OP_MEMSET0:
            return [(op.ptr, (op.B, op.T, op.row), 1)]
        if op.kind in (prg.OP_CLN_STATS, prg.OP_CLN_APPLY, prg.OP_GATE_ROWS):
            return []                               # these emulate their window (and carried state) themselves: self._win
        if op.kind == prg.OP_GAG_PACK:
            return [(op.enc_in, (op.B, op.T, op.F * 4), 1), (op.pre, (op.B, op.T, prg.GAG_PRE_LD), 1)]
        if op.kind == prg.OP_GAG_CRM:
            return [(op.pre_out, (op.B, op.T, prg.GAG_PRE_LD), 1), (op.planar, (op.B, 2, op.T, op.F), 2)]
        raise ValueError(op.kind)

    def run_stream(self):
        """Streaming semantics (eab_time_window): for each chunk every op may only change the time rows
        [pos, pos + chunk) of its outputs.  Emulated by running the op on the whole utterance (rows it has
        not reached yet hold NaN poison; causality keeps that out of the window) and restoring every row
        outside the window.  The LSTM's carried state equals a recomputation from t = 0, so it needs no
        special case here; the device kernel's state handling is what the GPU test checks against this."""
        p = self.p
        assert p.chunk > 0 and all(getattr(op, "win", False) for op in p.ops)
        for pos in range(0, p.T, p.chunk):
            hi = min(pos + p.chunk, p.T)
            self._win = (pos, hi)
            for op in p.ops:
                outs = self._outputs(op)
                saved = [self.v(r, shp).copy() for r, shp, _ in outs]
                with np.errstate(invalid="ignore"):
                    self.step(op)
                for (r, shp, ax), old in zip(outs, saved):
                    cur = self.v(r, shp)
                    idx = [slice(None)] * len(shp)
                    idx[ax] = slice(pos, hi)
                    new = cur[tuple(idx)].copy()
                    cur[:] = old
                    cur[tuple(idx)] = new
        if "in2" in self.arena:
            return self.arena["out"].reshape(self.stages, p.B, 2, p.T, p.F)
        return self.arena["out"].reshape(p.B, 2, p.T, p.F)

    def run(self):
        for op in self.p.ops:
            self.step(op)
        p = self.p
        if self.stages > 1 or "in2" in self.arena:
            return self.arena["out"].reshape(self.stages, p.B, 2, p.T, p.F)
        return self.arena["out"].reshape(p.B, 2, p.T, p.F)

    def act(self, a: prg.Act) -> np.ndarray:
        """Named activation as NCHW (B, C, T, F) like the reference's hooks."""
        p = self.p
        return self.v(a.ref, (p.B, p.T, a.F, a.C)).transpose(0, 3, 1, 2)
