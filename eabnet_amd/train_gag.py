"""Training of the GaGNet post-filter on the hand-written kernels (SURVEY §8 rows N1 x N3).

The reference trains the two-stage model with autograd (train_distributed.py:181,218-230: EaBNetWithPostNet, the post-filter
fed esti0.detach(), EaBNet.py:142).  Here GaGNet.forward (GaGNet.py:76-90) and its backward are two static op programs behind
one torch.autograd.Function, built with the machinery of eabnet_amd/train.py (TrainLowering: tape of late-binding closures,
gradient Slots, index images for parameter packing, deferred + batched weight gradients).  What is specific to GaGNet:

  * inputs inpt, pre_x (B,2,T,F) planar -> gag_pack (enc_in [B][T][F][4], pre [B][T][384]); neither needs a gradient;
  * U2-encoder = the same blocks as EaBNet's (TrainLowering.unet_module / conv2d_fwd), 4 input channels;
  * a GlanceGazeModule (GaGNet.py:93-133): two gated 1x1 in-convs on cat(feat, pre) = ONE two-source gated convolution each
    (N = 512, GLU epilogue with factor dump), three chains of p*|dilas| single-branch S-TCMs (GaGNet.py:303-327), three
    linears d_feat -> 161 (rows padded to 192), the gain/residual tail y = pre * act(g) + (r, i) (eab_gag_crm_f32) whose
    backward is eab_gag_crm_bwd_f32;
  * the q stage outputs live in the 'out' arena [q][B][2][T][F]; their gradients arrive in 'dout' (same shape).

`pre` rows are 384 floats here (the inference program uses 324): the gradient w.r.t. pre is a dgrad convolution whose output
channel count must be a multiple of 64."""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import numpy as np
import torch

from . import _lib
from . import program as prg
from . import train as tr
from .program import Ref, glu_row_order
from .spec import GagConfig, gag_param_specs
from .train import GenOp, TVar, TrainBound, TrainLowering, TrainProgram, _split64

OP_GAG_PACK, OP_GAG_CRM, OP_GAG_CRM_BWD = prg.OP_GAG_PACK, prg.OP_GAG_CRM, 36
PRE_LD = 384            # floats per (b, t) row of the interleaved previous estimate: 2*161 padded to a multiple of 64
LIN_LD = prg.GAG_LIN_LD


def supported(cfg: GagConfig) -> bool:
    return (cfg.intra_connect in ("cat", "add") and cfg.norm_type in ("IN", "BN") and cfg.cin == 2 and 2 * cfg.freq <= PRE_LD)


class GagTrainLowering(TrainLowering):
    spec_fn = staticmethod(gag_param_specs)
    supports = staticmethod(supported)

    # ---- blocks ------------------------------------------------------------------------------------------------
    def gated_in(self, pfx: str, feat: TVar, pre: TVar, feat_perm: np.ndarray) -> TVar:
        """in_conv_main(cat) * sigmoid(in_conv_gate(cat)) (GaGNet.py:191,251): one gated two-source 1x1 convolution"""
        cfg = self.cfg
        D, Fq = cfg.d_feat, cfg.freq
        w = np.concatenate([self.idx(f"{pfx}.in_conv_main.weight"), self.idx(f"{pfx}.in_conv_gate.0.weight")], axis=0)[:, :, 0]
        cols = np.arange(2 * Fq)
        cols_pre = D + (cols % 2) * Fq + cols // 2                  # memory channel f*2+ri <- reference ri*F+f
        wk = np.full((2 * D, D + PRE_LD), -1, dtype=np.int64)
        wk[:, :D] = w[:, feat_perm]
        wk[:, D:D + 2 * Fq] = w[:, cols_pre]
        order = glu_row_order(2 * D)
        wk = wk[order]
        wimg = self.pack_taps_idx(wk[:, :, None], [0])
        bimg = np.concatenate([self.idx(f"{pfx}.in_conv_main.bias"), self.idx(f"{pfx}.in_conv_gate.0.bias")])[order]
        out = self.act(1, D)
        rows = self.B * self.T
        dump = self.alloc(rows * 2 * D)
        self.conv_op(f"{pfx}.in_conv", [feat, pre], self.wadd(f"{pfx}.in_conv.w", wimg), self.wadd(f"{pfx}.in_conv.b", bimg), 2 * D,
                     wimg.shape[1], 1, 1, 1, 1, 0, 1, [0], [0], prg.EPI_GLU, out.ref, D, bm=64, glu_dump=dump)

        def back():
            dz = self.alloc(rows * 2 * D)
            self.bwd.append(GenOp(tr.OP_GLU_BWD, [self.grad_of(out), dump, dz], list(_split64(rows)) + [2 * D], name=f"{pfx}.in_conv.glu_bwd"))
            self.wgrad_op(f"{pfx}.in_conv.wgrad", dz, 2 * D, 1, [feat, pre], 1, 1, 0, 1, [0], [0], wimg, dbias=self.gadd([bimg]))
            c_lo = 0
            for s in (feat, pre):
                if s.needs_grad:
                    img = self.pack_taps_idx(np.ascontiguousarray(wk[:, c_lo:c_lo + s.C].T)[:, :, None], [0])      # (C_s, 2D, 1)
                    self.dgrad(f"{pfx}.in_conv.dgrad{c_lo}", s, dz, 2 * D, 1, [(self.wadd(f"{pfx}.in_conv.wd{c_lo}", img), 1, 1, 0, 1, [0], [0])])
                c_lo += s.C
        self.tape.append(back)
        return out

    def tcm1(self, pre: str, x: TVar, dilation: int) -> TVar:
        """GaGNet's single-branch SqueezedTCM (GaGNet.py:303-327): in_conv -> PReLU/norm/dilated conv -> PReLU/norm/out_conv
        + residual"""
        cfg = self.cfg
        kd = cfg.kd1
        y, back_in = self.conv1d(f"{pre}.in_conv", x, self.idx(f"{pre}.in_conv.weight"), [0], None, prg.EPI_LINEAR, wname=f"{pre}.in_conv")
        self.tape.append(lambda: back_in(self.grad_of(y)))
        yd = self.in1d(f"{pre}.d", y, f"{pre}.d_conv.1", f"{pre}.d_conv.0")
        span = (kd - 1) * dilation
        lead = span if cfg.is_causal else span // 2
        dts = [j * dilation - lead for j in range(kd)]
        z, back_d = self.conv1d(f"{pre}.d_conv", yd, self.idx(f"{pre}.d_conv.3.weight"), dts, None, prg.EPI_LINEAR, wname=f"{pre}.d_conv")
        self.tape.append(lambda: back_d(self.grad_of(z)))
        zo = self.in1d(f"{pre}.out", z, f"{pre}.out_conv.1", f"{pre}.out_conv.0")
        out, back_out = self.conv1d(f"{pre}.out_conv", zo, self.idx(f"{pre}.out_conv.2.weight"), [0], None, prg.EPI_ADD, aux=x,
                                    wname=f"{pre}.out_conv")

        def back():
            d = self.grad_of(out)
            self.contribute(x, d)                                            # residual
            back_out(d)
        self.tape.append(back)
        return out

    def chain(self, pre: str, x: TVar) -> TVar:
        for j in range(self.cfg.p):
            for k, d in enumerate(self.cfg.dilas):
                x = self.tcm1(f"{pre}.{j}.tcns.{k}", x, d)
        return x

    def linear(self, key: str, x: TVar) -> Tuple[TVar, Callable[[Ref], None]]:
        """Conv1d(d_feat -> 161, 1) (GaGNet.py:176,241), rows padded to LIN_LD (padded rows: zero weights, gradient discarded)"""
        D, Fq = self.cfg.d_feat, self.cfg.freq
        w = np.full((LIN_LD, D, 1), -1, dtype=np.int64)
        b = np.full(LIN_LD, -1, dtype=np.int64)
        w[:Fq], b[:Fq] = self.idx(f"{key}.weight"), self.idx(f"{key}.bias")
        return self.conv1d(key, x, w, [0], b, prg.EPI_LINEAR, wname=key)

    # ---- whole network -----------------------------------------------------------------------------------------
    def build(self) -> TrainProgram:
        cfg, B, T, F = self.cfg, self.B, self.T, self.F
        assert F == cfg.freq
        c = cfg.c
        enc_in = TVar(self.alloc(B * T * F * 4), F, 4, tr.Slot(), needs_grad=False)
        pre = TVar(self.alloc(B * T * PRE_LD), 1, PRE_LD, tr.Slot(), needs_grad=False)
        self.fwd.append(GenOp(OP_GAG_PACK, [Ref("in"), Ref("in2"), enc_in.ref, pre.ref], [B, T, F, PRE_LD], name="pack"))
        x = enc_in
        if cfg.is_u2:
            for i in range(4):
                x = self.unet_module(f"en.meta_unet_list.{i}", [x], 4 - i, False)
            x = self.conv2d_fwd("en.last_conv", [x], "en.last_conv.0.conv.1", True, "en.last_conv.1", "en.last_conv.2")
        else:
            for i in range(5):                          # GaGNet's UNet_Encoder (GaGNet.py:417-452): every layer normed
                q = f"en.unet_list.{i}"
                x = self.conv2d_fwd(q, [x], f"{q}.0.conv.1", True, f"{q}.1", f"{q}.2")
        assert x.F * x.C == cfg.d_feat
        k = np.arange(cfg.d_feat)
        feat_perm = (k % c) * x.F + k // c                 # memory channel f*64+c <- reference c*4+f (GaGNet.py:83-84)
        feat = x.view(1, cfg.d_feat)
        self.gtaps["feat"] = feat
        act = {"sigmoid": prg.ACT_SIGMOID, "tanh": prg.ACT_TANH, "relu": prg.ACT_RELU}[cfg.acti_type]
        n_stage = B * 2 * T * F
        for gi in range(cfg.q):
            gl, gz = f"gags.{gi}.glance_block", f"gags.{gi}.gaze_block"
            xg0 = self.gated_in(gl, feat, pre, feat_perm)
            xz = self.gated_in(gz, feat, pre, feat_perm)
            # The glance chain and the two gaze chains only meet again in the tail; each of their launches fills a fraction
            # of the chip (57 workgroups at 6 x 6 s), so they run as three parallel branches, forward and backward.  The gaze
            # chains both read xz: each gets its own gradient slot, summed after the backward join (two lanes must not
            # accumulate into one buffer).
            xz_r, xz_i = TVar(xz.ref, xz.F, xz.C, tr.Slot()), TVar(xz.ref, xz.F, xz.C, tr.Slot())

            def join_back(xz=xz, xz_r=xz_r, xz_i=xz_i):
                self.mark("bwd", "join", [1, 2])
                self.contribute(xz, self.grad_of(xz_r))
                if not cfg.is_squeezed:
                    self.contribute(xz, self.grad_of(xz_i))
            self.tape.append(join_back)                 # replayed after the chains' closures, before the gated in-convs'
            self.mark("fwd", "fork", [1, 2])
            gain, back_g = self.linear(f"{gl}.linear_g.0", self.chain(f"{gl}.tcn_g", xg0))
            self.cur_lane = 1
            if cfg.is_squeezed:                         # one chain feeds both linear maps (GaGNet.py:236-237,255)
                ri = self.chain(f"{gz}.tcm_ri", xz_r)
                lr, back_r = self.linear(f"{gz}.linear_r", ri)
                li, back_i = self.linear(f"{gz}.linear_i", ri)
            else:
                lr, back_r = self.linear(f"{gz}.linear_r", self.chain(f"{gz}.tcm_r", xz_r))
                self.cur_lane = 2
                li, back_i = self.linear(f"{gz}.linear_i", self.chain(f"{gz}.tcm_i", xz_i))
            self.cur_lane = 0
            self.mark("fwd", "join", [1, 2])
            nxt = TVar(self.alloc(B * T * PRE_LD), 1, PRE_LD, tr.Slot(), needs_grad=gi + 1 < cfg.q)
            self.fwd.append(GenOp(OP_GAG_CRM, [pre.ref, gain.ref, lr.ref, li.ref, nxt.ref, Ref("out", gi * n_stage)],
                                  [B, T, F, PRE_LD, LIN_LD, act], name=f"gags.{gi}.crm"))

            def back(gi=gi, pre=pre, gain=gain, nxt=nxt, back_g=back_g, back_r=back_r, back_i=back_i):
                n = B * T * LIN_LD
                dg, dr, di = self.alloc(n), self.alloc(n), self.alloc(n)
                dst, aux = self.grad_target(pre) if pre.needs_grad else (None, None)
                self.bwd.append(GenOp(OP_GAG_CRM_BWD, [pre.ref, gain.ref, Ref("dout", gi * n_stage), nxt.slot.ref, aux, dg, dr, di, dst],
                                      [B, T, F, PRE_LD, LIN_LD, act], name=f"gags.{gi}.crm_bwd"))
                self.mark("bwd", "fork", [1, 2])
                back_g(dg)
                self.cur_lane = 1
                back_r(dr)
                self.cur_lane = 1 if cfg.is_squeezed else 2
                back_i(di)
                self.cur_lane = 0
            self.tape.append(back)
            pre = nxt
        for fn in reversed(self.tape):
            fn()
        prog = self.finish()
        prog.out_shape = (cfg.q, B, 2, T, F)
        prog.has_in2 = True
        return prog


def lower_train(cfg: GagConfig, B: int, T: int, F: int = 161, precision: str = "f32") -> TrainProgram:
    return GagTrainLowering(cfg, B, T, F, precision).build()


class _GagTrainFn(torch.autograd.Function):
    """One autograd node for the whole post-filter: forward program, backward program."""

    @staticmethod
    def forward(ctx, bound: TrainBound, inpt: torch.Tensor, pre_x: torch.Tensor, *params: torch.Tensor) -> torch.Tensor:
        prog = bound.prog
        st = torch.cuda.current_stream().cuda_stream
        flat = torch.cat([p.detach().reshape(-1) for p in params])      # (one batched copy; fp32 parameters)
        if flat.dtype != torch.float32:
            flat = flat.to(torch.float32)
        bound.pack(flat, st)
        bound.serial += 1
        ctx.serial = bound.serial
        if bound.capture(tuple(inpt.shape)):
            bound.static_x.copy_(inpt)
            bound.static_x2.copy_(pre_x)
            bound.graphs[0].replay()
            out = bound.static_out.clone()
            ctx.io = None
        else:
            out = torch.empty(prog.out_shape, dtype=torch.float32, device=inpt.device)
            dout = torch.empty_like(out)
            bound.bind(inpt.data_ptr(), out.data_ptr(), dout.data_ptr(), pre_x.data_ptr())
            bound.run("fwd", st)
            ctx.io = (inpt, pre_x, out, dout)
        ctx.bound = bound
        ctx.shapes = [p.shape for p in params]
        ctx.dtypes = [p.dtype for p in params]
        return out

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        bound, prog = ctx.bound, ctx.bound.prog
        st = torch.cuda.current_stream().cuda_stream
        if ctx.serial != bound.serial:
            raise RuntimeError("eabnet_amd: backward of a forward pass whose saved activations were overwritten by a later "
                               "forward of the same module (one training program holds one set of activations)")
        bound.g.zero_()
        if ctx.io is None:
            bound.static_dout.copy_(grad_out)
            bound.graphs[1].replay()
        else:
            inpt, pre_x, out, dout = ctx.io
            dout.copy_(grad_out.to(torch.float32))
            bound.bind(inpt.data_ptr(), out.data_ptr(), dout.data_ptr(), pre_x.data_ptr())
            bound.run("bwd", st)
        gflat = torch.empty(prog.n_params, dtype=torch.float32, device=bound.device)
        bound.unpack_grads(gflat, st)
        grads = tr.finish_flat_gradient(gflat, getattr(ctx.bound, "sync_group", None), ctx.shapes, ctx.dtypes, ctx.needs_input_grad[3:])
        return (None, None, None, *grads)


def forward_train(module, inpt: torch.Tensor, pre_x: torch.Tensor) -> List[torch.Tensor]:
    """GaGNet.forward under autograd on the HIP training programs: list of q estimates (B, 2, F, T) (views of one
    (q, B, 2, T, F) tensor).  No gradient flows to inpt / pre_x (the reference detaches the beam-former's estimate)."""
    _lib.load()
    B, _, T, F = inpt.shape
    a = inpt.detach().to(torch.float32).contiguous()
    b = pre_x.detach().to(torch.float32).contiguous()
    cache = module.__dict__.setdefault("_train_bound", {})
    prec = "bf16" if module.precision == "bf16" else "f32"
    key = (B, T, F, str(a.device), prec)
    bound = cache.pop(key, None)
    if bound is None:
        # a small LRU of bound programs (variable-length batches, a smaller last batch, alternating train / validation
        # shapes): re-lowering and re-capturing two hipGraphs on every shape change costs seconds
        while len(cache) >= tr.TRAIN_BOUND_CACHE:
            torch.cuda.synchronize(a.device)      # the dropped program's arenas may still be read by kernels in flight
            cache.pop(next(iter(cache)))
        with torch.cuda.device(a.device):
            bound = TrainBound(lower_train(module.cfg, B, T, F, prec), a.device)
    cache[key] = bound                               # most recently used last
    bound.use_graph = bool(getattr(module, "use_graph", True)) and not torch.cuda.is_current_stream_capturing()
    bound.sync_group = module.__dict__.get("grad_allreduce", None)
    sd = dict(module.named_parameters())
    params = [sd[k] for k in bound.prog.keys]
    with torch.cuda.device(a.device):
        out = _GagTrainFn.apply(bound, a, b, *params)
        bound.update_bn_buffers(module)
    out = out.to(inpt.dtype)
    return [out[j].permute(0, 1, 3, 2) for j in range(out.shape[0])]
