"""Training on the hand-written kernels (BASELINE config 4): EaBNet.forward and its backward pass
(what autograd executes for train_distributed.py:221-228 on the reference) lowered to TWO static op lists that
libeabnet_hip.so replays -- forward (keeping what the backward needs) and backward -- behind one
``torch.autograd.Function``.  PyTorch sees a single node: parameters in, (B,2,T,F) out; gradients of all
parameters come back as views of one flat buffer.

Differences to the inference lowering (program.py): every normalised activation is materialised (the
convolutions read plain sources, so the same tensors are the operands of the weight gradients); gated
convolutions also dump their two GLU factors; InstanceNorm finalisation keeps (mean, rstd); the LSTM layers keep
their activated gates and cell states; the MLP and the filter-and-sum are separate ops.

Backward contractions:
  * dgrad = eab_conv_f32 on the output gradient with re-packed weights: the dgrad of a strided convolution
    (GateConv2d / Conv2dunit, EaBNet.py:402,450) is the gather form of a transposed convolution, one launch per
    input-column parity; the dgrad of a transposed convolution (EaBNet.py:423,478) is a strided convolution;
  * wgrad = eab_wgrad_f32 (csrc/wgrad.hip) with the forward's own gather geometry;
  * LSTM: eab_lstm64_bwd_f32 (reverse time) -> dgates, then dx / dW_ih / dW_hh / db as a 1x1 dgrad, two wgrads
    (taps dt = 0 and dt = -1) and a column sum.
Parameters reach the kernels through ONE gather launch (flat parameter vector -> every packed operand, forward
and dgrad layouts; index tables built here from the same packing rules as program.py) and the packed gradient
arena goes back to the flat gradient through the inverse table.

Scope: every constructor branch of the reference (`supported`): U2 / plain U-Net, LSTM / cnn head, mimo / miso, cat / add,
causal or not, InstanceNorm, BatchNorm (train mode) and cLN, up to 32 microphones.  This is the package's only differentiable
path; what it does not cover (a gradient w.r.t. the input, eval-mode BatchNorm under autograd) is refused by model.py.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import os

import numpy as np
import torch

from . import _lib
from . import program as prg
from .graphs import LaneGraphs, plan_segments, single_lane
from .program import ALIGN, EPS_IN, EPS_LN, Ref, conv_tiles, glu_row_order
from .spec import NetConfig, gate_key, param_specs

XF_NORM_PRELU, XF_PRELU_NORM = prg.XF_NORM_PRELU, prg.XF_PRELU_NORM
(OP_GATHER, OP_IN_STATS, OP_TR_NORM_ACT, OP_NORM_BWD, OP_GLU_BWD, OP_GATE_FWD, OP_GATE_BWD, OP_ADD, OP_RELU_BWD, OP_COLSUM,
 OP_FILTER_SUM, OP_FS_BWD, OP_LN_FWD, OP_LN_BWD, OP_LSTM_TRAIN, OP_LSTM_BWD, OP_WGRAD) = range(16, 33)
OP_CLN_STATS, OP_CLN_APPLY, OP_CLN_BWD = prg.OP_CLN_STATS, prg.OP_CLN_APPLY, 37      # include/eabnet_hip.h EAB_OP_CLN_*
NB_SUMS_ZEROED = 0x100   # include/eabnet_hip.h EAB_NB_SUMS_ZEROED
NB_SUM_COPIES = 8        # include/eabnet_hip.h EAB_NB_SUM_COPIES: the reduce pass spreads its atomics over that many copies
STORE_BF16 = 0x200       # include/eabnet_hip.h EAB_STORE_BF16: the op's output tensor is stored as bf16
MLP_LD = 64          # the second Linear of w_dnn is run with its 2M rows padded to one 64-column tile
TRAIN_BOUND_CACHE = 3   # bound training programs kept per module (LRU over (B, T, F, device, precision))


def supported(cfg: NetConfig) -> bool:
    return (cfg.bf_type in ("lstm", "cnn") and cfg.topo_type in ("mimo", "miso") and cfg.intra_connect in ("cat", "add")
            and cfg.norm_type in ("IN", "BN", "cLN") and 2 * cfg.M <= MLP_LD)


def unsupported_reason(cfg: NetConfig) -> str:
    """which constructor switch keeps a configuration off the HIP training programs ('' when supported)"""
    why = []
    if cfg.bf_type not in ("lstm", "cnn"):
        why.append(f"bf_type={cfg.bf_type!r}")
    if cfg.topo_type not in ("mimo", "miso"):
        why.append(f"topo_type={cfg.topo_type!r}")
    if cfg.intra_connect not in ("cat", "add"):
        why.append(f"intra_connect={cfg.intra_connect!r}")
    if cfg.norm_type not in ("IN", "BN", "cLN"):
        why.append(f"norm_type={cfg.norm_type!r}")
    if 2 * cfg.M > MLP_LD:
        why.append(f"M={cfg.M} > {MLP_LD // 2}")
    return ", ".join(why)


@dataclass
class GenOp:
    """A non-convolution op of the training programs: kind + the i / f / p fields of eab_op."""
    kind: int
    p: List[Optional[Ref]]
    i: List[int] = field(default_factory=list)
    f: List[float] = field(default_factory=list)
    name: str = ""


@dataclass
class WgradOp:
    dz: Ref
    src0: Ref
    src1: Optional[Ref]
    dw: Ref
    N: int
    C0: int
    C1: int
    Kpad: int
    B: int
    T: int
    Fin: int
    Fz: int
    No: int
    ostride: int
    ophase: int
    istride: int
    dt: List[int]
    ioff: List[int]
    name: str = ""
    kind: int = OP_WGRAD
    dbias: Optional[Ref] = None          # bias gradient (column sums of dz over this launch's rows) rides along
    precision: int = 0                   # EAB_PREC_F32, or EAB_PREC_BF16 (operands rounded to bf16, fp32 accumulation)
    bf16_mask: int = 0                   # bit 0 / 1 / 2: dz / src0 / src1 is stored as bf16 (eab_wgrad_desc.bf16_mask)


class Slot:
    """Gradient accumulator of one tensor (shared by all views of it)."""

    def __init__(self):
        self.ref: Optional[Ref] = None
        self.owned = False


@dataclass
class TVar:
    """A materialised channels-last activation [B][T][F][C] of the training forward."""
    ref: Ref
    F: int
    C: int
    slot: Slot = field(default_factory=Slot)
    needs_grad: bool = True

    def view(self, F: int, C: int) -> "TVar":
        assert F * C == self.F * self.C
        return TVar(self.ref, F, C, self.slot, self.needs_grad)


def _split64(n: int) -> Tuple[int, int]:
    return n & 0xFFFFFFFF, n >> 32


class _LaneList(list):
    """op list that tags every appended op with the lane (stream) the lowering is currently emitting for"""

    def __init__(self, owner):
        super().__init__()
        self.owner = owner

    def append(self, op) -> None:
        op.lane = self.owner.cur_lane
        super().append(op)


class _Tape(list):
    """backward closures; each one is replayed on the lane that was current when it was recorded"""

    def __init__(self, owner):
        super().__init__()
        self.owner = owner

    def append(self, fn) -> None:
        owner, lane = self.owner, self.owner.cur_lane

        def run():
            prev, owner.cur_lane = owner.cur_lane, lane
            fn()
            owner.cur_lane = prev
        super().append(run)


class TrainLowering:
    spec_fn = staticmethod(param_specs)              # parameter inventory of the network being lowered (subclasses: GaGNet)
    supports = staticmethod(supported)

    def __init__(self, cfg, B: int, T: int, F: int = 161, precision: str = "f32"):
        if precision not in ("f32", "bf16"):
            raise ValueError("training precision is 'f32' or 'bf16' (bf16: forward, dgrad and wgrad contractions on the bf16 "
                             "matrix cores with fp32 accumulation -- the LSTM's recurrent products included; cell state, norms, "
                             "gradients and the optimiser state stay fp32)")
        self.prec = prg.PREC_CODE[precision]
        # bf16 programs: the LSTM's recurrent products on the bf16 matrix cores too (what torch.autocast does for nn.LSTM);
        # EAB_BF16_LSTM=0 keeps the fp32 recurrence of rounds 2-3
        self.lstm_prec = prg.PREC_BF16 if (self.prec == prg.PREC_BF16 and os.environ.get("EAB_BF16_LSTM", "1") != "0") else prg.PREC_F32
        # small-tile kernel (csrc/conv_st.hip) for the 1-D convolutions of the S-TCMs and their dgrads
        self.st = os.environ.get("EAB_ST", "1") != "0"
        if not self.supports(cfg):
            raise NotImplementedError("the HIP training path does not cover this configuration: " + (unsupported_reason(cfg) if self.supports is supported else "post-filter topology"))
        cfg.check_supported()
        self.cfg, self.B, self.T, self.F = cfg, B, T, F
        # (the BatchNorm buffers are not parameters of the programs: train mode normalises with batch statistics)
        self.specs = {k: s for k, s in self.spec_fn(cfg).items() if s.kind not in ("bn_count", "bn_mean", "bn_var")}
        # BatchNorm in train mode (NormSwitch BN branch, EaBNet.py:677-681; nn.BatchNorm{1,2}d): statistics per channel over
        # the whole batch.  Every tensor is [B][positions][C], channels-last and contiguous, so the batch is ONE virtual
        # utterance of B * positions rows: the InstanceNorm kernels (statistics merge, apply, both backward launches) run
        # unchanged with (B, P) -> (1, B * P).  bn_layers: (norm key, offset of its (mean, rstd) table, C, samples per channel)
        # cLN (the cumulative LayerNorm NormSwitch means to build, EaBNet.py:696-769): its own statistics / apply / backward ops
        self.cln = getattr(cfg, "norm_type", "IN") == "cLN"
        self.bn = getattr(cfg, "norm_type", "IN") == "BN"
        self.nB, self.nP = (1, B * T) if self.bn else (B, T)
        self.bn_layers: List[Tuple[str, int, int, int]] = []
        self.poff: Dict[str, int] = {}
        n = 0
        for k, s in self.specs.items():
            self.poff[k] = n
            n += int(np.prod(s.shape)) if s.shape else 1
        self.n_params = n
        self.a_size = 0
        self.allocs: List[Tuple[int, int]] = []                               # (offset, floats) of every activation-arena region
        # bf16 STORAGE of the tensors only bf16 contractions read (assign_bf16_storage); EAB_BF16_STORE=0: everything fp32
        self.bf16_store = os.environ.get("EAB_BF16_STORE", "1") != "0"
        self.bf16_tensors = 0
        self.w_imgs: List[Tuple[np.ndarray, Optional[np.ndarray]]] = []      # packed-parameter images (flat indices, -1 = 0)
        self.w_size = 0
        self.w_index: Dict[str, Ref] = {}
        self.g_imgs: List[Tuple[int, List[np.ndarray]]] = []                 # (offset, param-index images) of gradient entries
        self.g_size = 0
        self.cur_lane = 0                                                     # 0 = the caller's stream; > 0: parallel branches
        self.fwd: list = _LaneList(self)
        self.bwd: list = _LaneList(self)
        self.sync = {"fwd": {}, "bwd": {}}                                    # op index -> [("fork" | "join", lanes)]
        self.tape: List[Callable[[], None]] = _Tape(self)
        self.deferred: List[WgradOp] = []                                     # weight gradients, appended to bwd by finish()
        self.flops_fwd = 0
        self.flops_bwd = 0
        self.emit = self.fwd                                                  # the list ops are appended to
        self.gtaps: Dict[str, TVar] = {}                                      # named activations whose gradient tests read back

    def mark(self, which: str, what: str, lanes: Sequence[int]) -> None:
        """fork / join of parallel branches in front of the next op of the forward or backward list (executed by TrainBound.run
        with side streams and events; inside a hipGraph capture they become parallel branches)"""
        lst = self.fwd if which == "fwd" else self.bwd
        self.sync[which].setdefault(len(lst), []).append((what, list(lanes)))

    # ---- arenas ------------------------------------------------------------------------------------
    def alloc(self, nfloats: int) -> Ref:
        ref = Ref("a", self.a_size)
        self.allocs.append((self.a_size, nfloats))
        self.a_size += nfloats + ((-nfloats) % ALIGN)
        return ref

    def act(self, F: int, C: int, needs_grad: bool = True) -> TVar:
        return TVar(self.alloc(self.B * self.T * F * C), F, C, Slot(), needs_grad)

    def idx(self, key: str) -> np.ndarray:
        s = self.specs[key]
        return (self.poff[key] + np.arange(int(np.prod(s.shape)), dtype=np.int64)).reshape(s.shape)

    def wadd(self, name: str, img: np.ndarray, img2: Optional[np.ndarray] = None) -> Ref:
        """packed parameter operand: value[i] = flat[img[i]] (+ flat[img2[i]]), -1 = 0"""
        if name in self.w_index:
            return self.w_index[name]
        flat = np.ascontiguousarray(img, dtype=np.int64).reshape(-1)
        pad = (-flat.size) % ALIGN
        ref = Ref("w", self.w_size)
        f2 = None if img2 is None else np.ascontiguousarray(img2, dtype=np.int64).reshape(-1)
        self.w_imgs.append((np.concatenate([flat, np.full(pad, -1, np.int64)]),
                            None if f2 is None else np.concatenate([f2, np.full(pad, -1, np.int64)])))
        self.w_size += flat.size + pad
        self.w_index[name] = ref
        return ref

    def gadd(self, imgs: Sequence[np.ndarray]) -> Ref:
        """gradient entry shaped like imgs[0]; element i is the gradient of flat[img[i]] for every image (-1: discarded)"""
        flats = [np.ascontiguousarray(m, dtype=np.int64).reshape(-1) for m in imgs]
        n = flats[0].size
        assert all(f.size == n for f in flats)
        ref = Ref("g", self.g_size)
        self.g_imgs.append((self.g_size, flats))
        self.g_size += n + ((-n) % ALIGN)
        return ref

    def vec(self, key: str) -> Ref:
        return self.wadd(key, self.idx(key))

    def gvec(self, key: str) -> Ref:
        return self.gadd([self.idx(key)])

    # ---- gradient bookkeeping ------------------------------------------------------------------------
    def contribute(self, var: TVar, ref: Ref) -> None:
        """grad(var) += tensor at `ref` (same shape), emitted into the backward list"""
        if not var.needs_grad:
            return
        s = var.slot
        n = self.B * self.T * var.F * var.C
        if s.ref is None:
            s.ref, s.owned = ref, False                          # alias: nobody may write into it
            return
        dst = s.ref if s.owned else self.alloc(n)
        self.bwd.append(GenOp(OP_ADD, [s.ref, ref, dst], list(_split64(n)), name="grad+="))
        s.ref, s.owned = dst, True

    def grad_target(self, var: TVar) -> Tuple[Ref, Optional[Ref]]:
        """(dst, aux) for a kernel that can compute dst = contribution + aux (aux None: plain store)"""
        s = var.slot
        n = self.B * self.T * var.F * var.C
        if s.ref is None:
            s.ref, s.owned = self.alloc(n), True
            return s.ref, None
        if s.owned:
            return s.ref, s.ref
        old = s.ref
        s.ref, s.owned = self.alloc(n), True
        return s.ref, old

    def grad_of(self, var: TVar) -> Ref:
        assert var.slot.ref is not None, "backward reached a tensor nobody differentiated"
        return var.slot.ref

    # ---- convolution machinery --------------------------------------------------------------------------
    def pick_bm(self, No: int) -> int:
        return 128 if self.B * conv_tiles(self.T, No, 128) >= 2 * prg.CUS else 64

    @staticmethod
    def pack_taps_idx(w_nck: np.ndarray, taps_k: Sequence[int]) -> np.ndarray:
        N, Cc, _ = w_nck.shape
        upt = (Cc + 15) // 16
        out = np.full((N, len(taps_k), upt * 16), -1, dtype=np.int64)
        for j, k in enumerate(taps_k):
            out[:, j, :Cc] = w_nck[:, :, k]
        return out.reshape(N, -1)

    def conv_op(self, name, srcs: Sequence[TVar], w: Ref, bias: Optional[Ref], N: int, Kpad: int, Fin: int, Fout: int, No: int,
                ostride: int, ophase: int, istride: int, dt, ioff, epi: int, dst: Ref, Cout: int, stats=None, stat_tiles=0,
                stat_tile0=0, bm=None, aux=None, dst_acc=None, glu_dump=None, st: bool = False) -> prg.ConvOp:
        """st: `w` is in MFMA-fragment order and the launch goes to the small-tile kernel (csrc/conv_st.hip, KORDER_FRAG)"""
        s0, s1 = srcs[0], (srcs[1] if len(srcs) > 1 else None)
        bm = bm or self.pick_bm(No)
        op = prg.ConvOp(src0=s0.ref, src1=s1.ref if s1 else None, xf0=None, xf1=None, slope0=None, slope1=None,
                        C0=s0.C, C1=s1.C if s1 else 0, xf_mode=prg.XF_NONE, w=w, bias=bias, N=N, Kpad=Kpad, B=self.B, T=self.T,
                        Fin=Fin, Fout=Fout, No=No, ostride=ostride, ophase=ophase, istride=istride, dt=list(dt), ioff=list(ioff),
                        epi=epi, aux=aux, dst=dst, dst_acc=dst_acc, Cout=Cout, stats=stats, nsets=1 if stats else 0,
                        stat_slope0=None, stat_slope1=None, stat_tiles=stat_tiles, stat_tile0=stat_tile0, bm=bm, name=name)
        op.glu_dump = glu_dump
        if st:
            op.korder = prg.KORDER_FRAG
        if self.prec != prg.PREC_F32 and s0.ref.arena != "in" and s0.C % 4 == 0 and (s1 is None or s1.C % 4 == 0):
            op.precision = self.prec                  # the convolution on the raw network input stays exact
        self.emit.append(op)
        fl = 2 * self.B * self.T * No * N * len(dt) * (s0.C + (s1.C if s1 else 0))
        if self.emit is self.fwd:
            self.flops_fwd += fl
        else:
            self.flops_bwd += fl
        return op

    def wgrad_op(self, name, dz: Ref, N: int, Fz: int, srcs: Sequence[TVar], No, ostride, ophase, istride, dt, ioff, gimg,
                 dbias: Optional[Ref] = None) -> None:
        s0, s1 = srcs[0], (srcs[1] if len(srcs) > 1 else None)
        Ctot = s0.C + (s1.C if s1 else 0)
        Kpad = len(dt) * ((Ctot + 15) // 16) * 16
        assert tuple(gimg.shape) == (N, Kpad), (gimg.shape, N, Kpad)
        # Weight gradients are leaves of the backward graph (nothing reads dW before the step ends; their operands -- the
        # finished gradient of a convolution output and a forward activation -- are never written again, no buffer is
        # recycled), so they are all emitted at the END of the backward program, sorted by geometry: eab_run_program
        # then serves every run of identical geometry (the 18 S-TCMs, the repeated U-Net levels) with one launch.
        self.deferred.append(WgradOp(dz=dz, src0=s0.ref, src1=s1.ref if s1 else None, dw=self.gadd([gimg]), N=N, C0=s0.C,
                                     C1=s1.C if s1 else 0, Kpad=Kpad, B=self.B, T=self.T, Fin=s0.F, Fz=Fz, No=No,
                                     ostride=ostride, ophase=ophase, istride=istride, dt=list(dt), ioff=list(ioff), name=name,
                                     dbias=dbias, precision=self.prec if self.prec == prg.PREC_BF16 else prg.PREC_F32))
        self.flops_bwd += 2 * self.B * self.T * No * N * len(dt) * Ctot

    def colsum(self, name, x: Ref, rows: int, N: int, imgs: Sequence[np.ndarray]) -> None:
        self.bwd.append(GenOp(OP_COLSUM, [x, self.gadd(imgs)], list(_split64(rows)) + [N], name=name))

    def dgrad(self, name: str, var: TVar, dz: Ref, Kd: int, Fz: int, launches: Sequence[tuple], st_bm: int = 0) -> None:
        """grad(var) (+)= conv(dz; w) with the forward kernel on the gradient tensor dz [B][T][Fz][Kd]; `launches` =
        (w, No, ostride, ophase, istride, dt, ioff) per launch -- the launches of one call write disjoint output
        columns and share the accumulate operand."""
        if not var.needs_grad or not launches:
            return
        dst, aux = self.grad_target(var)
        src = TVar(dz, Fz, Kd)
        self.emit = self.bwd
        for k, (w, No, ostride, ophase, istride, dt, ioff) in enumerate(launches):
            self.conv_op(f"{name}.{k}", [src], w, None, var.C, len(dt) * ((Kd + 15) // 16) * 16, Fz, var.F, No, ostride, ophase,
                         istride, dt, ioff, prg.EPI_ADD if aux is not None else prg.EPI_LINEAR, dst, var.C, aux=aux,
                         st=st_bm > 0, bm=st_bm or None)
        self.emit = self.fwd

    # ---- norm + activation ---------------------------------------------------------------------------------
    def norm_act(self, name: str, raw: TVar, norm: str, act: str, mode: int, xf: Ref, mr: Ref, add: Optional[TVar] = None) -> TVar:
        """a = f(raw) [+ add]; records the backward (norm/PReLU gradients, dparams)"""
        out = self.act(raw.F, raw.C)
        P = self.nP * raw.F
        slp = self.vec(f"{act}.weight")
        # norm None: PReLU only (xf / mr NULL -- the kernels' no-norm form; the plain U-Net's middle encoder layers)
        gam, bet = (self.vec(f"{norm}.norm.weight"), self.vec(f"{norm}.norm.bias")) if norm else (None, None)
        self.fwd.append(GenOp(OP_TR_NORM_ACT, [raw.ref, xf, slp, add.ref if add else None, out.ref], [self.nB, P, raw.C, mode],
                              name=name))

        self.tape.append(self._norm_back(name, raw, out, add, norm, act, mode, mr, gam, bet, slp, P))
        return out

    def _norm_back(self, name, raw: TVar, out: TVar, add: Optional[TVar], norm: str, act: str, mode: int, mr: Ref, gam: Ref,
                   bet: Ref, slp: Ref, P: int) -> Callable[[], None]:
        def back():
            d = self.grad_of(out)
            if add is not None:
                self.contribute(add, d)
            dst, aux = self.grad_target(raw)
            # reduction scratch in the gradient arena: that arena is zero-filled once before every backward run, so the
            # kernel needs no zero-fill launch of its own (EAB_NB_SUMS_ZEROED)
            sums = Ref("g", self.g_size)
            self.g_size += NB_SUM_COPIES * self.nB * raw.C * 4 + ((-NB_SUM_COPIES * self.nB * raw.C * 4) % ALIGN)
            self.bwd.append(GenOp(OP_NORM_BWD, [d, raw.ref, mr, gam, bet, slp, sums, aux, dst,
                                                self.gvec(f"{norm}.norm.weight") if norm else None,
                                                self.gvec(f"{norm}.norm.bias") if norm else None, self.gvec(f"{act}.weight")],
                                  [self.nB, P, raw.C, mode | NB_SUMS_ZEROED], name=name + ".bwd"))
        return back

    def finalize(self, name: str, stats: Ref, C: int, tiles: int, count: int, norm: str) -> Tuple[Ref, Ref]:
        xf, mr = self.alloc(self.nB * C * 2), self.alloc(self.nB * C * 2)
        # BatchNorm in train mode: the partials are [b][tile][C][4] back to back, so the batch statistics are the same
        # merge over B * tiles partials of ONE virtual utterance (see __init__)
        bt = self.B // self.nB
        op = prg.FinalizeOp(stats=stats, B=self.nB, C=C, nsets=1, stat_tiles=tiles * bt, count=count * bt, eps=EPS_IN,
                            gamma0=self.vec(f"{norm}.norm.weight"), beta0=self.vec(f"{norm}.norm.bias"), xf0=xf, name=name)
        op.mr0 = mr
        self.fwd.append(op)
        self.note_bn(norm, mr, 0, C, count * bt)
        return xf, mr

    def cln_unit(self, name: str, raw: TVar, norm: str, act: str, mode: int, add: Optional[TVar] = None) -> TVar:
        """cumulative LayerNorm + PReLU (either order) [+ add] as statistics + apply ops (eab_cln_stats_f32 / eab_cln_apply_f32,
        whole utterance) and ONE backward op (eab_train_cln_bwd_f32: row sums, reverse scan over t, apply) followed by the
        column sum of its per-row parameter-gradient partials"""
        B, T, C = self.B, self.T, raw.C
        P = raw.F * C
        out = self.act(raw.F, C)
        sums, mr = self.alloc(B * T * 4), self.alloc(B * T * 2)              # [B][T][2] doubles | (cum_mean, rstd)
        gimg, bimg, simg = self.idx(f"{norm}.norm.gain").reshape(C), self.idx(f"{norm}.norm.bias").reshape(C), self.idx(f"{act}.weight")
        gain, bias, slp = self.wadd(f"{norm}.norm.gain", gimg), self.wadd(f"{norm}.norm.bias", bimg), self.vec(f"{act}.weight")
        self.fwd.append(GenOp(OP_CLN_STATS, [raw.ref, slp if mode == XF_PRELU_NORM else None, sums, None, mr], [B, T, P, C], [EPS_IN],
                              name=name + ".cln_stats"))
        self.fwd.append(GenOp(OP_CLN_APPLY, [raw.ref, mr, gain, bias, slp, add.ref if add else None, out.ref], [B, T, P, C, mode],
                              name=name + ".cln"))

        def back():
            d = self.grad_of(out)
            if add is not None:
                self.contribute(add, d)
            dst, aux = self.grad_target(raw)
            rs, ab, part = self.alloc(B * T * 4), self.alloc(B * T * 2), self.alloc(B * T * 3 * C)
            self.bwd.append(GenOp(OP_CLN_BWD, [d, raw.ref, mr, gain, bias, slp, rs, ab, part, aux, dst], [B, T, P, C, mode],
                                  name=name + ".cln_bwd"))
            self.colsum(name + ".cln_dparams", part, B * T, 3 * C, [np.concatenate([gimg, bimg, simg.reshape(C)])])
        self.tape.append(back)
        return out

    def note_bn(self, norm: str, mr: Ref, c0: int, C: int, n: int) -> None:
        """BatchNorm: where this layer's batch (mean, rstd) table lies, for the update of its running buffers"""
        if self.bn:
            self.bn_layers.append((norm, mr.off + 2 * c0, C, n))

    # ---- 2-D units -----------------------------------------------------------------------------------------
    def conv2d_fwd(self, name: str, srcs: Sequence[TVar], wkey: str, glu: bool, norm: str, act: str,
                   in_perm: Optional[np.ndarray] = None, add: Optional[TVar] = None) -> TVar:
        """Strided causal Conv2d (+GLU) + InstanceNorm + PReLU (GateConv2d EaBNet.py:434-460 / Conv2dunit :391-407)."""
        wkey = gate_key({f"{k}": 1 for k in self.specs}, wkey)
        wi = self.idx(f"{wkey}.weight")                                      # (N, Cin, kt, kf) flat indices
        N, Cin, kt, kf = wi.shape
        if in_perm is not None:
            wi = wi[:, in_perm]
        Fin = srcs[0].F
        Fout = (Fin - kf) // 2 + 1
        order = glu_row_order(N) if glu else np.arange(N)
        taps = [(a, c) for a in range(kt) for c in range(kf)]
        wimg = self.pack_taps_idx(wi.reshape(N, Cin, kt * kf)[order], [a * kf + c for a, c in taps])
        bimg = self.idx(f"{wkey}.bias")[order]
        Cout = N // 2 if glu else N
        raw = self.act(Fout, Cout)
        bm = self.pick_bm(Fout)
        tiles = conv_tiles(self.T, Fout, bm)
        stats = self.alloc(self.B * tiles * Cout * 4) if (norm and not self.cln) else None
        dump = self.alloc(self.B * self.T * Fout * N) if glu else None
        dts, ios = [a - (kt - 1) for a, _ in taps], [c for _, c in taps]
        self.conv_op(name, srcs, self.wadd(f"{wkey}.w", wimg), self.wadd(f"{wkey}.b", bimg), N, wimg.shape[1], Fin, Fout, Fout, 1, 0,
                     2, dts, ios, prg.EPI_GLU if glu else prg.EPI_LINEAR, raw.ref, Cout, stats, tiles if stats else 0, 0, bm,
                     glu_dump=dump)
        xf, mr = self.finalize(name + ".in", stats, Cout, tiles, self.T * Fout, norm) if stats is not None else (None, None)
        rows = self.B * self.T * Fout

        def back():
            dr = self.grad_of(raw)
            if glu:
                dz = self.alloc(rows * N)
                self.bwd.append(GenOp(OP_GLU_BWD, [dr, dump, dz], list(_split64(rows)) + [N], name=name + ".glu_bwd"))
            else:
                dz = dr
            self.wgrad_op(name + ".wgrad", dz, N, Fout, srcs, Fout, 1, 0, 2, dts, ios, wimg, dbias=self.gadd([bimg]))
            # dgrad: per source, per input-column parity p:  dx[t'][2o'+p] = sum_{a, c = p, p+2, ..} W[:, ci, a, c]^T dz[t'+(kt-1)-a][o' - (c-p)/2]
            c_lo = 0
            for s in srcs:
                launches = []
                for p in (0, 1):
                    tp = [(a, c) for a in range(kt) for c in range(p, kf, 2)]
                    No = (Fin - p + 1) // 2
                    if No <= 0 or not tp:
                        continue
                    wd = wi[order][:, c_lo:c_lo + s.C]                       # (N packed, C_s, kt, kf)
                    img = self.pack_taps_idx(np.ascontiguousarray(wd.transpose(1, 0, 2, 3)).reshape(s.C, N, kt * kf),
                                             [a * kf + c for a, c in tp])
                    launches.append((self.wadd(f"{wkey}.wd.{c_lo}.{p}", img), No, 2, p, 1,
                                     [(kt - 1) - a for a, _ in tp], [-(c - p) // 2 for _, c in tp]))
                self.dgrad(f"{name}.dgrad", s, dz, N, Fout, launches)
                c_lo += s.C
        self.tape.append(back)          # before the norm/PReLU closure on the tape = after it in the backward
        if self.cln and norm:
            return self.cln_unit(name + ".act", raw, norm, act, XF_NORM_PRELU, add)
        return self.norm_act(name + ".act", raw, norm, act, XF_NORM_PRELU, xf, mr, add)

    def conv2d_transposed(self, name: str, srcs: Sequence[TVar], wkey: str, glu: bool, norm: str, act: str,
                          add: Optional[TVar] = None) -> TVar:
        """ConvTranspose2d (+chomp, +GLU) + InstanceNorm + PReLU as two gather-form launches (program.py), EaBNet.py:463-490, 410-431."""
        wkey = gate_key({f"{k}": 1 for k in self.specs}, wkey)
        wi = self.idx(f"{wkey}.weight")                                      # (Cin, N, kt, kf)
        Cin, N, kt, kf = wi.shape
        assert Cin == sum(s.C for s in srcs)
        Fin = srcs[0].F
        Fout = (Fin - 1) * 2 + kf
        order = glu_row_order(N) if glu else np.arange(N)
        wn = np.ascontiguousarray(wi.transpose(1, 0, 2, 3)).reshape(N, Cin, kt * kf)[order]
        bimg = self.idx(f"{wkey}.bias")[order]
        bref = self.wadd(f"{wkey}.b", bimg)
        Cout = N // 2 if glu else N
        raw = self.act(Fout, Cout)
        Nos = [(Fout + 1) // 2, Fout // 2]
        bm = self.pick_bm(Nos[0])
        tiles = [conv_tiles(self.T, n, bm) for n in Nos]
        stats = None if self.cln else self.alloc(self.B * sum(tiles) * Cout * 4)
        dump = self.alloc(self.B * self.T * Fout * N) if glu else None
        phases = []
        for ph in (0, 1):
            tp = [(a, c) for a in range(kt) for c in range(ph, kf, 2)]
            wimg = self.pack_taps_idx(wn, [a * kf + c for a, c in tp])
            dts, ios = [-a for a, _ in tp], [-(c - ph) // 2 for _, c in tp]
            self.conv_op(f"{name}.ph{ph}", srcs, self.wadd(f"{wkey}.w.ph{ph}", wimg), bref, N, wimg.shape[1], Fin, Fout, Nos[ph], 2, ph,
                         1, dts, ios, prg.EPI_GLU if glu else prg.EPI_LINEAR, raw.ref, Cout, stats, sum(tiles) if stats else 0,
                         (0 if ph == 0 else tiles[0]) if stats else 0, bm, glu_dump=dump)
            phases.append((ph, dts, ios, wimg))
        xf, mr = self.finalize(name + ".in", stats, Cout, sum(tiles), self.T * Fout, norm) if stats is not None else (None, None)
        rows = self.B * self.T * Fout

        def back():
            dr = self.grad_of(raw)
            if glu:
                dz = self.alloc(rows * N)
                self.bwd.append(GenOp(OP_GLU_BWD, [dr, dump, dz], list(_split64(rows)) + [N], name=name + ".glu_bwd"))
            else:
                dz = dr
            gb = self.gadd([bimg])               # both phases add their rows' column sums
            for ph, dts, ios, wimg in phases:
                self.wgrad_op(f"{name}.wgrad{ph}", dz, N, Fout, srcs, Nos[ph], 2, ph, 1, dts, ios, wimg, dbias=gb)
            # dgrad: a strided convolution over dz:  dx[t'][f] = sum_{a, c} W[ci, :, a, c] dz[t'+a][2f + c]
            taps = [(a, c) for a in range(kt) for c in range(kf)]
            c_lo = 0
            for s in srcs:
                wd = wi[c_lo:c_lo + s.C][:, order]                           # (C_s, N packed, kt, kf)
                img = self.pack_taps_idx(wd.reshape(s.C, N, kt * kf), [a * kf + c for a, c in taps])
                self.dgrad(f"{name}.dgrad", s, dz, N, Fout, [(self.wadd(f"{wkey}.wd.{c_lo}", img), Fin, 1, 0, 2,
                                                              [a for a, _ in taps], [c for _, c in taps])])
                c_lo += s.C
        self.tape.append(back)
        if self.cln:
            return self.cln_unit(name + ".act", raw, norm, act, XF_NORM_PRELU, add)
        return self.norm_act(name + ".act", raw, norm, act, XF_NORM_PRELU, xf, mr, add)

    def unet_module(self, pre: str, srcs: Sequence[TVar], scale: int, transposed: bool, in_perm=None) -> TVar:
        """En_unet_module.forward (EaBNet.py:372-388): the residual add is fused into the last norm+PReLU."""
        if transposed:
            g = self.conv2d_transposed(f"{pre}.in_conv", srcs, f"{pre}.in_conv.0.conv.0", True, f"{pre}.in_conv.1", f"{pre}.in_conv.2")
        else:
            g = self.conv2d_fwd(f"{pre}.in_conv", srcs, f"{pre}.in_conv.0.conv.1", True, f"{pre}.in_conv.1", f"{pre}.in_conv.2", in_perm)
        y, downs = g, []
        for j in range(scale):
            q = f"{pre}.enco.{j}.conv"
            y = self.conv2d_fwd(q, [y], f"{q}.0", False, f"{q}.1", f"{q}.2")
            downs.append(y)
        for j in range(scale):
            q = f"{pre}.deco.{j}.deconv"
            if j == 0:
                ins = [y]
            elif self.cfg.intra_connect == "add":
                ins = [self.add_vars(f"{q}.skip_add", y, downs[-(j + 1)])]       # Skip_connect 'add' (EaBNet.py:499-500)
            else:
                ins = [y, downs[-(j + 1)]]
            y = self.conv2d_transposed(q, ins, f"{q}.0", False, f"{q}.1", f"{q}.2", add=g if j == scale - 1 else None)
        return y

    def add_vars(self, name: str, a: TVar, b: TVar) -> TVar:
        """s = a + b, materialised (the training programs materialise every operand of a backward op anyway); the gradient
        of s flows to both summands unchanged"""
        assert a.F == b.F and a.C == b.C
        out = self.act(a.F, a.C)
        n = self.B * self.T * a.F * a.C
        self.fwd.append(GenOp(OP_ADD, [a.ref, b.ref, out.ref], list(_split64(n)), name=name))

        def back():
            d = self.grad_of(out)
            self.contribute(a, d)
            self.contribute(b, d)
        self.tape.append(back)
        return out

    # ---- 1-D units (S-TCM, Linear) -------------------------------------------------------------------------
    def conv1d(self, name: str, src: TVar, wimg_nck: np.ndarray, dts: Sequence[int], bimg: Optional[np.ndarray], epi: int,
               aux: Optional[TVar] = None, dst_acc: Optional[Ref] = None, wname: str = "") -> Tuple[TVar, Callable[[Ref], None]]:
        """out[t] = sum_j W[:, :, j] src[t + dts[j]] (+ bias) with epilogue LINEAR / RELU / ADD(aux).  Returns the output
        and a function that, given the gradient w.r.t. the pre-epilogue sum, emits wgrad / dbias / dgrad."""
        N, Cc, K = wimg_nck.shape
        wimg = self.pack_taps_idx(wimg_nck, range(K))
        out = self.act(1, N)

        def st_geometry(n, c, kp, epi_=prg.EPI_LINEAR):
            """rows per tile of a small-tile launch (csrc/conv_st.hip), 0 = conv_gemm_kernel"""
            if not (self.st and self.prec in (prg.PREC_F32, prg.PREC_BF16) and n in (64, 128, 256) and c in (64, 128, 256)
                    and kp <= {64: 320, 128: 320, 256: 64}[n] and epi_ in (prg.EPI_LINEAR, prg.EPI_RELU, prg.EPI_ADD)):
                return 0
            return 32 if self.B * ((self.T + 31) // 32) >= 2 * prg.CUS else 16
        st_bm = st_geometry(N, Cc, wimg.shape[1], epi)
        wref = self.wadd(wname + (".wf" if st_bm else ".w"), prg.pack_frag(wimg) if st_bm else wimg)
        bref = self.wadd(wname + ".b", bimg) if bimg is not None else None
        self.conv_op(name, [src], wref, bref, N, wimg.shape[1], 1, 1, 1, 1, 0, 1, list(dts), [0] * K, epi, out.ref, N,
                     bm=st_bm or 64, aux=aux.ref if aux is not None else None, dst_acc=dst_acc, st=st_bm > 0)
        rows = self.B * self.T

        def back(dz: Ref):
            self.wgrad_op(name + ".wgrad", dz, N, 1, [src], 1, 1, 0, 1, list(dts), [0] * K, wimg,
                          dbias=self.gadd([bimg]) if bimg is not None else None)
            img = self.pack_taps_idx(np.ascontiguousarray(wimg_nck.transpose(1, 0, 2)), range(K))      # (Cc, N, K)
            d_bm = st_geometry(Cc, N, img.shape[1])
            self.dgrad(name + ".dgrad", src, dz, N, 1, [(self.wadd(wname + (".wdf" if d_bm else ".wd"), prg.pack_frag(img) if d_bm else img),
                                                         1, 1, 0, 1, [-d for d in dts], [0] * K)], st_bm=d_bm)
        return out, back

    def in1d(self, name: str, raw: TVar, norm: str, act: str) -> TVar:
        """prelu -> InstanceNorm1d (S-TCM order, EaBNet.py:545-547) as ONE launch: statistics, (xf, mr) and the normalised
        tensor (eab_train_in1d_f32); backward = the norm backward of the PRELU_NORM form"""
        if self.cln:
            return self.cln_unit(name, raw, norm, act, XF_PRELU_NORM)
        xf, mr = self.alloc(self.nB * raw.C * 2), self.alloc(self.nB * raw.C * 2)
        out = self.act(raw.F, raw.C)
        P = self.nP * raw.F
        gam, bet, slp = self.vec(f"{norm}.norm.weight"), self.vec(f"{norm}.norm.bias"), self.vec(f"{act}.weight")
        self.fwd.append(GenOp(OP_IN_STATS, [raw.ref, slp, gam, bet, xf, mr, out.ref], [self.nB, P, raw.C], [EPS_IN], name=name))
        self.note_bn(norm, mr, 0, raw.C, P)
        self.tape.append(self._norm_back(name, raw, out, None, norm, act, XF_PRELU_NORM, mr, gam, bet, slp, P))
        return out

    def in1d_multi(self, name: str, raw: TVar, norms: Sequence[str], acts: Sequence[str]) -> List[TVar]:
        """Two units prelu -> InstanceNorm1d on the SAME input (the branch norms of an S-TCM): one forward launch writing two
        contiguous tensors, one two-launch backward that sums both input gradients"""
        assert len(norms) == 2 and len(acts) == 2
        if self.cln:      # two separate units on the same input: their input gradients add up through the gradient slot
            return [self.cln_unit(f"{name}.{v}", raw, norms[v], acts[v], XF_PRELU_NORM) for v in range(2)]
        C = raw.C * 2
        P = self.nP * raw.F
        n = self.nB * P * raw.C
        xf, mr = self.alloc(self.nB * C * 2), self.alloc(self.nB * C * 2)
        for v, k in enumerate(norms):
            self.note_bn(k, mr, v * raw.C, raw.C, P)
        base = self.alloc(2 * n)                                            # view v at base + v * n
        outs = [TVar(Ref("a", base.off + v * n), raw.F, raw.C) for v in range(2)]
        img_g = np.concatenate([self.idx(f"{k}.norm.weight") for k in norms])
        img_b = np.concatenate([self.idx(f"{k}.norm.bias") for k in norms])
        img_s = np.concatenate([self.idx(f"{k}.weight") for k in acts])
        gam, bet, slp = self.wadd(name + ".gamma", img_g), self.wadd(name + ".beta", img_b), self.wadd(name + ".slope", img_s)
        self.fwd.append(GenOp(OP_IN_STATS, [raw.ref, slp, gam, bet, xf, mr, base], [self.nB, P, C, raw.C], [EPS_IN], name=name))

        def back():
            d0, d1 = self.grad_of(outs[0]), self.grad_of(outs[1])
            dst, aux = self.grad_target(raw)
            sums = Ref("g", self.g_size)                                     # zero-filled with the gradient arena
            self.g_size += NB_SUM_COPIES * self.nB * C * 4 + ((-NB_SUM_COPIES * self.nB * C * 4) % ALIGN)
            # (p[4], beta in the one-view form, carries the second view's gradient here: the PRELU_NORM backward never reads beta)
            self.bwd.append(GenOp(OP_NORM_BWD, [d0, raw.ref, mr, gam, d1, slp, sums, aux, dst, self.gadd([img_g]), self.gadd([img_b]),
                                                self.gadd([img_s])], [self.nB, P, C, XF_PRELU_NORM | NB_SUMS_ZEROED, raw.C],
                                  name=name + ".bwd"))
        self.tape.append(back)
        return outs

    def tcm(self, pre: str, x: TVar, dilation: int, x_acc: Optional[Ref], perm: np.ndarray) -> TVar:
        """SqueezedTCM.forward, EaBNet.py:572-578.  The closures go on the tape in forward order (they bind their
        operands late): in_conv | left norm | right norm | branch convs + gate | out norm | out_conv."""
        cfg = self.cfg
        kd = cfg.kd1
        n = self.B * self.T * cfg.cd1
        w_in = self.idx(f"{pre}.in_conv.weight")[:, perm, :]                 # (cd, D, 1)
        y, back_in = self.conv1d(f"{pre}.in_conv", x, w_in, [0], None, prg.EPI_LINEAR, wname=f"{pre}.in_conv")
        self.tape.append(lambda: back_in(self.grad_of(y)))
        # both branch norms read y: one launch normalises it twice (eab_train_in1d_multi_f32, two contiguous outputs) and one
        # two-launch backward sums the two input gradients into dy
        yL, yR = self.in1d_multi(f"{pre}.lr", y, [f"{pre}.left_conv.1", f"{pre}.right_conv.1"],
                                 [f"{pre}.left_conv.0", f"{pre}.right_conv.0"])
        span = (kd - 1) * dilation
        lead = span if cfg.is_causal else span // 2
        dts = [j * dilation - lead for j in range(kd)]
        a, back_l = self.conv1d(f"{pre}.left_conv", yL, self.idx(f"{pre}.left_conv.3.weight"), dts, None, prg.EPI_LINEAR,
                                wname=f"{pre}.left_conv")
        r, back_r = self.conv1d(f"{pre}.right_conv", yR, self.idx(f"{pre}.right_conv.3.weight"), dts, None, prg.EPI_LINEAR,
                                wname=f"{pre}.right_conv")
        z = self.act(1, cfg.cd1)
        self.fwd.append(GenOp(OP_GATE_FWD, [a.ref, r.ref, z.ref], list(_split64(n)), name=f"{pre}.gate"))

        def back_gate():
            da, dr = self.alloc(n), self.alloc(n)
            self.bwd.append(GenOp(OP_GATE_BWD, [self.grad_of(z), a.ref, r.ref, da, dr], list(_split64(n)), name=f"{pre}.gate_bwd"))
            back_l(da)
            back_r(dr)
        self.tape.append(back_gate)
        zo = self.in1d(f"{pre}.out", z, f"{pre}.out_conv.1", f"{pre}.out_conv.0")
        w_out = self.idx(f"{pre}.out_conv.2.weight")[perm]                   # (D, cd, 1), rows permuted
        out, back_out = self.conv1d(f"{pre}.out_conv", zo, w_out, [0], None, prg.EPI_ADD, aux=x, dst_acc=x_acc,
                                    wname=f"{pre}.out_conv")

        def back():
            d = self.grad_of(out)
            self.contribute(x, d)                                            # residual
            back_out(d)
        self.tape.append(back)
        return out

    # ---- whole network -----------------------------------------------------------------------------------------
    def build(self) -> "TrainProgram":
        cfg, B, T, F = self.cfg, self.B, self.T, self.F
        M, c = cfg.M, cfg.c
        x_in = TVar(Ref("in"), F, 2 * M, Slot(), needs_grad=False)
        mem = np.arange(2 * M)
        in_perm = (mem % 2) * M + mem // 2                                   # memory channel m*2+ri <- reference ri*M+m
        skips: List[TVar] = []
        x = x_in
        if cfg.is_u2:
            for i in range(4):
                x = self.unet_module(f"en.meta_unet_list.{i}", [x], 4 - i, False, in_perm if i == 0 else None)
                skips.append(x)
                self.gtaps[f"en.{i}"] = x
            x = self.conv2d_fwd("en.last_conv", [x], "en.last_conv.0.conv.1", True, "en.last_conv.1", "en.last_conv.2")
            skips.append(x)
            self.gtaps["en.4"] = x
        else:
            # UNet_Encoder (EaBNet.py:213-239): five gated convolutions; layers 1 and 2 have no norm in front of their PReLU
            from .spec import unet_encoder_layers
            for i, (_, _, _, has_norm) in enumerate(unet_encoder_layers(cfg)):
                q = f"en.unet_list.{i}"
                x = self.conv2d_fwd(q, [x], f"{q}.0.conv.1", True, f"{q}.1" if has_norm else None,
                                    f"{q}.2" if has_norm else f"{q}.1", in_perm if i == 0 else None)
                skips.append(x)
                self.gtaps[f"en.{i}"] = x
        Fb = x.F
        assert Fb * x.C == cfg.d_feat
        k = np.arange(cfg.d_feat)
        perm = (k % c) * Fb + k // c                                         # memory channel f*64+c <- reference c*4+f
        xt = x.view(1, cfg.d_feat)
        x_acc = self.act(1, cfg.d_feat)
        self.fwd.append(prg.MemsetOp(x_acc.ref, B * T * cfg.d_feat, name="stcns.acc0"))
        lasts = []
        for gi in range(cfg.q):
            for i in range(cfg.p):
                xt = self.tcm(f"stcns.{gi}.tcm_list.{i}", xt, 2 ** i, x_acc.ref if i == cfg.p - 1 else None, perm)
            lasts.append(xt)

        def back_acc():                                                      # x_acc = sum of the group outputs (EaBNet.py:101-105)
            d = self.grad_of(x_acc)
            for v in lasts:
                self.contribute(v, d)
        # the group outputs are produced before x_acc is complete, but every consumer of x_acc comes later:
        # running this first in the backward hands d(x_acc) to the three group outputs before their own closures
        self.tape.append(back_acc)
        x = x_acc.view(Fb, c)
        self.gtaps["stcns"] = x
        if cfg.is_u2:
            for i in range(4):
                x = self.unet_module(f"de.meta_unet_list.{i}", [x, skips[-(i + 1)]], i + 1, True)
                self.gtaps[f"de.{i}"] = x
            e = self.conv2d_transposed("de.last_conv", [x, skips[0]], "de.last_conv.0.conv.0", True, "de.last_conv.1",
                                       "de.last_conv.2")
        else:
            for i in range(5):                        # UNet_Decoder (EaBNet.py:297-328)
                q = f"de.unet_list.{i}"
                x = e = self.conv2d_transposed(q, [x, skips[-(i + 1)]], f"{q}.0.conv.0", True, f"{q}.1", f"{q}.2")
                self.gtaps[f"de.{i}"] = x
        assert e.F == F and e.C == 64
        self.gtaps["de.4"] = e

        rows = B * T * F
        if cfg.bf_type == "cnn" or cfg.topo_type == "miso":
            # ---- pointwise head (EaBNet.py:80-81,111-113): Conv2d(64 -> 2M, 1x1); output plane m*2+ri is the column order the
            # filter-and-sum kernels read (rows padded to one 64-column tile).  miso (EaBNet.py:78-79,118-125): Conv2d(64 -> 2),
            # one complex mask on microphone 0 = the same head with zero rows for the other microphones; the caller sums the
            # (B,2,T,F) result over frequency, as the reference does
            wk = self.idx("bf_map.weight").reshape(-1, 64)
            wc = np.full((MLP_LD, 64, 1), -1, np.int64)
            wc[:wk.shape[0]] = wk[:, :, None]
            bcimg = np.full(MLP_LD, -1, np.int64)
            bcimg[:wk.shape[0]] = self.idx("bf_map.bias")
            wcimg = self.pack_taps_idx(wc, [0])
            bw = self.act(F, MLP_LD)
            self.conv_op("bf_map", [e], self.wadd("bf_map.w", wcimg), self.wadd("bf_map.b", bcimg), MLP_LD, 64, F, F, F, 1, 0, 1,
                         [0], [0], prg.EPI_LINEAR, bw.ref, MLP_LD)
            self.fwd.append(GenOp(OP_FILTER_SUM, [bw.ref, Ref("in"), Ref("out")], [B, T, F, M, MLP_LD], name="filter_sum"))
            self.gtaps["bf_w"] = bw

            def back_head_cnn():
                dbw = self.alloc(rows * MLP_LD)
                bw.slot.ref = dbw
                self.bwd.append(GenOp(OP_FS_BWD, [Ref("dout"), Ref("in"), dbw], [B, T, F, M, MLP_LD], name="filter_sum.bwd"))
                self.wgrad_op("bf_map.wgrad", dbw, MLP_LD, F, [e], F, 1, 0, 1, [0], [0], wcimg, dbias=self.gadd([bcimg]))
                wcd = self.pack_taps_idx(np.ascontiguousarray(wc.transpose(1, 0, 2)), [0])      # (64 channels of e, 64 padded rows)
                self.dgrad("bf_map.dgrad", e, dbw, MLP_LD, F, [(self.wadd("bf_map.wd", wcd), F, 1, 0, 1, [0], [0])])
            self.tape.append(back_head_cnn)
            return self._finish_build()

        # ---- LSTM_BF (EaBNet.py:600-614), unfused
        x_ln, mr_ln = self.act(F, 64), self.alloc(rows * 2)
        lg, lb = self.vec("bf_map.norm.weight"), self.vec("bf_map.norm.bias")
        self.fwd.append(GenOp(OP_LN_FWD, [e.ref, lg, lb, x_ln.ref, mr_ln], list(_split64(rows)), [EPS_LN], name="bf_map.norm"))
        h_in, layers = x_ln, []
        for nm in ("rnn1", "rnn2"):
            p = f"bf_map.{nm}"
            wcat_img = np.concatenate([self.idx(f"{p}.weight_ih_l0"), self.idx(f"{p}.weight_hh_l0")], axis=1)
            wcat = self.wadd(f"{p}.wcat", wcat_img)
            bias = self.wadd(f"{p}.bias", self.idx(f"{p}.bias_ih_l0"), self.idx(f"{p}.bias_hh_l0"))
            h, gates = self.act(F, 64), self.alloc(rows * 5 * 64)
            self.fwd.append(GenOp(OP_LSTM_TRAIN, [h_in.ref, wcat, bias, h.ref, gates], [B, T, F, self.lstm_prec], name=p))
            self.flops_fwd += 2 * rows * 256 * 128
            layers.append((p, h_in, h, gates, wcat))
            self.gtaps[p] = h
            h_in = h
        h2 = h_in
        # w_dnn: Linear 64->64 + ReLU, Linear 64->2M (rows padded to one 64-column tile), then the filter-and-sum
        w1 = self.idx("bf_map.w_dnn.0.weight")[:, :, None]
        y1 = self.act(F, 64)
        w1img = self.pack_taps_idx(w1, [0])
        b1img = self.idx("bf_map.w_dnn.0.bias")
        self.conv_op("bf_map.w_dnn.0", [h2], self.wadd("w_dnn.0.w", w1img), self.wadd("w_dnn.0.b", b1img), 64, 64, F, F, F, 1, 0, 1,
                     [0], [0], prg.EPI_RELU, y1.ref, 64)
        w2 = np.full((MLP_LD, 64, 1), -1, np.int64)
        w2[:2 * M] = self.idx("bf_map.w_dnn.2.weight")[:, :, None]
        b2img = np.full(MLP_LD, -1, np.int64)
        b2img[:2 * M] = self.idx("bf_map.w_dnn.2.bias")
        w2img = self.pack_taps_idx(w2, [0])
        bw = self.act(F, MLP_LD)
        self.conv_op("bf_map.w_dnn.2", [y1], self.wadd("w_dnn.2.w", w2img), self.wadd("w_dnn.2.b", b2img), MLP_LD, 64, F, F, F, 1, 0,
                     1, [0], [0], prg.EPI_LINEAR, bw.ref, MLP_LD)
        self.fwd.append(GenOp(OP_FILTER_SUM, [bw.ref, Ref("in"), Ref("out")], [B, T, F, M, MLP_LD], name="filter_sum"))

        self.gtaps["bf_w"] = bw
        self.gtaps["bf_map.ln"] = x_ln
        self.gtaps["bf_map.y1"] = y1

        def back_head():
            dbw = self.alloc(rows * MLP_LD)
            bw.slot.ref = dbw
            self.bwd.append(GenOp(OP_FS_BWD, [Ref("dout"), Ref("in"), dbw], [B, T, F, M, MLP_LD], name="filter_sum.bwd"))
            self.wgrad_op("w_dnn.2.wgrad", dbw, MLP_LD, F, [y1], F, 1, 0, 1, [0], [0], w2img, dbias=self.gadd([b2img]))
            dy1 = self.alloc(rows * 64)
            src = TVar(dbw, F, MLP_LD)
            self.emit = self.bwd
            w2d = self.pack_taps_idx(np.ascontiguousarray(w2.transpose(1, 0, 2)), [0])          # (64 y1-channels, 64 padded rows)
            self.conv_op("w_dnn.2.dgrad", [src], self.wadd("w_dnn.2.wd", w2d), None, 64, 64, F, F, F, 1, 0, 1, [0], [0],
                         prg.EPI_LINEAR, dy1, 64)
            self.emit = self.fwd
            dpre = self.alloc(rows * 64)
            self.bwd.append(GenOp(OP_RELU_BWD, [dy1, y1.ref, dpre], list(_split64(rows * 64)), name="w_dnn.relu_bwd"))
            self.wgrad_op("w_dnn.0.wgrad", dpre, 64, F, [h2], F, 1, 0, 1, [0], [0], w1img, dbias=self.gadd([b1img]))
            w1d = self.pack_taps_idx(np.ascontiguousarray(w1.transpose(1, 0, 2)), [0])
            self.dgrad("w_dnn.0.dgrad", h2, dpre, 64, F, [(self.wadd("w_dnn.0.wd", w1d), F, 1, 0, 1, [0], [0])])
            for p, hin, h, gates, wcat in reversed(layers):
                dg = self.alloc(rows * 256)
                self.bwd.append(GenOp(OP_LSTM_BWD, [gates, self.grad_of(h), wcat, dg], [B, T, F, self.lstm_prec], name=p + ".bwd"))
                self.flops_bwd += 2 * rows * 256 * 64
                self.wgrad_op(p + ".wgrad_ih", dg, 256, F, [hin], F, 1, 0, 1, [0], [0], self.idx(f"{p}.weight_ih_l0"),
                              dbias=self.gadd([self.idx(f"{p}.bias_ih_l0"), self.idx(f"{p}.bias_hh_l0")]))
                self.wgrad_op(p + ".wgrad_hh", dg, 256, F, [h], F, 1, 0, 1, [-1], [0], self.idx(f"{p}.weight_hh_l0"))
                wih_d = self.pack_taps_idx(np.ascontiguousarray(self.idx(f"{p}.weight_ih_l0").T)[:, :, None], [0])   # (64, 256)
                self.dgrad(p + ".dgrad", hin, dg, 256, F, [(self.wadd(p + ".wd", wih_d), F, 1, 0, 1, [0], [0])])
            de = self.alloc(rows * 64)
            self.bwd.append(GenOp(OP_LN_BWD, [self.grad_of(x_ln), e.ref, mr_ln, lg, de, self.gvec("bf_map.norm.weight"),
                                              self.gvec("bf_map.norm.bias")], list(_split64(rows)), name="bf_map.norm.bwd"))
            self.contribute(e, de)
        self.tape.append(back_head)
        return self._finish_build()

    def _finish_build(self) -> "TrainProgram":
        # ---- backward: replay the tape in reverse
        for fn in reversed(self.tape):
            fn()
        return self.finish()

    def assign_bf16_storage(self) -> None:
        """bf16 programs: STORE as bf16 every tensor that nothing but bf16 contractions read (BASELINE configs[3]: "bf16").
        Those kernels round their operands to bf16 on the way into LDS, so a tensor whose only readers are the gather of a bf16
        convolution (forward or dgrad) and the bf16 weight-gradient kernel can be written in bf16 by its producer: the
        contractions see the same bits, the producer writes and the consumers read half the bytes, and the consumers convert
        nothing.  What qualifies in practice: the normalised activations between the 2-D units (producer tr_norm_act) and the
        gradients of the 2-D convolution outputs (producers: the apply pass of the norm backward, the GLU backward).  Everything
        else -- raw convolution outputs and their statistics, residual operands, gradients that accumulate, the S-TCN (small-
        tile kernel), LSTM, head -- stays fp32.  Decided on the finished op lists: a region of the activation arena becomes bf16
        iff it has exactly one writer, that writer can store bf16, and every reader is such a contraction; regions keep their
        size and offset (a bf16 tensor occupies the first half), so nothing else of the program changes."""
        import bisect
        starts = [a for a, _ in self.allocs]

        def region(ref: Optional[Ref]) -> Optional[int]:
            if ref is None or ref.arena != "a":
                return None
            k = bisect.bisect_right(starts, ref.off) - 1
            return k if k >= 0 and ref.off < starts[k] + max(self.allocs[k][1], 1) else None
        uses: Dict[int, list] = {}

        def note(ref, what, op, arg=None):
            k = region(ref)
            if k is not None:
                # a use that does not start at the region's first element is a partial view: never converted
                uses.setdefault(k, []).append((what if ref.off == starts[k] else "other", op, arg))

        def conv_reads_bf16(op) -> bool:
            return (op.precision == prg.PREC_BF16 and op.korder == prg.KORDER_TAP and op.xf_mode == prg.XF_NONE and op.Fin > 1
                    and op.C0 % 16 == 0 and op.C1 % 16 == 0 and op.epi != prg.EPI_DUALGATE and not getattr(op, "win", False)
                    and op.fin_stats is None)
        for op in list(self.fwd) + list(self.bwd) + list(self.deferred):
            if isinstance(op, prg.ConvOp):
                ok = conv_reads_bf16(op)
                note(op.src0, "read" if ok and op.C0 % 32 == 0 else "other", op, ("conv", 1))      # (walked in 32-channel units)
                note(op.src1, "read" if ok and op.C1 % 32 == 0 else "other", op, ("conv", 2))
                for f in ("aux", "dst", "dst_acc", "stats", "glu_dump", "xf0", "xf1", "fin_stats", "f2_dst", "f2_stats"):
                    note(getattr(op, f, None), "other", op)
            elif isinstance(op, WgradOp):
                ok = op.precision == prg.PREC_BF16
                note(op.dz, "read" if ok and op.N % 4 == 0 else "other", op, ("wgrad", 1))
                note(op.src0, "read" if ok and op.C0 % 16 == 0 else "other", op, ("wgrad", 2))
                note(op.src1, "read" if ok and op.C0 % 16 == 0 and op.C1 % 16 == 0 else "other", op, ("wgrad", 4))
            elif isinstance(op, GenOp):
                writer = {OP_TR_NORM_ACT: 4, OP_GLU_BWD: 2}.get(op.kind)
                if op.kind == OP_NORM_BWD and len(op.i) == 4 and op.p[7] is None:
                    writer = 8                                   # dx of the single-tensor norm backward, not accumulating
                for j, r in enumerate(op.p):
                    note(r, "write" if j == writer else "other", op, j)
            else:                                                # MemsetOp, FinalizeOp, ...: whatever they touch stays fp32
                for v in vars(op).values():
                    if isinstance(v, Ref):
                        note(v, "other", op)
        chosen = {k for k, lst in uses.items()
                  if "other" not in [w for w, _, _ in lst] and [w for w, _, _ in lst].count("write") == 1
                  and "read" in [w for w, _, _ in lst]}
        # the two sources of a concatenation are read by ONE launch: keep the weight-gradient kernel's operand loads uniform
        # (both sources bf16, or both fp32) -- drop a tensor whose partner in some two-source launch does not qualify
        while True:
            drop = set()
            for op in list(self.fwd) + list(self.bwd) + list(self.deferred):
                if isinstance(op, (prg.ConvOp, WgradOp)) and op.src1 is not None:
                    a, b = region(op.src0), region(op.src1)
                    if (a in chosen) != (b in chosen):
                        drop |= {a, b} & chosen
            if not drop:
                break
            chosen -= drop
        for k in sorted(chosen):
            lst = uses[k]
            self.bf16_tensors += 1
            for what, op, arg in lst:
                if what == "write":
                    if op.kind == OP_GLU_BWD:
                        op.i = list(op.i[:3]) + [STORE_BF16]
                    else:
                        op.i[3] |= STORE_BF16                    # (mode word of eab_train_norm_act_f32 / eab_train_norm_bwd_f32)
                elif arg[0] == "conv":
                    op.src_bf16 = getattr(op, "src_bf16", 0) | arg[1]
                else:
                    op.bf16_mask |= arg[1]

    def finish(self) -> "TrainProgram":
        ia = np.concatenate([a for a, _ in self.w_imgs]).astype(np.int32)
        has_b = any(b is not None for _, b in self.w_imgs)
        ib = np.concatenate([(b if b is not None else np.full(a.size, -1, np.int64)) for a, b in self.w_imgs]).astype(np.int32) \
            if has_b else None
        inv = np.full(self.n_params, -1, np.int64)
        for off, flats in self.g_imgs:
            for fl in flats:
                m = fl >= 0
                tgt = fl[m]
                assert (inv[tgt] == -1).all(), "a parameter element received two gradient entries"
                inv[tgt] = off + np.nonzero(m)[0]
        if self.prec == prg.PREC_BF16 and self.bf16_store:
            self.assign_bf16_storage()

        def geometry(o: WgradOp):
            return (o.N, o.C0, o.C1, o.Kpad, o.Fin, o.Fz, o.No, o.ostride, o.ophase, o.istride, tuple(o.dt), tuple(o.ioff),
                    o.src1 is None, o.dbias is None, o.precision, o.bf16_mask)
        order: Dict[tuple, int] = {}
        for o in self.deferred:
            order.setdefault(geometry(o), len(order))
        for o in self.deferred:
            o.lane = 0
        self.bwd.extend(sorted(self.deferred, key=lambda o: order[geometry(o)]))      # stable: first-seen geometry first
        self.deferred = []
        return TrainProgram(cfg=self.cfg, B=self.B, T=self.T, F=self.F, fwd=self.fwd, bwd=self.bwd, a_floats=self.a_size,
                            w_floats=self.w_size, g_floats=self.g_size, ia=ia, ib=ib, inv=inv.astype(np.int32),
                            n_params=self.n_params, keys=list(self.specs), shapes=[tuple(s.shape) for s in self.specs.values()],
                            flops_fwd=self.flops_fwd, flops_bwd=self.flops_bwd,
                            grad_taps={k: (v.slot.ref, v.F, v.C) for k, v in self.gtaps.items() if v.slot.ref is not None},
                            lanes={"fwd": [o.lane for o in self.fwd], "bwd": [o.lane for o in self.bwd]}, sync=self.sync,
                            bn_layers=list(self.bn_layers))


@dataclass
class TrainProgram:
    cfg: NetConfig
    B: int
    T: int
    F: int
    fwd: list
    bwd: list
    a_floats: int
    w_floats: int
    g_floats: int
    ia: np.ndarray
    ib: Optional[np.ndarray]
    inv: np.ndarray
    n_params: int
    keys: List[str]
    shapes: List[tuple]
    flops_fwd: int = 0
    flops_bwd: int = 0
    grad_taps: Dict[str, tuple] = field(default_factory=dict)       # name -> (Ref of d loss / d activation, F, C)
    out_shape: Optional[tuple] = None        # shape of the 'out' / 'dout' arenas (default: the beam-former's (B, 2, T, F))
    has_in2: bool = False                    # a second input arena 'in2' (the post-filter's previous estimate)
    lanes: Dict[str, list] = field(default_factory=dict)          # per program: lane of every op (0 = the caller's stream)
    sync: Dict[str, dict] = field(default_factory=dict)           # per program: op index -> [("fork" | "join", lanes)]
    # BatchNorm (train mode): (norm key, offset of the layer's batch (mean, rstd) table in the activation arena, C, samples)
    bn_layers: List[tuple] = field(default_factory=list)


def lower_train(cfg: NetConfig, B: int, T: int, F: int = 161, precision: str = "f32") -> TrainProgram:
    return TrainLowering(cfg, B, T, F, precision).build()


# ----------------------------------------------------------------------------------------------------------------
# binding and execution
# ----------------------------------------------------------------------------------------------------------------
class TrainBound:
    """Device arenas + the two ctypes op arrays of a lowered training program."""

    def __init__(self, prog: TrainProgram, device: torch.device):
        self.prog, self.device = prog, device
        self.acts = torch.empty(max(prog.a_floats, 1), dtype=torch.float32, device=device)
        self.w = torch.empty(max(prog.w_floats, 1), dtype=torch.float32, device=device)
        self.g = torch.empty(max(prog.g_floats, 1), dtype=torch.float32, device=device)
        self.ia = torch.from_numpy(prog.ia).to(device)
        self.ib = torch.from_numpy(prog.ib).to(device) if prog.ib is not None else None
        self.inv = torch.from_numpy(prog.inv).to(device)
        self.fwd = (_lib.Op * len(prog.fwd))()
        self.bwd = (_lib.Op * len(prog.bwd))()
        self._bound = None
        self.serial = 0                     # forward passes run so far: a backward must belong to the latest one
        self.use_graph = True
        self.graphs = None                  # (forward hipGraph, backward hipGraph) on the static boundary buffers
        self.graph_failed = False
        self.static_x = self.static_x2 = self.static_out = self.static_dout = None
        self._direct = {}                   # which -> graphs.LaneGraphs for direct (uncaptured) multi-lane runs
        self.parallel_branches = os.environ.get("EAB_TRAIN_BRANCHES", "1") != "0"      # A/B knob

    def capture(self, x_shape) -> bool:
        """Both programs as hipGraphs on static boundary buffers (input, output, output gradient): a step is two graph
        launches instead of ~840 host-side kernel launches.  False (direct launches) if the runtime refuses."""
        if self.graphs is not None or self.graph_failed or not self.use_graph:
            return self.graphs is not None
        prog = self.prog
        try:
            self.static_x = torch.zeros(x_shape, dtype=torch.float32, device=self.device)
            self.static_x2 = torch.zeros(x_shape, dtype=torch.float32, device=self.device) if prog.has_in2 else None
            self.static_out = torch.zeros(prog.out_shape or (prog.B, 2, prog.T, prog.F), dtype=torch.float32, device=self.device)
            self.static_dout = torch.zeros_like(self.static_out)
            self.bind(self.static_x.data_ptr(), self.static_out.data_ptr(), self.static_dout.data_ptr(),
                      self.static_x2.data_ptr() if prog.has_in2 else None)
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                 # warm-up outside the capture (weights are packed already)
                self.g.zero_()
                self._launch("fwd", side.cuda_stream, 0, len(self.prog.fwd))       # (program order on one stream)
                self._launch("bwd", side.cuda_stream, 0, len(self.prog.bwd))
            torch.cuda.current_stream().wait_stream(side)
            # one single-stream hipGraph per lane segment (graphs.py: a hipGraph with internal branches can crash the HIP
            # runtime at replay); programs without parallel branches are one graph each
            graphs = []
            for which in ("fwd", "bwd"):
                lg = LaneGraphs(self.device, self._plan(which), lambda st, first, count, w=which: self._launch(w, st, first, count))
                lg.capture()
                graphs.append(lg)
            self.graphs = tuple(graphs)
        except Exception as e:                            # noqa: BLE001 - any capture failure -> direct launches
            import warnings
            warnings.warn(f"eabnet_amd: hipGraph capture of the training programs failed ({e!r}); using direct launches")
            self.graphs, self.graph_failed, self._bound = None, True, None
        return self.graphs is not None

    def bind(self, in_ptr: int, out_ptr: int, dout_ptr: int, in2_ptr: Optional[int] = None) -> None:
        if self._bound == (in_ptr, out_ptr, dout_ptr, in2_ptr):
            return
        bases = {"a": self.acts.data_ptr(), "w": self.w.data_ptr(), "g": self.g.data_ptr(), "in": in_ptr, "out": out_ptr,
                 "dout": dout_ptr, "in2": in2_ptr}

        def A(r):
            return None if r is None else bases[r.arena] + 4 * r.off
        for ops, arr in ((self.prog.fwd, self.fwd), (self.prog.bwd, self.bwd)):
            for k, op in enumerate(ops):
                o = arr[k]
                o.kind = op.kind
                if op.kind == prg.OP_CONV:
                    d = o.conv
                    for f in ("src0", "src1", "xf0", "xf1", "slope0", "slope1", "w", "bias", "aux", "dst", "dst_acc", "stats",
                              "stat_slope0", "stat_slope1", "fin_stats", "fin_gamma0", "fin_beta0", "fin_gamma1", "fin_beta1"):
                        setattr(d, f, A(getattr(op, f)))
                    d.glu_dump = A(getattr(op, "glu_dump", None))
                    d.src_bf16 = int(getattr(op, "src_bf16", 0))
                    for f in ("C0", "C1", "xf_mode", "N", "Kpad", "B", "T", "Fin", "Fout", "No", "ostride", "ophase", "istride",
                              "epi", "Cout", "nsets", "stat_tiles", "stat_tile0", "bm", "fin_tiles", "fin_nsets", "fin_count",
                              "precision", "korder"):
                        setattr(d, f, int(getattr(op, f)))
                    d.fin_eps = float(op.fin_eps)
                    d.ntaps = len(op.dt)
                    for j in range(_lib.MAX_TAPS):
                        d.dt[j] = op.dt[j] if j < len(op.dt) else 0
                        d.ioff[j] = op.ioff[j] if j < len(op.ioff) else 0
                elif op.kind == prg.OP_IN_FINALIZE:
                    o.i[0:5] = [op.B, op.C, op.nsets, op.stat_tiles, op.count]
                    o.f[0] = op.eps
                    for j, r in enumerate((op.stats, op.gamma0, op.beta0, op.xf0, op.gamma1, op.beta1, op.xf1,
                                           getattr(op, "mr0", None), None)):
                        o.p[j] = A(r)
                elif op.kind == prg.OP_MEMSET0:
                    nbytes = 4 * op.nfloats
                    o.i[0] = C.c_int32(nbytes & 0xFFFFFFFF).value
                    o.i[1] = nbytes >> 32
                    o.p[0] = A(op.ptr)
                elif op.kind == OP_WGRAD:
                    d = o.wgrad
                    d.dz, d.src0, d.src1, d.dw, d.dbias = A(op.dz), A(op.src0), A(op.src1), A(op.dw), A(op.dbias)
                    for f in ("N", "C0", "C1", "Kpad", "B", "T", "Fin", "Fz", "No", "ostride", "ophase", "istride"):
                        setattr(d, f, int(getattr(op, f)))
                    d.ntaps = len(op.dt)
                    d.precision = int(op.precision)
                    d.bf16_mask = int(op.bf16_mask)
                    for j in range(_lib.MAX_TAPS):
                        d.dt[j] = op.dt[j] if j < len(op.dt) else 0
                        d.ioff[j] = op.ioff[j] if j < len(op.ioff) else 0
                else:
                    for j, r in enumerate(op.p):
                        o.p[j] = A(r)
                    for j, v in enumerate(op.i):
                        o.i[j] = C.c_int32(int(v) & 0xFFFFFFFF).value if v > 0x7FFFFFFF else int(v)
                    for j, v in enumerate(op.f):
                        o.f[j] = float(v)
        self._bound = (in_ptr, out_ptr, dout_ptr, in2_ptr)

    def _plan(self, which: str) -> list:
        from .model import graph_branches_allowed
        n = len(self.prog.fwd if which == "fwd" else self.prog.bwd)
        sync = self.prog.sync.get(which) if (self.parallel_branches and graph_branches_allowed()) else None
        return plan_segments(n, self.prog.lanes[which], sync) if sync else single_lane(n)

    def _launch(self, which: str, stream: int, first: int, n: int) -> None:
        arr = self.fwd if which == "fwd" else self.bwd
        ops = C.cast(C.byref(arr, first * C.sizeof(_lib.Op)), C.POINTER(_lib.Op))
        _lib.check(_lib.load().eab_run_program(ops, n, C.c_void_p(stream)), f"eab_run_program({which})")

    def run(self, which: str, stream: int, first: int = 0, count: Optional[int] = None) -> None:
        """Direct launches of ops [first, first + count) of one program.  Whole programs with parallel branches (the
        post-filter's three S-TCM chains) fork onto side streams with events, exactly as their captured form replays."""
        arr = self.fwd if which == "fwd" else self.bwd
        if first == 0 and count is None:
            plan = self._plan(which)
            if len(plan) > 1:
                assert torch.cuda.current_stream().cuda_stream == stream, "multi-lane programs run on torch's current stream"
                if which not in self._direct:
                    self._direct[which] = LaneGraphs(self.device, plan, lambda st, f, c, w=which: self._launch(w, st, f, c))
                self._direct[which].run_direct()
                return
        self._launch(which, stream, first, len(arr) - first if count is None else count)

    def update_bn_buffers(self, module, momentum: float = 0.1) -> None:
        """nn.BatchNorm's train-mode side effect (NormSwitch BN branch, EaBNet.py:677-681): running_mean / running_var move
        by `momentum` towards the batch mean / UNBIASED batch variance, num_batches_tracked counts the step.  The batch
        statistics are read back from the (mean, rstd) tables the forward program just wrote -- a handful of torch
        foreach launches on the caller's stream for all norms of the network."""
        L = self.prog.bn_layers
        if not L:
            return
        if getattr(self, "_bn_idx", None) is None:
            idx = np.concatenate([off + 2 * np.arange(C, dtype=np.int64) for _, off, C, _ in L])
            unb = np.concatenate([np.full(C, n / max(n - 1, 1), np.float32) for _, _, C, n in L])
            self._bn_idx = torch.from_numpy(idx).to(self.acts.device)
            self._bn_unb = torch.from_numpy(unb).to(self.acts.device)
            self._bn_split = [C for _, _, C, _ in L]
        with torch.no_grad():
            mean = self.acts[self._bn_idx]
            rstd = self.acts[self._bn_idx + 1]
            var = (1.0 / (rstd * rstd) - EPS_IN).clamp_min_(0.0) * self._bn_unb
            rm = [module.get_buffer(f"{k}.norm.running_mean") for k, *_ in L]
            rv = [module.get_buffer(f"{k}.norm.running_var") for k, *_ in L]
            torch._foreach_lerp_(rm, list(mean.split(self._bn_split)), momentum)
            torch._foreach_lerp_(rv, list(var.split(self._bn_split)), momentum)
            torch._foreach_add_([module.get_buffer(f"{k}.norm.num_batches_tracked") for k, *_ in L], 1)

    def pack(self, flat: torch.Tensor, stream: int) -> None:
        _lib.check(_lib.load().eab_gather_f32(flat.data_ptr(), self.ia.data_ptr(), self.ib.data_ptr() if self.ib is not None else None,
                                              self.w.data_ptr(), self.prog.w_floats, C.c_void_p(stream)), "eab_gather_f32")

    def unpack_grads(self, gflat: torch.Tensor, stream: int) -> None:
        _lib.check(_lib.load().eab_gather_f32(self.g.data_ptr(), self.inv.data_ptr(), None, gflat.data_ptr(), self.prog.n_params,
                                              C.c_void_p(stream)), "eab_gather_f32(grads)")


class _EaBNetTrainFn(torch.autograd.Function):
    """One autograd node for the whole network: forward program, backward program."""

    @staticmethod
    def forward(ctx, bound: TrainBound, sync_group, x: torch.Tensor, *params: torch.Tensor) -> torch.Tensor:
        prog = bound.prog
        ctx.sync_group = sync_group
        st = torch.cuda.current_stream().cuda_stream
        flat = torch.cat([p.detach().reshape(-1) for p in params])      # (one batched copy; fp32 parameters)
        if flat.dtype != torch.float32:
            flat = flat.to(torch.float32)
        bound.pack(flat, st)
        bound.serial += 1
        ctx.serial = bound.serial
        if bound.capture(tuple(x.shape)):
            bound.static_x.copy_(x)
            bound.graphs[0].replay()
            out = bound.static_out.clone()
            ctx.bound, ctx.x, ctx.dout, ctx.out = bound, None, None, None
        else:
            out = torch.empty((prog.B, 2, prog.T, prog.F), dtype=torch.float32, device=x.device)
            dout = torch.empty_like(out)
            bound.bind(x.data_ptr(), out.data_ptr(), dout.data_ptr())
            bound.run("fwd", st)
            ctx.bound, ctx.x, ctx.dout, ctx.out = bound, x, dout, out
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
        if ctx.x is None:
            bound.static_dout.copy_(grad_out)
            bound.graphs[1].replay()
        else:
            ctx.dout.copy_(grad_out.to(torch.float32))
            bound.bind(ctx.x.data_ptr(), ctx.out.data_ptr(), ctx.dout.data_ptr())
            bound.run("bwd", st)
        gflat = torch.empty(prog.n_params, dtype=torch.float32, device=bound.device)     # fresh per call: .grad may keep views of it
        bound.unpack_grads(gflat, st)
        grads = finish_flat_gradient(gflat, ctx.sync_group, ctx.shapes, ctx.dtypes, ctx.needs_input_grad[3:])
        return (None, None, None, *grads)


def finish_flat_gradient(gflat: torch.Tensor, sync_group, shapes, dtypes, needs) -> list:
    """What a training autograd node does with the flat gradient its backward program produced: average it over the ranks
    (ONE all-reduce of the contiguous buffer -- the reference's DDP bucket all-reduce, train_distributed.py:198,228, without
    per-parameter hooks or bucket copies; sync_group None = no synchronisation, True = the default process group) and hand it
    back as one view per parameter (None for parameters that need no gradient).  Device-agnostic: tests/test_dist_gloo.py runs
    it on two gloo ranks."""
    if sync_group is not None:
        import torch.distributed as td
        grp = sync_group if sync_group is not True else None
        td.all_reduce(gflat, group=grp)
        gflat.div_(td.get_world_size(grp))
    # one split call + a view per parameter (a Python loop over ~500-800 parameters is on the step's critical path)
    sizes = [int(np.prod(shp)) if len(shp) else 1 for shp in shapes]
    grads = [g.view(shp) if dt == torch.float32 else g.view(shp).to(dt) for g, shp, dt in zip(gflat.split(sizes), shapes, dtypes)]
    if not all(needs):
        grads = [g if need else None for g, need in zip(grads, needs)]
    return grads


def forward_train(module, inpt: torch.Tensor) -> torch.Tensor:
    """EaBNet.forward under autograd on the HIP training programs.  The gradient w.r.t. the input spectrogram is not
    produced (the reference's training never needs it)."""
    _lib.load()
    B, T, F, M, _ = inpt.shape
    x = inpt.detach().to(torch.float32).contiguous()
    cache = module.__dict__.setdefault("_train_bound", {})
    prec = "bf16" if module.precision == "bf16" else "f32"
    key = (B, T, F, str(x.device), prec)
    bound = cache.pop(key, None)
    if bound is None:
        # a small LRU of bound programs (variable-length batches, a smaller last batch, alternating train / validation
        # shapes): re-lowering and re-capturing two hipGraphs on every shape change costs seconds
        while len(cache) >= TRAIN_BOUND_CACHE:
            torch.cuda.synchronize(x.device)      # the dropped program's arenas may still be read by kernels in flight
            cache.pop(next(iter(cache)))
        with torch.cuda.device(x.device):
            bound = TrainBound(lower_train(module.cfg, B, T, F, prec), x.device)
    cache[key] = bound                               # most recently used last
    bound.use_graph = bool(getattr(module, "use_graph", True)) and not torch.cuda.is_current_stream_capturing()
    sd = dict(module.named_parameters())
    params = [sd[k] for k in bound.prog.keys]
    sync = module.__dict__.get("grad_allreduce", None)          # None | True (default group) | a process group
    with torch.cuda.device(x.device):
        out = _EaBNetTrainFn.apply(bound, sync, x, *params)
        bound.update_bn_buffers(module)
    return out.to(inpt.dtype)


def enable_flat_allreduce(module, group=True) -> None:
    """Data-parallel training without a DistributedDataParallel wrapper: every backward of ``module`` (an
    eabnet_amd.EaBNet on the HIP training path) all-reduces its flat gradient once and averages it over the ranks.
    Parameters must start identical on all ranks (broadcast them once, as DDP's constructor does)."""
    module.grad_allreduce = group


def broadcast_parameters(module, src: int = 0) -> None:
    import torch.distributed as td
    if td.is_available() and td.is_initialized():
        for t in list(module.parameters()) + list(module.buffers()):
            td.broadcast(t.data, src)
