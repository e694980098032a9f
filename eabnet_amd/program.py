"""Lowering of EaBNet.forward (reference EaBNet.py:88-117) to the op program
that libeabnet_hip.so executes (include/eabnet_hip.h, eab_run_program).

Everything here is host logic on numpy: weight re-packing, convolution
geometry (taps, phases of the transposed convolutions), the layer schedule and
the workspace plan.  Pointers are symbolic (arena, float offset) until
``bind()`` turns them into device addresses, so the whole lowering can be
checked on a machine without a GPU (tests/test_program_emulated.py interprets
the same program with numpy against the oracle).

Data layout: activations are channels-last
``[B][T][F][C]`` fp32.  Consequences folded into the packed weights:
  * network input (B,T,F,M,2) is read in place as C = 2M channels with memory
    channel m*2+ri, while the reference feeds channel ri*M+m (EaBNet.py:96-97)
    -> the first conv's input channels are permuted;
  * the bottleneck (B,256,T) of EaBNet.py:100 has channel c*4+f; here it is the
    same memory as the encoder output [B][T][4][64], i.e. channel f*64+c
    -> S-TCM in_conv columns / out_conv rows are permuted, and the two
    transposes of EaBNet.py:100,106 disappear.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .spec import GagConfig, NetConfig, gag_param_specs, gate_key, param_specs, unet_decoder_layers, unet_encoder_layers

# mirrors of the C enums (include/eabnet_hip.h)
XF_NONE, XF_NORM_PRELU, XF_PRELU_NORM = 0, 1, 2
EPI_LINEAR, EPI_GLU, EPI_RELU, EPI_MULSIG, EPI_ADD, EPI_DUALGATE, EPI_PHASE2 = 0, 1, 2, 3, 4, 5, 6
OP_CONV, OP_IN_FINALIZE, OP_NORM_ACT, OP_LSTM64, OP_BFW_FS, OP_MEMSET0, OP_GAG_PACK, OP_GAG_CRM = 1, 2, 3, 4, 5, 6, 7, 8
OP_CLN_STATS, OP_CLN_APPLY, OP_GATE_ROWS = 33, 34, 35
OP_CLN_STEP = 38         # run-time only (model._Bound): statistics + apply of a cLN unit for a one-frame streaming step in one launch
OP_CONV_CHAIN = 9        # run-time only (model._Bound): a run of small-tile launches executed by one launch (csrc/conv_st.hip)
ACT_SIGMOID, ACT_TANH, ACT_RELU = 0, 1, 2
GAG_PRE_LD = 324   # floats per (b, t) row of the interleaved previous estimate: 2*161 rounded up to a float4
GAG_LIN_LD = 192   # 161 linear outputs padded to three 64-column tiles
PREC_F32, PREC_F16X3, PREC_BF16 = 0, 1, 2
PREC_CODE = {"f32": PREC_F32, "f16x3": PREC_F16X3, "bf16": PREC_BF16}
KORDER_TAP, KORDER_CHUNK, KORDER_FRAG = 0, 1, 2
# deepest gated launch sent to the small-tile kernel (csrc/conv_st.hip takes up to 1024, weights in passes of 320): the
# 768-deep first phase of the last decoder measured slower there than on 128-row tiles (C1: 2.89 vs 2.75 ms per utterance)
ST_GLU_KMAX = 512
PATCH_MAX = 352    # CG_PMAX in csrc/conv_gemm.hip
MAX_TAPS = 16
EPS_IN = 1e-5      # nn.InstanceNorm*d default (reference EaBNet.py:684,686)
EPS_LN = 1e-5      # nn.LayerNorm default (reference EaBNet.py:598)
ALIGN = 64         # floats: every arena allocation starts on a 256-byte boundary
CUS = 256


@dataclass(frozen=True)
class Ref:
    """Symbolic device pointer: float offset into one of the arenas
    'w' (packed weights), 'a' (activations/workspace), 'in', 'in2' (second network input), 'out'."""
    arena: str
    off: int = 0


@dataclass
class Act:
    """A channels-last activation [B][T][F][C].  If ``xf`` is set the tensor in
    memory is RAW (pre-norm) and consumers apply f = (scale,shift) table +
    PReLU slope in mode ``mode`` while loading it."""
    ref: Ref
    F: int
    C: int
    xf: Optional[Ref] = None
    slope: Optional[Ref] = None
    mode: int = XF_NONE
    raw: bool = False          # un-normalised network input: consumed on exact fp32 in every precision mode


@dataclass
class ConvOp:
    src0: Ref
    src1: Optional[Ref]
    xf0: Optional[Ref]
    xf1: Optional[Ref]
    slope0: Optional[Ref]
    slope1: Optional[Ref]
    C0: int
    C1: int
    xf_mode: int
    w: Ref
    bias: Optional[Ref]
    N: int
    Kpad: int
    B: int
    T: int
    Fin: int
    Fout: int
    No: int
    ostride: int
    ophase: int
    istride: int
    dt: List[int]
    ioff: List[int]
    epi: int
    aux: Optional[Ref]
    dst: Ref
    dst_acc: Optional[Ref]
    Cout: int
    stats: Optional[Ref]
    nsets: int
    stat_slope0: Optional[Ref]
    stat_slope1: Optional[Ref]
    stat_tiles: int
    stat_tile0: int
    bm: int
    fin_stats: Optional[Ref] = None      # in-kernel InstanceNorm finalisation (few producer tiles)
    fin_gamma0: Optional[Ref] = None
    fin_beta0: Optional[Ref] = None
    fin_gamma1: Optional[Ref] = None
    fin_beta1: Optional[Ref] = None
    fin_tiles: int = 0
    fin_nsets: int = 0
    fin_count: int = 0
    fin_eps: float = 0.0
    precision: int = PREC_F32
    korder: int = KORDER_TAP
    win: bool = False                    # streaming: only the time rows of the current chunk
    name: str = ""
    kind: int = OP_CONV
    fz_counter: Optional[Ref] = None     # fused InstanceNorm finalisation by the last-arriving tile (eab_conv_desc.fz_*)
    fz_gamma0: Optional[Ref] = None
    fz_beta0: Optional[Ref] = None
    fz_xf0: Optional[Ref] = None
    fz_gamma1: Optional[Ref] = None
    fz_beta1: Optional[Ref] = None
    fz_xf1: Optional[Ref] = None
    fz_eps: float = 0.0
    # KORDER_FRAG (small-tile kernel): second output-column phase of a transposed convolution in the same launch
    ph1_w: Optional[Ref] = None
    ph1_No: int = 0
    ph1_ophase: int = 0
    ph1_Kpad: int = 0
    ph1_dt: List[int] = field(default_factory=list)
    ph1_ioff: List[int] = field(default_factory=list)
    # KORDER_FRAG, N = 256, 1-D: fused second 1x1 convolution on this launch's output rows (eab_conv_desc.f2_*)
    f2_w: Optional[Ref] = None
    f2_dst: Optional[Ref] = None
    f2_stats: Optional[Ref] = None
    f2_stat_slope0: Optional[Ref] = None
    f2_stat_slope1: Optional[Ref] = None
    f2_N: int = 0
    f2_nsets: int = 0
    f2_stat_tiles: int = 0
    # EPI_PHASE2: bit j = tap j also feeds the phase-1 columns (eab_conv_desc.p2_mask1)
    p2_mask1: int = 0


@dataclass
class FinalizeOp:
    stats: Ref
    B: int
    C: int
    nsets: int
    stat_tiles: int
    count: int
    eps: float
    gamma0: Ref
    beta0: Ref
    xf0: Ref
    gamma1: Optional[Ref] = None
    beta1: Optional[Ref] = None
    xf1: Optional[Ref] = None
    name: str = ""
    kind: int = OP_IN_FINALIZE


@dataclass
class NormActOp:
    a: Ref
    xfa: Ref
    slopea: Ref
    b: Optional[Ref]
    xfb: Optional[Ref]
    slopeb: Optional[Ref]
    out: Ref
    B: int
    P: int
    C: int
    T: int = 0
    win: bool = False
    name: str = ""
    kind: int = OP_NORM_ACT


@dataclass
class LstmOp:
    x: Ref
    ln_g: Optional[Ref]
    ln_b: Optional[Ref]
    ln_eps: float
    wcat: Ref
    bias: Ref
    h_out: Ref
    B: int
    T: int
    F: int
    precision: int = PREC_F32
    c_state: Optional[Ref] = None        # streaming: cell state [B*F][64] carried between chunks
    win: bool = False
    name: str = ""
    kind: int = OP_LSTM64


@dataclass
class BfwOp:
    y1: Ref
    w2: Ref
    b2: Ref
    x: Ref
    out: Ref
    bfw: Optional[Ref]
    B: int
    T: int
    F: int
    M: int
    w1: Optional[Ref] = None             # first Linear of the MLP fused in (then y1 is the LSTM output h)
    b1: Optional[Ref] = None
    win: bool = False
    name: str = ""
    kind: int = OP_BFW_FS


@dataclass
class MemsetOp:
    ptr: Ref
    nfloats: int
    B: int = 0                           # streaming: the tensor is [B][T][row] and only the chunk's rows are cleared
    T: int = 0
    row: int = 0
    win: bool = False
    name: str = ""
    kind: int = OP_MEMSET0


@dataclass
class GagPackOp:
    """two planar (B,2,T,F) inputs -> enc_in [B][T][F][4] = (in_r, in_i, pre_r, pre_i) and
    pre [B][T][GAG_PRE_LD] (channel f*2+ri, zero padded)."""
    inpt: Ref
    pre_x: Ref
    enc_in: Ref
    pre: Ref
    B: int
    T: int
    F: int
    win: bool = False
    name: str = ""
    kind: int = OP_GAG_PACK


@dataclass
class GagCrmOp:
    """GlanceGazeModule tail (GaGNet.py:127-133): out = pre * act(g) + (r, i) per TF bin;
    pre/pre_out [B][T][GAG_PRE_LD], g/r/i [B][T][GAG_LIN_LD], planar [B][2][T][F]."""
    pre: Ref
    g: Ref
    r: Ref
    i: Ref
    pre_out: Ref
    planar: Ref
    B: int
    T: int
    F: int
    act: int
    win: bool = False
    name: str = ""
    kind: int = OP_GAG_CRM


@dataclass
class ClnStatsOp:
    """cumulative-LayerNorm statistics (eab_cln_stats_f32): mr[b][t] = (cum_mean, rstd) of x or prelu(x, slope)"""
    x: Ref
    slope: Optional[Ref]
    sums: Ref
    state: Optional[Ref]
    mr: Ref
    B: int
    T: int
    P: int
    C: int
    eps: float
    win: bool = False
    name: str = ""
    kind: int = OP_CLN_STATS


@dataclass
class ClnApplyOp:
    x: Ref
    mr: Ref
    gain: Ref
    bias: Ref
    slope: Ref
    add: Optional[Ref]
    out: Ref
    B: int
    T: int
    P: int
    C: int
    mode: int
    win: bool = False
    name: str = ""
    kind: int = OP_CLN_APPLY


@dataclass
class GateRowsOp:
    a: Ref
    r: Ref
    z: Ref
    B: int
    T: int
    row: int
    win: bool = False
    name: str = ""
    kind: int = OP_GATE_ROWS


def conv_tiles(T: int, No: int, bm: int) -> int:
    return (T * No + bm - 1) // bm


def patch_positions(bm: int, No: int, Fin: int, istride: int, dt, ioff) -> int:
    """worst-case input positions a bm-row tile touches (cg_patch_positions in conv_gemm.hip)"""
    dt_min, io_min, io_max = min(0, min(dt)), min(0, min(ioff)), max(0, max(ioff))
    hi_need = (No - 1) * istride + io_max - (Fin - 1)
    Fp = Fin - io_min + max(hi_need, 0)
    return ((bm - 1) // No + 2 - dt_min) * Fp


# ----------------------------------------------------------------------------
# weight packing
# ----------------------------------------------------------------------------
def glu_row_order(N: int) -> np.ndarray:
    """packed row r -> original row, so that a wave's two 32-column MFMA tiles
    hold value and gate of the same 32 channels (include/eabnet_hip.h)."""
    r = np.arange(N)
    return (r % 64 // 32) * (N // 2) + (r // 64) * 32 + r % 32


def pack_taps(w_nck: np.ndarray, taps_k: Sequence[int]) -> np.ndarray:
    """w_nck: [N][C][ntaps_all] (taps flattened); taps_k: which flattened taps,
    in kernel order.  Returns [N][len(taps_k)*UPT*16] with unit layout
    (tap, 16-channel block), zero padded."""
    N, C, _ = w_nck.shape
    upt = (C + 15) // 16
    out = np.zeros((N, len(taps_k), upt * 16), dtype=np.float32)
    for j, k in enumerate(taps_k):
        out[:, j, :C] = w_nck[:, :, k]
    return out.reshape(N, -1)


def pack_f16x3(wp: np.ndarray) -> np.ndarray:
    """fp32 packed weights [N][Kpad] -> the f16x3 operand layout (include/eabnet_hip.h,
    EAB_PREC_F16X3): per row and per group of 4 consecutive k, 4 fp16 hi then 4 fp16 lo with
    w = hi + lo; returned as float32-typed storage of identical byte size."""
    N, K = wp.shape
    assert K % 4 == 0
    if np.abs(wp).max(initial=0.0) >= 65504.0:
        raise ValueError("f16x3 precision needs |weight| < 65504")
    hi = wp.astype(np.float16)
    lo = (wp - hi.astype(np.float32)).astype(np.float16)
    out = np.empty((N, K // 4, 8), dtype=np.float16)
    out[:, :, :4] = hi.reshape(N, K // 4, 4)
    out[:, :, 4:] = lo.reshape(N, K // 4, 4)
    return np.ascontiguousarray(out).view(np.float32).reshape(N, K)


def _frag_index(N: int, K: int):
    """(n, k) of every element of the fragment-order array [N/16][K/16][64 lanes][4] (include/eabnet_hip.h, EAB_KORDER_FRAG)"""
    nb, m2, lane, j = np.meshgrid(np.arange(N // 16), np.arange(K // 16), np.arange(64), np.arange(4), indexing="ij")
    return nb * 16 + (lane & 15), 16 * m2 + 8 * (j >> 1) + 2 * (lane >> 4) + (j & 1)


def frag_row_order(N: int, dual: bool) -> np.ndarray:
    """fragment row -> original row.  Natural, except the dual-gate form: row block 2w+g (g = 0 value, 1 gate) holds
    the original rows g*N/2 + 16w .. +15, so that a wave owns value and gate of the same 16 channels."""
    r = np.arange(N)
    if not dual:
        return r
    return ((r // 16) % 2) * (N // 2) + (r // 32) * 16 + r % 16


def pack_frag(wp: np.ndarray, dual: bool = False) -> np.ndarray:
    """[N][Kpad] (rows in original order, k = tap*UPT*16 + channel) -> MFMA-fragment order for conv_st_kernel:
    a wave's b128 load of (row block, K step) is one contiguous 1-KB read of exactly its v_mfma_f32_16x16x4 operands."""
    N, K = wp.shape
    assert N % 16 == 0 and K % 16 == 0
    n, k = _frag_index(N, K)
    # (integer input = an index image of the training programs' parameter gather: same permutation, dtype kept)
    return np.ascontiguousarray(wp[frag_row_order(N, dual)][n, k],
                                dtype=wp.dtype if np.issubdtype(wp.dtype, np.integer) else np.float32).reshape(-1)


def unpack_frag(flat: np.ndarray, N: int, K: int, dual: bool = False) -> np.ndarray:
    """inverse of pack_frag (tests/emulator.py)"""
    n, k = _frag_index(N, K)
    w = np.empty((N, K), np.float32)
    w[n, k] = np.asarray(flat, np.float32).reshape(n.shape)
    out = np.empty_like(w)
    out[frag_row_order(N, dual)] = w
    return out


class WeightArena:
    """Flat fp32 buffer of packed parameters + name -> Ref table."""

    def __init__(self):
        self.chunks: List[np.ndarray] = []
        self.chunks_by_name: Dict[str, np.ndarray] = {}
        self.size = 0
        self.index: Dict[str, Ref] = {}

    def add(self, name: str, arr: np.ndarray) -> Ref:
        if name in self.index:
            return self.index[name]
        flat = np.ascontiguousarray(arr, dtype=np.float32).reshape(-1)
        pad = (-flat.size) % ALIGN
        ref = Ref("w", self.size)
        self.chunks.append(flat)
        self.chunks_by_name[name] = flat
        if pad:
            self.chunks.append(np.zeros(pad, dtype=np.float32))
        self.size += flat.size + pad
        self.index[name] = ref
        return ref

    def flat(self) -> np.ndarray:
        return np.concatenate(self.chunks) if self.chunks else np.zeros(0, np.float32)


# ----------------------------------------------------------------------------
# the lowering
# ----------------------------------------------------------------------------
@dataclass
class Program:
    cfg: object                    # NetConfig or GagConfig
    B: int
    T: int
    F: int
    ops: list
    weights: np.ndarray            # packed parameter arena (fp32)
    act_floats: int                # workspace arena size (floats)
    taps: Dict[str, Act]           # named materialised activations (debug / tests)
    flops: int = 0                 # MAC*2 of all MFMA ops (algorithmic, un-padded)
    # independent branches (GaGNet's three S-TCM chains per stage): lanes[k] = stream lane of op k,
    # sync[k] = [("fork" | "join", lanes), ...] applied BEFORE op k (index len(ops) = after the last op).
    # fork: the lanes wait for lane 0; join: lane 0 waits for the lanes.  Program order is always a
    # valid sequential order, so an executor may ignore both (tests/emulator.py does).
    lanes: List[int] = field(default_factory=list)
    sync: Dict[int, list] = field(default_factory=dict)
    chunk: int = 0                 # > 0: streaming program, every op works on `chunk` frames from a device-side position
    zero_init: List[Tuple[Ref, int]] = field(default_factory=list)      # (arena ref, floats) that must be zero before the first replay


class Lowering:
    spec_fn = staticmethod(param_specs)

    def __init__(self, cfg, params: Dict[str, np.ndarray], B: int, T: int, F: int = 161,
                 dump_bfw: bool = False, precision: str = "f32", chunk: int = 0):
        cfg.check_supported()
        self.chunk = chunk
        if chunk:
            # streaming = the same program restricted to a moving window of `chunk` frames (eab_time_window):
            # needs every op to be causal with data-independent statistics
            if cfg.norm_type not in ("BN", "cLN") or not cfg.is_causal:
                raise NotImplementedError("streaming needs norm_type='BN' (eval) or 'cLN' and is_causal=True: InstanceNorm "
                                          "statistics and centred S-TCM taps look at the whole utterance")
            if precision == "f16x3":
                raise NotImplementedError("streaming runs in 'f32' or 'bf16' (BASELINE config 5); 'f16x3' is an offline mode")
        if precision not in ("f32", "f16x3", "bf16"):
            raise ValueError(f"precision must be 'f32', 'f16x3' or 'bf16', got {precision!r}")
        self.precision = precision
        self.patch = os.environ.get("EAB_PATCH", "1") != "0"      # tuning knob: 0 = gather pipeline everywhere
        # both output-column phases of a unit transposed convolution in one launch on one staged patch (EPI_PHASE2); 0 = one
        # gather launch per phase (rounds 1-3)
        self.phase2 = os.environ.get("EAB_PHASE2", "1") != "0"
        # small-tile kernel (csrc/conv_st.hip; exact fp32 or bf16 products) for the latency-bound launches: the S-TCN and the 64-column unit
        # convolutions with at most `st_maxno` output columns; EAB_ST=0 puts everything back on conv_gemm_kernel
        self.st = os.environ.get("EAB_ST", "1") != "0" and precision in ("f32", "bf16")
        self.st_glu = os.environ.get("EAB_ST_GLU", "1") != "0"            # gated convolutions of latency-bound layers too
        self.st_glu_tiles = int(os.environ.get("EAB_ST_GLU_TILES", str(2 * CUS)))
        self.st_maxno = int(os.environ.get("EAB_ST_MAXNO", "5"))
        self.fuse_out_in = os.environ.get("EAB_ST_FUSE", "1") != "0"      # out_conv of one S-TCM + in_conv of the next in one launch
        specs = self.spec_fn(cfg)
        specs = {k: v for k, v in specs.items() if v.kind != "bn_count"}     # the step counter is not arithmetic
        missing = [k for k in specs if k not in params]
        if missing:
            raise KeyError(f"missing parameters: {missing[:4]} ...")
        for k, s in specs.items():
            if tuple(params[k].shape) != tuple(s.shape):
                raise ValueError(f"{k}: shape {tuple(params[k].shape)} != {tuple(s.shape)}")
        self.cfg, self.P, self.B, self.T, self.F = cfg, params, B, T, F
        self.W = WeightArena()
        self.ops: list = []
        self.act_size = 0
        self.taps: Dict[str, Act] = {}
        self.flops = 0
        self.dump_bfw = dump_bfw
        self.bn = cfg.norm_type == "BN"
        self.cln = cfg.norm_type == "cLN"
        self.add = cfg.intra_connect == "add"
        self.zero_init: List[Tuple[Ref, int]] = []              # arena regions that must be zero before the first replay
        # InstanceNorm finalisation inside the producing convolution (last-arriving tile): implemented and parity-green, but
        # the agent-scope release every tile needs (__threadfence = L2 write-back on a multi-XCD part) doubled the
        # convolution time (7.2 -> 14.3 ms per step, gpurun_out/r02_b4.json), so the stand-alone launches stay the default
        self.fuse_fin = os.environ.get("EAB_FUSE_FIN", "0") == "1"
        self._lane_marks: List[Tuple[int, int]] = [(0, 0)]      # (first op index, stream lane from there on)
        self.sync: Dict[int, list] = {}

    def set_lane(self, lane: int) -> None:
        """Ops emitted from now on belong to parallel branch `lane` (0 = the main stream)."""
        self._lane_marks.append((len(self.ops), lane))

    def mark(self, what: str, lanes) -> None:
        self.sync.setdefault(len(self.ops), []).append((what, list(lanes)))

    def lane_of_ops(self) -> List[int]:
        out, marks = [], self._lane_marks + [(len(self.ops), 0)]
        for (a, lane), (b, _) in zip(marks[:-1], marks[1:]):
            out += [lane] * (b - a)
        return out

    # -- arenas ---------------------------------------------------------------
    def alloc(self, nfloats: int) -> Ref:
        ref = Ref("a", self.act_size)
        self.act_size += nfloats + ((-nfloats) % ALIGN)
        return ref

    def alloc_act(self, F: int, C: int) -> Ref:
        return self.alloc(self.B * self.T * F * C)

    def vec(self, key: str) -> Ref:
        return self.W.add(key, self.P[key])

    def static_xf(self, name: str, scale: np.ndarray, shift: np.ndarray) -> Ref:
        """(scale, shift) table that does not depend on the data, laid out like the output of
        eab_in_finalize_f32 ([B][C][2]) so every consumer reads it the same way."""
        tab = np.stack([scale, shift], axis=-1).astype(np.float32)            # (C, 2)
        return self.W.add(f"{name}#xf", np.broadcast_to(tab, (self.B,) + tab.shape))

    def bn_xf(self, norm: str) -> Ref:
        """BatchNorm in eval mode (NormSwitch BN branch, EaBNet.py:677-681) is a per-channel affine
        map of the running statistics: scale = g / sqrt(var + eps), shift = b - mean * scale."""
        g, b = self.P[f"{norm}.norm.weight"].astype(np.float64), self.P[f"{norm}.norm.bias"].astype(np.float64)
        mu = self.P[f"{norm}.norm.running_mean"].astype(np.float64)
        var = self.P[f"{norm}.norm.running_var"].astype(np.float64)
        scale = g / np.sqrt(var + EPS_IN)
        return self.static_xf(norm, scale, b - mu * scale)

    def fixed_norm(self, norm: Optional[str], C: int) -> Optional[Ref]:
        """Table of a norm that needs no statistics pass (BatchNorm eval; no norm at all = identity),
        None for InstanceNorm."""
        if norm is None:
            return self.static_xf(f"identity{C}", np.ones(C), np.zeros(C))
        return self.bn_xf(norm) if self.bn else None

    def cln_norm(self, name: str, raw: Ref, F: int, C: int, norm: str, slope_key: str, mode: int,
                 add: Optional[Act] = None) -> Act:
        """Cumulative LayerNorm + PReLU (reference CumulativeLayerNorm1d/2d, EaBNet.py:696-769) as a statistics op and a
        materialising apply op; the running sums are the streaming state."""
        B, T = self.B, self.T
        Pn = F * C
        sums = self.alloc(B * T * 4)                                  # [B][T][2] doubles
        state = self.alloc(B * 4) if self.chunk else None             # [B][2] doubles
        mr = self.alloc(B * T * 2)
        gain = self.W.add(f"{norm}.norm.gain#c", self.P[f"{norm}.norm.gain"].reshape(C))
        bias = self.W.add(f"{norm}.norm.bias#c", self.P[f"{norm}.norm.bias"].reshape(C))
        slope = self.vec(slope_key)
        win = bool(self.chunk)
        self.ops.append(ClnStatsOp(x=raw, slope=slope if mode == XF_PRELU_NORM else None, sums=sums, state=state, mr=mr, B=B, T=T,
                                   P=Pn, C=C, eps=EPS_IN, win=win, name=name + ".cln_stats"))
        out = self.alloc_act(F, C)
        self.ops.append(ClnApplyOp(x=raw, mr=mr, gain=gain, bias=bias, slope=slope, add=add.ref if add is not None else None,
                                   out=out, B=B, T=T, P=Pn, C=C, mode=mode, win=win, name=name + ".cln"))
        return Act(out, F, C)

    # -- generic conv emission -----------------------------------------------------
    def pick_bm(self, No: int) -> int:
        import os
        if os.environ.get("EAB_BM"):                     # tuning knob
            return int(os.environ["EAB_BM"])
        return 128 if self.B * conv_tiles(self.T, No, 128) >= 2 * CUS else 64

    def pick_st_bm(self, No: int, N: int, Kpad: int, dual: bool = False, max_tiles: Optional[int] = None) -> int:
        """Rows per tile of a small-tile launch: the largest of 64 / 32 / 16 that still gives the chip two workgroups per
        CU (else 16), within the kernel's LDS budget (the whole K extent of a tile is staged at once) and, when a consumer
        merges this launch's InstanceNorm partials itself, within its 64-tile limit."""
        if os.environ.get("EAB_ST_BM"):                  # tuning knob
            return int(os.environ["EAB_ST_BM"])
        rows = (self.chunk or self.T) * No
        cands = [bm for bm in (64, 32, 16)
                 if (bm < 64 or N == 64) and (2 if dual else 1) * bm * (Kpad + 4) * 4 <= 96 * 1024]
        ok = [bm for bm in cands if max_tiles is None or conv_tiles(self.T, No, bm) <= max_tiles]
        assert ok, "no small-tile geometry fits"
        for bm in ok:
            if self.B * ((rows + bm - 1) // bm) >= 2 * CUS:
                return bm
        return ok[-1]

    def emit_conv(self, name: str, srcs: Sequence[Act], w: Ref, bias: Optional[Ref], N: int, Kpad: int,
                  Fout: int, No: int, ostride: int, ophase: int, istride: int, dt, ioff, epi: int,
                  dst: Ref, stats: Optional[Ref] = None, nsets: int = 0, stat_slopes=(None, None),
                  stat_tiles: int = 0, stat_tile0: int = 0, bm: Optional[int] = None, aux: Optional[Ref] = None,
                  dst_acc: Optional[Ref] = None, fin: Optional[dict] = None, slope1: Optional[Ref] = None,
                  xf1: Optional[Ref] = None, patch_ok: bool = True, st: bool = False, ph1: Optional[dict] = None,
                  p2_mask1: int = 0) -> ConvOp:
        """fin = dict(stats, tiles, nsets, count, norms=[...]) asks the kernel to reduce the
        producer's InstanceNorm partials itself (single source, transform order from srcs[0].mode);
        slope1 (+ xf1 when the table is static) = second transform of the SAME source for EPI_DUALGATE.
        st = small-tile kernel (KORDER_FRAG): `w` (and ph1["w"]) are [N][Kpad] in ORIGINAL row order and are re-packed in
        fragment order here; ph1 = dict(w, Kpad, No, ophase, dt, ioff): second output-column phase in the same launch."""
        assert 1 <= len(srcs) <= 2 and len(dt) == len(ioff) <= MAX_TAPS
        assert st or ph1 is None
        s0 = srcs[0]
        s1 = srcs[1] if len(srcs) == 2 else None
        modes = {s.mode for s in srcs if s.xf is not None or (fin is not None and s.mode != XF_NONE)}
        assert len(modes) <= 1, "both concat sources must use the same transform order"
        mode = modes.pop() if modes else XF_NONE
        if s1 is not None:
            assert s0.F == s1.F and s0.C % 16 == 0
        C0, C1 = s0.C, (s1.C if s1 else 0)
        upt = (C0 + C1 + 15) // 16
        assert Kpad == len(dt) * upt * 16
        bm = bm or self.pick_bm(No)
        # patch pipeline (input patch of a 16-channel chunk staged once, taps read it shifted): pays
        # when several taps re-read the same inputs; needs the patch of a tile to fit its LDS area
        korder = KORDER_TAP
        ph1kw = {}
        if st:
            assert self.precision in ("f32", "bf16") and N in (64, 128, 256) and bm in (16, 32, 64)
            glu_st = epi == EPI_GLU                       # exact fp32 only; weights in passes of 320, not all resident
            assert Kpad <= (ST_GLU_KMAX if glu_st else {64: 320, 128: 320, 256: 64}[N]), \
                "small-tile kernel: the K extent must fit the wave's registers"
            assert epi in (EPI_LINEAR, EPI_RELU, EPI_ADD, EPI_DUALGATE, EPI_GLU)
            assert not glu_st or (self.precision == "f32" and N == 128 and mode == XF_NONE and fin is None)
            korder = KORDER_FRAG
            dual = epi in (EPI_DUALGATE, EPI_GLU)         # a wave owns value and gate of the same 16 channels

            def frag(ref: Ref, K: int) -> Ref:
                key = next(k for k, r in self.W.index.items() if r == ref)
                return self.W.add(key + ".frag", pack_frag(self.W.chunks_by_name[key].reshape(N, K), dual))
            w = frag(w, Kpad)
            if ph1 is not None:
                assert ph1["Kpad"] == len(ph1["dt"]) * upt * 16 and ph1["Kpad"] <= Kpad
                ph1kw = dict(ph1_w=frag(ph1["w"], ph1["Kpad"]), ph1_No=ph1["No"], ph1_ophase=ph1["ophase"], ph1_Kpad=ph1["Kpad"],
                             ph1_dt=list(ph1["dt"]), ph1_ioff=list(ph1["ioff"]))
                self.flops += 2 * self.B * self.T * ph1["No"] * N * len(ph1["dt"]) * (C0 + C1)
        patch_min_n = int(os.environ.get("EAB_PATCH_MIN_N", "128"))      # tuning knob
        if (not st and self.patch and N >= patch_min_n and len(dt) >= 2 and s0.F > 1 and epi != EPI_DUALGATE and mode != XF_PRELU_NORM
                and C0 % 4 == 0 and C1 % 4 == 0 and patch_ok):
            for cand in ((bm,) if bm == 64 else (128, 64)):
                if patch_positions(cand, No, s0.F, istride, dt, ioff) <= PATCH_MAX:
                    korder, bm_p = KORDER_CHUNK, cand
                    break
            if korder == KORDER_CHUNK and bm_p != bm:
                korder = KORDER_TAP        # tile counts of the statistics were planned for bm
        if epi == EPI_PHASE2:              # phase pair of a transposed convolution: exists in the patch pipeline only
            assert not st and N == 128 and patch_positions(bm, No, s0.F, istride, dt, ioff) <= PATCH_MAX and mode != XF_PRELU_NORM
            korder = KORDER_CHUNK
        if korder == KORDER_CHUNK:
            key = next(k for k, r in self.W.index.items() if r == w)
            wt = self.W.chunks_by_name[key].reshape(N, len(dt), upt, 16)
            w = self.W.add(key + ".chunk", np.ascontiguousarray(wt.transpose(0, 2, 1, 3)).reshape(N, Kpad))
        # f16x3 needs bounded operands: every source except the raw network input is either
        # instance-normalised or a sum of such tensors; the first conv stays on exact fp32.
        # bf16 (torch.autocast semantics for the reference's convolutions): same rule, weights stay plain fp32 in
        # memory -- the kernel rounds both operands on their way into LDS
        lowp = self.precision != "f32" and s0.ref.arena != "in" and C0 % 4 == 0 and C1 % 4 == 0 and not any(s.raw for s in srcs)
        prec = PREC_CODE[self.precision] if lowp else PREC_F32
        if prec == PREC_F16X3:
            key = next(k for k, r in self.W.index.items() if r == w)
            wf = self.W.chunks_by_name[key].reshape(N, Kpad)
            w = self.W.add(key + ".f16x3", pack_f16x3(wf))
        finkw = {}
        if fin is not None:
            assert s1 is None and s0.xf is None and mode != XF_NONE
            g = [self.vec(f"{n}.norm.weight") for n in fin["norms"]]
            b = [self.vec(f"{n}.norm.bias") for n in fin["norms"]]
            finkw = dict(fin_stats=fin["stats"], fin_tiles=fin["tiles"], fin_nsets=fin["nsets"], fin_count=fin["count"],
                         fin_eps=EPS_IN, fin_gamma0=g[0], fin_beta0=b[0],
                         fin_gamma1=g[1] if len(g) > 1 else None, fin_beta1=b[1] if len(b) > 1 else None)
        sl1 = slope1 if slope1 is not None else (s1.slope if s1 else None)
        op = ConvOp(src0=s0.ref, src1=s1.ref if s1 else None, xf0=s0.xf, xf1=s1.xf if s1 else xf1,
                    slope0=s0.slope, slope1=sl1, C0=C0, C1=C1, xf_mode=mode, w=w, bias=bias,
                    N=N, Kpad=Kpad, B=self.B, T=self.T, Fin=s0.F, Fout=Fout, No=No, ostride=ostride, ophase=ophase,
                    istride=istride, dt=list(dt), ioff=list(ioff), epi=epi, aux=aux, dst=dst, dst_acc=dst_acc,
                    Cout=N // 2 if epi in (EPI_GLU, EPI_DUALGATE, EPI_PHASE2) else N, stats=stats, nsets=nsets,
                    stat_slope0=stat_slopes[0], stat_slope1=stat_slopes[1], stat_tiles=stat_tiles,
                    stat_tile0=stat_tile0, bm=bm, name=name, precision=prec, korder=korder, win=bool(self.chunk),
                    p2_mask1=p2_mask1, **finkw, **ph1kw)
        self.ops.append(op)
        if epi == EPI_PHASE2:              # phase-1 columns take the masked taps only, and one column fewer when Fout is odd
            n1 = bin(p2_mask1).count("1")
            self.flops += 2 * self.B * self.T * (N // 2) * (C0 + C1) * (No * len(dt) + (Fout // 2) * n1)
        else:
            self.flops += 2 * self.B * self.T * No * N * len(dt) * (C0 + C1)
        return op

    def fuse_finalize(self, ops: Sequence[ConvOp], C: int, norms: Sequence[str]) -> List[Ref]:
        """InstanceNorm finalisation inside the producing launches `ops` (all feeding the same statistics): the tile
        that arrives last per utterance merges the partials and writes the (scale, shift) tables -- no extra launch.
        The arrival counters live in the activation arena (zeroed when the program is bound, re-armed by the kernel)."""
        nsets = len(norms)
        xfs = [self.alloc(self.B * C * 2) for _ in range(nsets)]
        counter = self.alloc(self.B)
        self.zero_init.append((counter, self.B))
        g = [self.vec(f"{n}.norm.weight") for n in norms]
        b = [self.vec(f"{n}.norm.bias") for n in norms]
        for op in ops:
            op.fz_counter, op.fz_eps = counter, EPS_IN
            op.fz_gamma0, op.fz_beta0, op.fz_xf0 = g[0], b[0], xfs[0]
            if nsets == 2:
                op.fz_gamma1, op.fz_beta1, op.fz_xf1 = g[1], b[1], xfs[1]
        return xfs

    def emit_finalize(self, name, stats, C, nsets, stat_tiles, count, norms: Sequence[str]) -> List[Ref]:
        xfs = [self.alloc(self.B * C * 2) for _ in range(nsets)]
        g = [self.vec(f"{n}.norm.weight") for n in norms]
        b = [self.vec(f"{n}.norm.bias") for n in norms]
        self.ops.append(FinalizeOp(stats=stats, B=self.B, C=C, nsets=nsets, stat_tiles=stat_tiles, count=count,
                                   eps=EPS_IN, gamma0=g[0], beta0=b[0], xf0=xfs[0],
                                   gamma1=g[1] if nsets == 2 else None, beta1=b[1] if nsets == 2 else None,
                                   xf1=xfs[1] if nsets == 2 else None, name=name))
        return xfs

    def st_small(self, No: int, glu: bool = False) -> bool:
        """Is a launch with `No` output columns per frame latency-bound?  Yes when 64-row tiles would give the chip fewer
        than two workgroups per CU (one utterance, a streaming chunk, the deepest U-Net levels of a batch) or when the
        layer has at most `st_maxno` columns; those go to the small-tile kernel."""
        # (T, not the streaming chunk: a streamed program must pick the kernel its offline twin of the same (B, T) picks --
        # the two kernels sum in different orders, and streamed frames are promised bit-identical to the offline pass)
        rows = self.T * No
        if glu:      # deep-K gated launches: only where 64-row tiles leave CUs idle (measured: the wide-tile kernel wins above)
            return self.B * ((rows + 63) // 64) < self.st_glu_tiles
        return No <= self.st_maxno or self.B * ((rows + 63) // 64) < 2 * CUS

    def st_ok(self, srcs: Sequence[Act], N: int, glu: bool) -> bool:
        """the small-tile kernel's domain: plain 64/128/256-column launches, source channels 4 * 2^k; gated (GLU) launches
        in exact fp32 on materialised sources"""
        if glu and not (self.st_glu and self.precision == "f32" and N == 128 and all(a.xf is None for a in srcs)):
            return False
        return (self.st and not self.cln and N in (64, 128, 256) and all(a.C in (64, 128, 256) for a in srcs)
                and not any(a.raw or a.ref.arena == "in" for a in srcs)
                and all(a.C <= 128 for a in srcs if a.xf is not None))

    # -- 2-D units -------------------------------------------------------------------
    def conv2d_fwd(self, name: str, srcs: Sequence[Act], wkey: str, glu: bool, norm: Optional[str], act: str,
                   in_perm: Optional[np.ndarray] = None, add: Optional[Act] = None) -> Act:
        """Strided causal Conv2d [(kt,kf), stride (1,2)] (+GLU) -> raw output with
        norm+PReLU pending (norm=None: PReLU only).  Reference GateConv2d EaBNet.py:434-460 /
        Conv2dunit :391-407."""
        wkey = gate_key(self.P, wkey)
        w = self.P[f"{wkey}.weight"]                       # (N, Cin, kt, kf)
        N, Cin, kt, kf = w.shape
        if in_perm is not None:
            w = w[:, in_perm]
        Fin = srcs[0].F
        Fout = (Fin - kf) // 2 + 1
        taps = [(a, c) for a in range(kt) for c in range(kf)]
        st = (self.st_ok(srcs, N, glu) and kt * kf * ((Cin + 15) // 16) * 16 <= (ST_GLU_KMAX if glu else 256)
              and self.st_small(Fout, glu))
        # (small-tile kernel: rows stay in the convolution's own order, emit_conv packs them in fragment order)
        order, tag = (glu_row_order(N), "packed") if glu and not st else (np.arange(N), "rows" if glu else "packed")
        wp = pack_taps(w.reshape(N, Cin, kt * kf)[order], [a * kf + c for a, c in taps])
        wref = self.W.add(f"{wkey}.weight#{tag}", wp)
        bref = self.W.add(f"{wkey}.bias#{tag}", self.P[f"{wkey}.bias"][order])
        Cout = N // 2 if glu else N
        dst = self.alloc_act(Fout, Cout)
        bm = self.pick_st_bm(Fout, N, wp.shape[1]) if st else self.pick_bm(Fout)
        tiles = conv_tiles(self.T, Fout, bm)
        cln = self.cln and norm is not None
        xf = None if cln else self.fixed_norm(norm, Cout)
        stats = self.alloc(self.B * tiles * Cout * 4) if (xf is None and not cln) else None
        op = self.emit_conv(name, srcs, wref, bref, N, wp.shape[1], Fout, Fout, 1, 0, 2,
                            [a - (kt - 1) for a, _ in taps], [c for _, c in taps],
                            EPI_GLU if glu else EPI_LINEAR, dst, stats, 1 if stats else 0, (None, None),
                            tiles if stats else 0, 0, bm, st=st)
        if cln:
            return self.cln_norm(name, dst, Fout, Cout, norm, f"{act}.weight", XF_NORM_PRELU, add)
        if xf is None:
            if self.fuse_fin:
                xf, = self.fuse_finalize([op], Cout, [norm])
            else:
                xf, = self.emit_finalize(name + ".in", stats, Cout, 1, tiles, self.T * Fout, [norm])
        return Act(dst, Fout, Cout, xf, self.vec(f"{act}.weight"), XF_NORM_PRELU)

    def conv2d_transposed(self, name: str, srcs: Sequence[Act], wkey: str, glu: bool, norm: str, act: str,
                          summed: bool = False, add: Optional[Act] = None) -> Act:
        """ConvTranspose2d [(kt,kf), stride (1,2)] + drop of the last kt-1 rows
        (+GLU) as two gather-form launches, one per output-column parity:
          out[t][2o+ph] = sum_{kt} sum_{kf = ph, ph+2, ..} W[kt][kf] . in[t-kt][o-(kf-ph)/2]
        Reference GateConvTranspose2d EaBNet.py:463-490 + Chomp_T :617-624 /
        Deconv2dunit :410-431."""
        wkey = gate_key(self.P, wkey)
        w = self.P[f"{wkey}.weight"]                       # (Cin, N, kt, kf)
        if summed:
            # Skip_connect 'add' (EaBNet.py:499-500): W.(f(a) + g(b)) = [W | W].cat(f(a), g(b)) -- the two
            # sources keep their own pending transforms and the sum is never materialised
            w = np.concatenate([w] * len(srcs), axis=0)
        Cin, N, kt, kf = w.shape
        assert Cin == sum(s.C for s in srcs)
        Fin = srcs[0].F
        Fout = (Fin - 1) * 2 + kf
        No = [(Fout + 1) // 2, Fout // 2]
        upt = (Cin + 15) // 16
        st = (self.st_ok(srcs, N, glu) and len(range(0, kf, 2)) * kt * upt * 16 <= (ST_GLU_KMAX if glu else 256)
              and self.st_small(No[0], glu))
        order, tag = (glu_row_order(N), "packed") if glu and not st else (np.arange(N), "rows" if glu else "packed")
        wn = np.ascontiguousarray(w.transpose(1, 0, 2, 3)).reshape(N, Cin, kt * kf)[order]
        bref = self.W.add(f"{wkey}.bias#{tag}", self.P[f"{wkey}.bias"][order])
        Cout = N // 2 if glu else N
        dst = self.alloc_act(Fout, Cout)
        bm = self.pick_st_bm(No[0] + No[1], N, len(range(0, kf, 2)) * kt * upt * 16) if st else self.pick_bm(No[0])
        tiles = [conv_tiles(self.T, n, bm) for n in No]
        cln = self.cln
        xf = None if cln else self.fixed_norm(norm, Cout)
        stats = self.alloc(self.B * sum(tiles) * Cout * 4) if (xf is None and not cln) else None
        phase_ops = []
        phases = []
        for ph in (0, 1):
            taps = [(a, c) for a in range(kt) for c in range(ph, kf, 2)]
            wp = pack_taps(wn, [a * kf + c for a, c in taps])
            wref = self.W.add(f"{wkey}.weight#{tag}.ph{ph}", wp)
            phases.append(dict(w=wref, Kpad=wp.shape[1], No=No[ph], ophase=ph, dt=[-a for a, _ in taps],
                               ioff=[-(c - ph) // 2 for _, c in taps]))
        # (InstanceNorm configurations only: BatchNorm / cLN programs can be streamed, and a streamed program must run the
        # kernels -- the summation orders -- of its offline twin; those keep one launch per phase)
        fused = (not st and not glu and self.phase2 and self.patch and not self.chunk and not self.bn and not self.cln
                 and N == 64 and Cin % 4 == 0 and all(s.C % 4 == 0 for s in srcs))
        if fused:
            # ONE launch for both output-column phases on ONE staged input patch (EPI_PHASE2, conv_gemm.hip): virtual
            # convolution with 2N columns (phase-0 rows, phase-1 rows of a channel paired like value / gate of the GLU form)
            # over the union of the two phases' taps; a tap that only phase 0 uses has zero phase-1 rows, skipped in the kernel.
            # The input is fetched and transformed once instead of once per (phase, tap): 3 -> 1 for the (1,3) unit kernels.
            shifts = sorted({(a, c // 2) for a in range(kt) for c in range(kf)})      # (kt index, input shift o - fi)
            wv = np.zeros((2 * N, Cin, len(shifts)), dtype=np.float32)
            mask1 = 0
            for j, (a, sft) in enumerate(shifts):
                wv[:N, :, j] = wn[:, :, a * kf + 2 * sft]                              # phase 0: kf = 2 * shift
                if 2 * sft + 1 < kf:
                    wv[N:, :, j] = wn[:, :, a * kf + 2 * sft + 1]                      # phase 1: kf = 2 * shift + 1
                    mask1 |= 1 << j
            dts, ios = [-a for a, _ in shifts], [-sft for _, sft in shifts]
            bm_f = self.pick_bm(No[0])
            fused = (mask1 & 1) == 1 and patch_positions(bm_f, No[0], Fin, 1, dts, ios) <= PATCH_MAX
        if fused:
            order2 = glu_row_order(2 * N)
            wp = pack_taps(wv[order2], list(range(len(shifts))))
            wref = self.W.add(f"{wkey}.weight#phase2", wp)
            b2 = np.concatenate([self.P[f"{wkey}.bias"]] * 2)[order2]
            bref2 = self.W.add(f"{wkey}.bias#phase2", b2)
            bm = bm_f
            tiles = [conv_tiles(self.T, No[0], bm)]
            if stats is not None:              # re-plan the partials for the fused launch's tiles (the region above was the
                self.act_size = stats.off      # most recent allocation: give it back)
                stats = self.alloc(self.B * tiles[0] * Cout * 4)
            phase_ops.append(self.emit_conv(name, srcs, wref, bref2, 2 * N, wp.shape[1], Fout, No[0], 2, 0, 1, dts, ios, EPI_PHASE2,
                                            dst, stats, 1 if stats else 0, (None, None), tiles[0] if stats else 0, 0, bm,
                                            p2_mask1=mask1))
        elif st:
            # small-tile kernel: both output-column phases in ONE launch (tiles of phase 0, then of phase 1, per utterance)
            p0, p1 = phases
            phase_ops.append(self.emit_conv(name, srcs, p0["w"], bref, N, p0["Kpad"], Fout, No[0], 2, 0, 1, p0["dt"], p0["ioff"],
                                            EPI_GLU if glu else EPI_LINEAR, dst, stats, 1 if stats else 0, (None, None),
                                            sum(tiles) if stats else 0, 0, bm, st=True, ph1=p1))
        else:
            for ph, q in enumerate(phases):
                phase_ops.append(self.emit_conv(f"{name}.ph{ph}", srcs, q["w"], bref, N, q["Kpad"], Fout, No[ph], 2, ph, 1,
                                                q["dt"], q["ioff"],
                                                EPI_GLU if glu else EPI_LINEAR, dst, stats, 1 if stats else 0, (None, None),
                                                sum(tiles) if stats else 0, (0 if ph == 0 else tiles[0]) if stats else 0, bm))
        if cln:
            return self.cln_norm(name, dst, Fout, Cout, norm, f"{act}.weight", XF_NORM_PRELU, add)
        if xf is None:
            if self.fuse_fin:
                xf, = self.fuse_finalize(phase_ops, Cout, [norm])
            else:
                xf, = self.emit_finalize(name + ".in", stats, Cout, 1, sum(tiles), self.T * Fout, [norm])
        return Act(dst, Fout, Cout, xf, self.vec(f"{act}.weight"), XF_NORM_PRELU)

    def materialise(self, name: str, a: Act, b: Optional[Act] = None) -> Act:
        """out = f_a(a) [+ f_b(b)]: the En_unet_module residual (EaBNet.py:386)
        or a plain norm+PReLU apply where the tensor must exist in memory."""
        assert a.xf is not None and a.mode == XF_NORM_PRELU
        out = self.alloc_act(a.F, a.C)
        self.ops.append(NormActOp(a=a.ref, xfa=a.xf, slopea=a.slope, b=b.ref if b else None,
                                  xfb=b.xf if b else None, slopeb=b.slope if b else None, out=out, B=self.B,
                                  P=self.T * a.F, C=a.C, T=self.T, win=bool(self.chunk), name=name))
        act = Act(out, a.F, a.C)
        self.taps[name] = act
        return act

    def unet_module(self, pre: str, srcs: Sequence[Act], scale: int, transposed: bool,
                    in_perm: Optional[np.ndarray] = None) -> Act:
        """En_unet_module.forward, reference EaBNet.py:372-388."""
        if transposed:
            g = self.conv2d_transposed(f"{pre}.in_conv", srcs, f"{pre}.in_conv.0.conv.0", True,
                                       f"{pre}.in_conv.1", f"{pre}.in_conv.2")
        else:
            g = self.conv2d_fwd(f"{pre}.in_conv", srcs, f"{pre}.in_conv.0.conv.1", True,
                                f"{pre}.in_conv.1", f"{pre}.in_conv.2", in_perm)
        y = g
        downs = []
        for j in range(scale):
            q = f"{pre}.enco.{j}.conv"
            y = self.conv2d_fwd(q, [y], f"{q}.0", False, f"{q}.1", f"{q}.2")
            downs.append(y)
        for j in range(scale):
            q = f"{pre}.deco.{j}.deconv"
            ins = [y] if j == 0 else [y, downs[-(j + 1)]]
            y = self.conv2d_transposed(q, ins, f"{q}.0", False, f"{q}.1", f"{q}.2", summed=self.add and j > 0,
                                       add=g if (self.cln and j == scale - 1) else None)
        if self.cln:                    # every activation is already materialised; the residual rode on the last apply
            self.taps[pre] = y
            return y
        return self.materialise(pre, g, y)

    # -- squeezed TCM --------------------------------------------------------------------
    def tcm(self, pre: str, x: Act, dilation: int, x_acc: Optional[Ref], perm: np.ndarray, next_pre: Optional[str] = None,
            have_in: Optional[dict] = None):
        """SqueezedTCM.forward, reference EaBNet.py:572-578, on [B][T][1][256], in three
        launches: in_conv (+ statistics of both branch PReLUs) -> left*sigmoid(right) in ONE
        dual-transform gated conv -> out_conv + residual.  T/64 tiles per utterance are few, so
        each consumer reduces the InstanceNorm partials itself (no finalize launches).
        Small-tile kernel, exact fp32: nothing but the residual stream lies between this block's out_conv and the NEXT
        block's in_conv (`next_pre`), so the two run as one launch (eab_conv_desc.f2_*); the next call then gets the
        finished in_conv as `have_in`.  Returns (output, have_in for the next block or None)."""
        if self.cln:
            return self.tcm_cln(pre, x, dilation, x_acc, perm), None
        cfg, T, B = self.cfg, self.T, self.B
        D, cd, kd = cfg.d_feat, cfg.cd1, cfg.kd1
        bn = self.bn                                   # BatchNorm eval: static tables, no statistics at all
        Kd = kd * ((cd + 15) // 16) * 16
        use_st = (self.st and D == 256 and cd == 64 and conv_tiles(T, 1, 32) <= 64)
        if use_st:
            # small-tile kernel: 16- or 32-row tiles (2-4 x the workgroups of a 64-row launch, one memory round trip each)
            mt = None if bn else 64
            bm_in, bm_lr, bm_out = (self.pick_st_bm(1, cd, D, max_tiles=mt), self.pick_st_bm(1, 2 * cd, Kd, dual=True, max_tiles=mt),
                                    self.pick_st_bm(1, D, cd, max_tiles=mt))     # (fused with the next in_conv: its tiles carry partials)
            # the branch pair keeps 160 registers of weights per wave, i.e. one workgroup per CU: 16-row tiles only while
            # they fit the chip in one round (measured at B = 16, T = 401: 416 tiles of 16 rows 22 us, 208 of 32 rows 16 us)
            if B * conv_tiles(T, 1, 16) > CUS and (bn or conv_tiles(T, 1, 32) <= 64):
                bm_lr = 32
            if os.environ.get("EAB_ST_BM_LR"):           # tuning knob
                bm_lr = int(os.environ["EAB_ST_BM_LR"])
        else:
            bm_in = bm_lr = bm_out = 64
        tiles, tiles_lr = conv_tiles(T, 1, bm_in), conv_tiles(T, 1, bm_lr)
        nL, nR, nO = f"{pre}.left_conv.1", f"{pre}.right_conv.1", f"{pre}.out_conv.1"
        slL, slR = self.vec(f"{pre}.left_conv.0.weight"), self.vec(f"{pre}.right_conv.0.weight")
        if have_in is not None:                         # produced by the previous block's fused out_conv launch
            y, st, tiles = have_in["y"], have_in["st"], have_in["tiles"]
        else:
            # in_conv 1x1 (no bias); statistics of BOTH branch PReLUs of its output
            w_in = self.P[f"{pre}.in_conv.weight"][:, perm, :]               # (cd, D, 1)
            wref = self.W.add(f"{pre}.in_conv.weight#packed", pack_taps(w_in, [0]))
            y = self.alloc_act(1, cd)
            st = None if bn else self.alloc(B * tiles * 2 * cd * 4)
            self.emit_conv(f"{pre}.in_conv", [x], wref, None, cd, D, 1, 1, 1, 0, 1, [0], [0], EPI_LINEAR, y,
                           st, 0 if bn else 2, (None, None) if bn else (slL, slR), 0 if bn else tiles, 0, bm_in, st=use_st)
        assert bn or max(tiles, tiles_lr) <= 64, "in-kernel finalisation is sized for <= 64 partial tiles per utterance"
        # z = left(y) * sigmoid(right(y)): columns [0,cd) see PReLU_L/IN_L(y), columns [cd,2cd) PReLU_R/IN_R(y)
        # taps (EaBNet.py:550-553): all in the past when causal, centred otherwise
        span = (kd - 1) * dilation
        lead = span if cfg.is_causal else span // 2
        dts = [j * dilation - lead for j in range(kd)]
        wlr = np.concatenate([self.P[f"{pre}.left_conv.3.weight"], self.P[f"{pre}.right_conv.3.weight"]], axis=0)
        if use_st:     # rows stay in original order: emit_conv packs them in fragment order (value / gate blocks per wave)
            wd = self.W.add(f"{pre}.lr_conv.weight#rows", pack_taps(wlr, range(kd)))
        else:
            wd = self.W.add(f"{pre}.lr_conv.weight#packed", pack_taps(wlr[glu_row_order(2 * cd)], range(kd)))
        z = self.alloc_act(1, cd)
        st2 = None if bn else self.alloc(B * tiles_lr * cd * 4)
        slO = self.vec(f"{pre}.out_conv.0.weight")
        self.emit_conv(f"{pre}.lr_conv", [Act(y, 1, cd, self.bn_xf(nL) if bn else None, slL, XF_PRELU_NORM)], wd, None,
                       2 * cd, Kd, 1, 1, 1, 0, 1, dts, [0] * kd, EPI_DUALGATE, z, st2,
                       0 if bn else 1, (None, None) if bn else (slO, None), 0 if bn else tiles_lr, 0, bm_lr, slope1=slR,
                       xf1=self.bn_xf(nR) if bn else None,
                       fin=None if bn else dict(stats=st, tiles=tiles, nsets=2, count=T, norms=[nL, nR]), st=use_st)
        w_out = self.P[f"{pre}.out_conv.2.weight"][perm]                  # (D, cd, 1), rows permuted
        wo = self.W.add(f"{pre}.out_conv.2.weight#packed", pack_taps(w_out, [0]))
        xn = self.alloc_act(1, D)
        op = self.emit_conv(f"{pre}.out_conv", [Act(z, 1, cd, self.bn_xf(nO) if bn else None, slO, XF_PRELU_NORM)], wo, None,
                            D, cd, 1, 1, 1, 0, 1, [0], [0], EPI_ADD, xn, bm=bm_out, aux=x.ref, dst_acc=x_acc,
                            fin=None if bn else dict(stats=st2, tiles=tiles_lr, nsets=1, count=T, norms=[nO]), st=use_st)
        nxt = None
        if use_st and next_pre is not None and self.precision == "f32" and self.fuse_out_in:
            # ... + the next block's in_conv on the rows just produced (one launch instead of two)
            tiles_n = conv_tiles(T, 1, bm_out)
            w_in_n = self.P[f"{next_pre}.in_conv.weight"][:, perm, :]
            op.f2_w = self.W.add(f"{next_pre}.in_conv.weight#frag2", pack_frag(pack_taps(w_in_n, [0])))
            op.f2_dst, op.f2_N = self.alloc_act(1, cd), cd
            if not bn:
                op.f2_stats, op.f2_nsets, op.f2_stat_tiles = self.alloc(B * tiles_n * 2 * cd * 4), 2, tiles_n
                op.f2_stat_slope0 = self.vec(f"{next_pre}.left_conv.0.weight")
                op.f2_stat_slope1 = self.vec(f"{next_pre}.right_conv.0.weight")
            op.name = f"{pre}.out_conv+{next_pre}.in_conv"
            self.flops += 2 * B * T * cd * D
            nxt = dict(y=op.f2_dst, st=op.f2_stats, tiles=tiles_n)
        return Act(xn, 1, D), nxt

    def tcm_cln(self, pre: str, x: Act, dilation: int, x_acc: Optional[Ref], perm: np.ndarray) -> Act:
        """SqueezedTCM.forward (EaBNet.py:572-578) with cumulative LayerNorms: every normalised tensor is materialised
        (the norm's statistics depend on the frame), the two branches are two convolutions and a gate op."""
        cfg, T, B = self.cfg, self.T, self.B
        D, cd, kd = cfg.d_feat, cfg.cd1, cfg.kd1
        bm = 64
        w_in = self.P[f"{pre}.in_conv.weight"][:, perm, :]
        y = self.alloc_act(1, cd)
        self.emit_conv(f"{pre}.in_conv", [x], self.W.add(f"{pre}.in_conv.weight#packed", pack_taps(w_in, [0])), None, cd, D, 1, 1,
                       1, 0, 1, [0], [0], EPI_LINEAR, y, bm=bm)
        yL = self.cln_norm(f"{pre}.left", y, 1, cd, f"{pre}.left_conv.1", f"{pre}.left_conv.0.weight", XF_PRELU_NORM)
        yR = self.cln_norm(f"{pre}.right", y, 1, cd, f"{pre}.right_conv.1", f"{pre}.right_conv.0.weight", XF_PRELU_NORM)
        span = (kd - 1) * dilation
        dts = [j * dilation - span for j in range(kd)]              # causal (cLN is only built for is_causal=True)
        Kp = kd * ((cd + 15) // 16) * 16
        a, r = self.alloc_act(1, cd), self.alloc_act(1, cd)
        for side, src, dst in (("left_conv", yL, a), ("right_conv", yR, r)):
            wref = self.W.add(f"{pre}.{side}.3.weight#packed", pack_taps(self.P[f"{pre}.{side}.3.weight"], range(kd)))
            self.emit_conv(f"{pre}.{side}", [src], wref, None, cd, Kp, 1, 1, 1, 0, 1, dts, [0] * kd, EPI_LINEAR, dst, bm=bm)
        z = self.alloc_act(1, cd)
        self.ops.append(GateRowsOp(a=a, r=r, z=z, B=B, T=T, row=cd, win=bool(self.chunk), name=f"{pre}.gate"))
        zo = self.cln_norm(f"{pre}.out", z, 1, cd, f"{pre}.out_conv.1", f"{pre}.out_conv.0.weight", XF_PRELU_NORM)
        w_out = self.P[f"{pre}.out_conv.2.weight"][perm]
        xn = self.alloc_act(1, D)
        self.emit_conv(f"{pre}.out_conv", [zo], self.W.add(f"{pre}.out_conv.2.weight#packed", pack_taps(w_out, [0])), None, D, cd, 1,
                       1, 1, 0, 1, [0], [0], EPI_ADD, xn, bm=bm, aux=x.ref, dst_acc=x_acc)
        return Act(xn, 1, D)

    # -- whole network ----------------------------------------------------------------------
    def build(self) -> Program:
        cfg, B, T, F = self.cfg, self.B, self.T, self.F
        M, c = cfg.M, cfg.c
        x_in = Act(Ref("in"), F, 2 * M)
        # memory channel m*2+ri  <-  reference channel ri*M+m
        mem = np.arange(2 * M)
        in_perm = (mem % 2) * M + mem // 2

        skips: List[Act] = []
        x = x_in
        if cfg.is_u2:
            for i in range(4):
                x = self.unet_module(f"en.meta_unet_list.{i}", [x], 4 - i, False, in_perm if i == 0 else None)
                skips.append(x)
            g = self.conv2d_fwd("en.last_conv", [x], "en.last_conv.0.conv.1", True, "en.last_conv.1", "en.last_conv.2")
        else:
            # UNet_Encoder (EaBNet.py:234-239): the skips stay raw with their norm/PReLU pending -- the
            # next encoder layer and the decoder apply it while loading
            for i, (_, _, _, has_norm) in enumerate(unet_encoder_layers(cfg)):
                q = f"en.unet_list.{i}"
                g = self.conv2d_fwd(q, [x], f"{q}.0.conv.1", True, f"{q}.1" if has_norm else None,
                                    f"{q}.2" if has_norm else f"{q}.1", in_perm if i == 0 else None)
                if i < 4:
                    skips.append(g)
                    x = g
        if self.cln:
            x = g
            self.taps["en.last_conv"] = x
        else:
            x = self.materialise("en.last_conv", g)   # the S-TCMs need the bottleneck itself in memory
        skips.append(x)
        assert x.F * x.C == cfg.d_feat, "bottleneck width must equal d_feat"

        # S-TCN on the same memory viewed as [B][T][1][256]; channel f*64+c <- reference c*4+f
        Fb = x.F
        k = np.arange(cfg.d_feat)
        perm = (k % c) * Fb + k // c
        xt = Act(x.ref, 1, cfg.d_feat)
        x_acc = self.alloc_act(1, cfg.d_feat)
        self.ops.append(MemsetOp(x_acc, B * T * cfg.d_feat, B=B, T=T, row=cfg.d_feat, win=bool(self.chunk),
                                 name="stcns.acc0"))
        names = [(gi, i) for gi in range(cfg.q) for i in range(cfg.p)]
        have_in = None
        for k, (gi, i) in enumerate(names):
            nxt = f"stcns.{names[k + 1][0]}.tcm_list.{names[k + 1][1]}" if k + 1 < len(names) else None
            xt, have_in = self.tcm(f"stcns.{gi}.tcm_list.{i}", xt, 2 ** i, x_acc if i == cfg.p - 1 else None, perm, nxt, have_in)
            if gi == 0 and i == 0:
                self.taps["stcns.0.0"] = xt
        x = Act(x_acc, Fb, c)
        self.taps["stcns"] = x

        if cfg.is_u2:
            for i in range(4):
                x = self.unet_module(f"de.meta_unet_list.{i}", [x, skips[-(i + 1)]], i + 1, True)
            g = self.conv2d_transposed("de.last_conv", [x, skips[0]], "de.last_conv.0.conv.0", True,
                                       "de.last_conv.1", "de.last_conv.2")
        else:
            for i in range(5):                        # UNet_Decoder (EaBNet.py:324-328)
                q = f"de.unet_list.{i}"
                x = g = self.conv2d_transposed(q, [x, skips[-(i + 1)]], f"{q}.0.conv.0", True, f"{q}.1", f"{q}.2")
        if self.cln:
            e = g
            self.taps["de.last_conv"] = e
        else:
            e = self.materialise("de.last_conv", g)
        assert e.F == F and e.C == cfg.embed_dim == 64

        if not (cfg.topo_type == "mimo" and cfg.bf_type == "lstm"):
            # pointwise heads (EaBNet.py:78-81,111-123): the fused "last linear + filter-and-sum" kernel
            # with the 1x1 conv as its linear map.  cnn: plane m*2+ri is the layout the kernel expects.
            # miso: one complex mask on microphone 0 = zero rows for every other microphone (the
            # reference's frequency sum is applied by the caller on the (B,2,T,F) result).
            w2 = np.zeros((2 * M, 64), np.float32)
            b2 = np.zeros(2 * M, np.float32)
            wk = self.P["bf_map.weight"].reshape(-1, 64)
            w2[:wk.shape[0]], b2[:wk.shape[0]] = wk, self.P["bf_map.bias"]
            bfw = self.alloc(B * T * F * 2 * M) if self.dump_bfw else None
            if bfw is not None:
                self.taps["bf_w"] = Act(bfw, F, 2 * M)
            self.ops.append(BfwOp(y1=e.ref, w2=self.W.add("bf_map.weight#rows", w2), b2=self.W.add("bf_map.bias#rows", b2),
                                  x=Ref("in"), out=Ref("out"), bfw=bfw, B=B, T=T, F=F, M=M, win=bool(self.chunk),
                                  name="bf_map+fs"))
            self.flops += 2 * B * T * F * 64 * wk.shape[0]
            return Program(cfg, B, T, F, self.ops, self.W.flat(), self.act_size, self.taps, self.flops,
                           lanes=[0] * len(self.ops), chunk=self.chunk, zero_init=self.zero_init)

        # LSTM_BF (EaBNet.py:600-614)
        h = e
        for li, nm in enumerate(("rnn1", "rnn2")):
            p = f"bf_map.{nm}"
            wcat = np.concatenate([self.P[f"{p}.weight_ih_l0"], self.P[f"{p}.weight_hh_l0"]], axis=1)
            bias = self.P[f"{p}.bias_ih_l0"] + self.P[f"{p}.bias_hh_l0"]
            out = self.alloc_act(F, 64)
            self.ops.append(LstmOp(x=h.ref, ln_g=self.vec("bf_map.norm.weight") if li == 0 else None,
                                   ln_b=self.vec("bf_map.norm.bias") if li == 0 else None, ln_eps=EPS_LN,
                                   wcat=self.W.add(f"{p}#wcat", wcat), bias=self.W.add(f"{p}#bias", bias),
                                   h_out=out, B=B, T=T, F=F, name=p,
                                   precision=PREC_CODE[os.environ.get("EAB_LSTM_PREC", self.precision)],
                                   c_state=self.alloc(B * F * 64) if self.chunk else None, win=bool(self.chunk)))
            self.flops += 2 * B * T * F * 256 * 128
            h = Act(out, F, 64)
            self.taps[p] = h
        # w_dnn (Linear 64->64, ReLU, Linear 64->2M; EaBNet.py:594-596) + filter-and-sum: one kernel, the
        # hidden layer lives in LDS only
        bfw = self.alloc(B * T * F * 2 * M) if self.dump_bfw else None
        if bfw is not None:
            self.taps["bf_w"] = Act(bfw, F, 2 * M)
        self.ops.append(BfwOp(y1=h.ref, w1=self.vec("bf_map.w_dnn.0.weight"), b1=self.vec("bf_map.w_dnn.0.bias"),
                              w2=self.vec("bf_map.w_dnn.2.weight"), b2=self.vec("bf_map.w_dnn.2.bias"),
                              x=Ref("in"), out=Ref("out"), bfw=bfw, B=B, T=T, F=F, M=M, win=bool(self.chunk),
                              name="bf_map.w_dnn+fs"))
        self.flops += 2 * B * T * F * 64 * (64 + 2 * M)
        return Program(cfg, B, T, F, self.ops, self.W.flat(), self.act_size, self.taps, self.flops,
                       lanes=[0] * len(self.ops), chunk=self.chunk, zero_init=self.zero_init)


class GagLowering(Lowering):
    """GaGNet.forward (reference GaGNet.py:76-90) on the same op vocabulary.  Arenas: 'in' = inpt,
    'in2' = pre_x, both planar (B,2,T,F); 'out' = the q stage outputs [q][B][2][T][F] (the reference's
    (B,2,F,T) tensors are permuted views of it).  Every 1-D tensor is [B][T][1][C]."""
    spec_fn = staticmethod(gag_param_specs)
    parallel_chains = True        # False: the three S-TCM chains of a stage back to back (no graph branches)

    def tcm1(self, pre: str, x: Act, dilation: int, next_pre: Optional[str] = None, have_in: Optional[dict] = None):
        """GaGNet's single-branch SqueezedTCM (GaGNet.py:303-327): in_conv -> PReLU/norm/dilated conv ->
        PReLU/norm/out_conv + residual, three launches, InstanceNorm partials reduced by the consumer.  As in Lowering.tcm,
        the out_conv of one block and the in_conv of the next (`next_pre`) of a chain run as ONE small-tile launch
        (eab_conv_desc.f2_*); the next call gets the finished in_conv as `have_in`.  Returns (output, have_in or None)."""
        cfg, T, B = self.cfg, self.T, self.B
        D, cd, kd = cfg.d_feat, cfg.cd1, cfg.kd1
        bn = self.bn
        Kd = kd * ((cd + 15) // 16) * 16
        use_st = self.st and D == 256 and cd == 64 and Kd <= 256 and conv_tiles(T, 1, 32) <= 64
        if use_st:     # small-tile kernel (see Lowering.tcm)
            mt = None if bn else 64
            bm, bm_d, bm_out = self.pick_st_bm(1, cd, D, max_tiles=mt), self.pick_st_bm(1, cd, Kd, max_tiles=mt), self.pick_st_bm(1, D, cd)
        else:
            bm = bm_d = bm_out = 64
        tiles, tiles_d = conv_tiles(T, 1, bm), conv_tiles(T, 1, bm_d)
        assert bn or max(tiles, tiles_d) <= 64
        nD, nO = f"{pre}.d_conv.1", f"{pre}.out_conv.1"
        slD, slO = self.vec(f"{pre}.d_conv.0.weight"), self.vec(f"{pre}.out_conv.0.weight")
        if have_in is not None:                         # produced by the previous block's fused out_conv launch
            y, st, tiles = have_in["y"], have_in["st"], have_in["tiles"]
        else:
            wref = self.W.add(f"{pre}.in_conv.weight#packed", pack_taps(self.P[f"{pre}.in_conv.weight"], [0]))
            y = self.alloc_act(1, cd)
            st = None if bn else self.alloc(B * tiles * cd * 4)
            self.emit_conv(f"{pre}.in_conv", [x], wref, None, cd, D, 1, 1, 1, 0, 1, [0], [0], EPI_LINEAR, y,
                           st, 0 if bn else 1, (None, None) if bn else (slD, None), 0 if bn else tiles, 0, bm, st=use_st)
        span = (kd - 1) * dilation
        lead = span if cfg.is_causal else span // 2
        dts = [j * dilation - lead for j in range(kd)]
        wd = self.W.add(f"{pre}.d_conv.3.weight#packed", pack_taps(self.P[f"{pre}.d_conv.3.weight"], range(kd)))
        z = self.alloc_act(1, cd)
        st2 = None if bn else self.alloc(B * tiles_d * cd * 4)
        self.emit_conv(f"{pre}.d_conv", [Act(y, 1, cd, self.bn_xf(nD) if bn else None, slD, XF_PRELU_NORM)], wd, None, cd,
                       Kd, 1, 1, 1, 0, 1, dts, [0] * kd, EPI_LINEAR, z, st2, 0 if bn else 1,
                       (None, None) if bn else (slO, None), 0 if bn else tiles_d, 0, bm_d,
                       fin=None if bn else dict(stats=st, tiles=tiles, nsets=1, count=T, norms=[nD]), st=use_st)
        wo = self.W.add(f"{pre}.out_conv.2.weight#packed", pack_taps(self.P[f"{pre}.out_conv.2.weight"], [0]))
        xn = self.alloc_act(1, D)
        op = self.emit_conv(f"{pre}.out_conv", [Act(z, 1, cd, self.bn_xf(nO) if bn else None, slO, XF_PRELU_NORM)], wo, None,
                            D, cd, 1, 1, 1, 0, 1, [0], [0], EPI_ADD, xn, bm=bm_out, aux=x.ref,
                            fin=None if bn else dict(stats=st2, tiles=tiles_d, nsets=1, count=T, norms=[nO]), st=use_st)
        nxt = None
        tiles_n = conv_tiles(T, 1, bm_out)
        if use_st and next_pre is not None and self.precision == "f32" and self.fuse_out_in and (bn or tiles_n <= 64):
            op.f2_w = self.W.add(f"{next_pre}.in_conv.weight#frag2", pack_frag(pack_taps(self.P[f"{next_pre}.in_conv.weight"], [0])))
            op.f2_dst, op.f2_N = self.alloc_act(1, cd), cd
            if not bn:
                op.f2_stats, op.f2_nsets, op.f2_stat_tiles = self.alloc(B * tiles_n * cd * 4), 1, tiles_n
                op.f2_stat_slope0 = self.vec(f"{next_pre}.d_conv.0.weight")
            op.name = f"{pre}.out_conv+{next_pre}.in_conv"
            self.flops += 2 * B * T * cd * D
            nxt = dict(y=op.f2_dst, st=op.f2_stats, tiles=tiles_n)
        return Act(xn, 1, D), nxt

    def chain(self, pre: str, x: Act) -> Act:
        names = [(f"{pre}.{j}.tcns.{k}", d) for j in range(self.cfg.p) for k, d in enumerate(self.cfg.dilas)]
        have_in = None
        for i, (nm, d) in enumerate(names):
            x, have_in = self.tcm1(nm, x, d, names[i + 1][0] if i + 1 < len(names) else None, have_in)
        return x

    def gated_in(self, pfx: str, feat: Act, pre: Act, feat_perm: np.ndarray) -> Act:
        """in_conv_main(cat) * sigmoid(in_conv_gate(cat)) (GaGNet.py:191,251) as ONE gated 1x1 conv over the
        two sources; the concatenation (:190,250) is the kernel's two-pointer K loop."""
        D, Fq = self.cfg.d_feat, self.cfg.freq
        cols_pre = np.arange(2 * Fq)
        cols_pre = D + (cols_pre % 2) * Fq + cols_pre // 2          # memory channel f*2+ri <- reference ri*F+f
        w = np.concatenate([self.P[f"{pfx}.in_conv_main.weight"], self.P[f"{pfx}.in_conv_gate.0.weight"]], axis=0)[:, :, 0]
        wk = np.zeros((2 * D, D + GAG_PRE_LD), np.float32)
        wk[:, :D] = w[:, feat_perm]
        wk[:, D:D + 2 * Fq] = w[:, cols_pre]
        order = glu_row_order(2 * D)
        wref = self.W.add(f"{pfx}.in_conv#packed", pack_taps(wk[order][:, :, None], [0]))
        bias = np.concatenate([self.P[f"{pfx}.in_conv_main.bias"], self.P[f"{pfx}.in_conv_gate.0.bias"]])[order]
        dst = self.alloc_act(1, D)
        Kpad = ((D + GAG_PRE_LD + 15) // 16) * 16
        self.emit_conv(f"{pfx}.in_conv", [feat, pre], wref, self.W.add(f"{pfx}.in_conv.bias#packed", bias), 2 * D, Kpad,
                       1, 1, 1, 0, 1, [0], [0], EPI_GLU, dst, bm=64, patch_ok=False)
        return Act(dst, 1, D)

    def linear(self, key: str, x: Act) -> Ref:
        """Conv1d(d_feat -> 161, 1) (GaGNet.py:176,241), rows padded to GAG_LIN_LD."""
        D, Fq = self.cfg.d_feat, self.cfg.freq
        w = np.zeros((GAG_LIN_LD, D, 1), np.float32)
        b = np.zeros(GAG_LIN_LD, np.float32)
        w[:Fq], b[:Fq] = self.P[f"{key}.weight"], self.P[f"{key}.bias"]
        dst = self.alloc_act(1, GAG_LIN_LD)
        self.emit_conv(key, [x], self.W.add(f"{key}.weight#packed", pack_taps(w, [0])), self.W.add(f"{key}.bias#packed", b),
                       GAG_LIN_LD, D, 1, 1, 1, 0, 1, [0], [0], EPI_LINEAR, dst, bm=64)
        return dst

    def build(self) -> Program:
        cfg, B, T, F = self.cfg, self.B, self.T, self.F
        assert F == cfg.freq
        c = cfg.c
        enc_in = Act(self.alloc_act(F, 4), F, 4, raw=True)
        pre = Act(self.alloc(B * T * GAG_PRE_LD), 1, GAG_PRE_LD, raw=True)
        self.ops.append(GagPackOp(inpt=Ref("in"), pre_x=Ref("in2"), enc_in=enc_in.ref, pre=pre.ref, B=B, T=T, F=F,
                                  win=bool(self.chunk), name="pack"))
        x = enc_in
        if cfg.is_u2:
            for i in range(4):
                x = self.unet_module(f"en.meta_unet_list.{i}", [x], 4 - i, False)
            g = self.conv2d_fwd("en.last_conv", [x], "en.last_conv.0.conv.1", True, "en.last_conv.1", "en.last_conv.2")
        else:
            for i in range(5):
                q = f"en.unet_list.{i}"
                x = g = self.conv2d_fwd(q, [x], f"{q}.0.conv.1", True, f"{q}.1", f"{q}.2")
        x = self.materialise("en.last_conv", g)
        assert x.F * x.C == cfg.d_feat
        k = np.arange(cfg.d_feat)
        feat_perm = (k % c) * x.F + k // c                 # memory channel f*64+c <- reference c*4+f (GaGNet.py:83-84)
        feat = Act(x.ref, 1, cfg.d_feat)
        act = {"sigmoid": ACT_SIGMOID, "tanh": ACT_TANH, "relu": ACT_RELU}[cfg.acti_type]
        for gi in range(cfg.q):
            gl, gz = f"gags.{gi}.glance_block", f"gags.{gi}.gaze_block"
            # the glance chain and the gaze chain(s) only meet again in the tail: they run as parallel
            # branches (each S-TCM launch fills less than half of the chip on its own)
            xg0 = self.gated_in(gl, feat, pre, feat_perm)
            xz = self.gated_in(gz, feat, pre, feat_perm)
            branches = ([1] if cfg.is_squeezed else [1, 2]) if self.parallel_chains else []
            if branches:
                self.mark("fork", branches)
            gain = self.linear(f"{gl}.linear_g.0", self.chain(f"{gl}.tcn_g", xg0))
            if branches:
                self.set_lane(1)
            if cfg.is_squeezed:
                xr = self.chain(f"{gz}.tcm_ri", xz)
                lr, li = self.linear(f"{gz}.linear_r", xr), self.linear(f"{gz}.linear_i", xr)
            else:
                lr = self.linear(f"{gz}.linear_r", self.chain(f"{gz}.tcm_r", xz))
                if branches:
                    self.set_lane(2)
                li = self.linear(f"{gz}.linear_i", self.chain(f"{gz}.tcm_i", xz))
            if branches:
                self.set_lane(0)
                self.mark("join", branches)
            nxt = Act(self.alloc(B * T * GAG_PRE_LD), 1, GAG_PRE_LD, raw=True)
            self.ops.append(GagCrmOp(pre=pre.ref, g=gain, r=lr, i=li, pre_out=nxt.ref, planar=Ref("out", gi * B * 2 * T * F),
                                     B=B, T=T, F=F, act=act, win=bool(self.chunk), name=f"gags.{gi}.crm"))
            pre = nxt
        return Program(cfg, B, T, F, self.ops, self.W.flat(), self.act_size, self.taps, self.flops,
                       lanes=self.lane_of_ops(), sync=self.sync, chunk=self.chunk, zero_init=self.zero_init)


def lower(cfg, params: Dict[str, np.ndarray], B: int, T: int, F: int = 161,
          dump_bfw: bool = False, precision: str = "f32", chunk: int = 0, parallel_chains: bool = True) -> Program:
    """chunk > 0 lowers the streaming form: T is then the longest utterance the resident activations can
    hold and every op advances `chunk` frames per replay.  parallel_chains (GaGNet): the glance / gaze S-TCM
    chains of a stage as parallel graph branches (best latency of one batch) or back to back (best when
    several batches are in flight anyway, eabnet_amd.Pipeline)."""
    if isinstance(cfg, GagConfig):
        low = GagLowering(cfg, params, B, T, F, False, precision, chunk)
        low.parallel_chains = parallel_chains
        return low.build()
    return Lowering(cfg, params, B, T, F, dump_bfw, precision, chunk).build()
