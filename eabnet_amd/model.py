"""Drop-in boundary: ``EaBNet`` (reference EaBNet.py:9-125), ``prepare_data``
(reference train_distributed.py:68-95), ``numParams`` (EaBNet.py:653-659) and
``com_mag_mse_loss`` (EaBNet.py:627-640) with the reference's names, argument
meaning and tensor signatures, executing on libeabnet_hip.so.

The module owns ordinary ``nn.Parameter``s under the reference's state-dict
keys (eabnet_amd/spec.py), so reference checkpoints load with
``strict=True``.  ``forward`` lowers the network once per (B, T, parameter
version) to an op program (eabnet_amd/program.py), binds it to device
addresses and replays it with ONE foreign call per forward.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from . import program as prg
from .graphs import LaneGraphs, plan_segments, single_lane
from .spec import GagConfig, NetConfig, ParamSpec, gag_param_specs, param_specs


# ----------------------------------------------------------------------------
# parameter container with the reference's dotted names
# ----------------------------------------------------------------------------
class _Scope(nn.Module):
    """Anonymous container: only exists so that ``state_dict()`` produces the
    dotted keys of the reference's module tree."""


def _default_init(spec: ParamSpec) -> torch.Tensor:
    """PyTorch's default initialisers for the layer types the reference uses."""
    t = torch.empty(spec.shape, dtype=torch.float32)
    k = spec.kind
    if k in ("conv_w", "convT_w", "lin_w", "bias", "lstm"):
        a = 1.0 / math.sqrt(max(spec.fan_in, 1))      # kaiming_uniform(a=sqrt 5) == U(-1/sqrt(fan_in), +)
        return t.uniform_(-a, a)
    if k in ("norm_w", "ln_w"):
        return t.fill_(1.0)
    if k in ("norm_b", "ln_b"):
        return t.zero_()
    if k == "prelu":
        return t.fill_(0.25)
    if k == "bn_mean":
        return t.zero_()
    if k == "bn_var":
        return t.fill_(1.0)
    if k == "bn_count":
        return torch.zeros((), dtype=torch.long)
    raise ValueError(k)


def _attach(root: nn.Module, dotted: str, p: torch.Tensor) -> None:
    """Register a parameter (nn.Parameter) or a buffer (plain tensor: BatchNorm statistics)."""
    node = root
    parts = dotted.split(".")
    for name in parts[:-1]:
        child = node._modules.get(name)
        if child is None:
            child = _Scope()
            node.add_module(name, child)
        node = child
    if isinstance(p, nn.Parameter):
        node.register_parameter(parts[-1], p)
    else:
        node.register_buffer(parts[-1], p)


# ----------------------------------------------------------------------------
# bound program
# ----------------------------------------------------------------------------
class _Bound:
    """A lowered program with device arenas and the ctypes op array."""

    def __init__(self, prog: prg.Program, device: torch.device):
        self.prog = prog
        self.device = device
        self.weights = torch.from_numpy(prog.weights).to(device)
        self.acts = torch.empty(max(prog.act_floats, 1), dtype=torch.float32, device=device)
        self.reset_counters()
        self.ops = (_lib.Op * len(prog.ops))()
        # streaming programs: the frame position every windowed op reads (device memory, so one captured
        # graph serves all chunks)
        self.t_pos = torch.zeros(1, dtype=torch.int32, device=device) if prog.chunk else None
        self._in_ptr = None
        self._out_ptr = None
        # hipGraph replay (optional): static boundary buffers + the captured program
        self.graph = None
        self.static_in = None
        self.static_in2 = None
        self.static_out = None
        self.graph_failed = False

    def reset_counters(self) -> None:
        """arrival counters of the fused InstanceNorm finalisation: zero before the first run (the kernels re-arm them
        after every complete run; needed again only if the arena was overwritten from outside, as the tests do)"""
        for ref, n in self.prog.zero_init:
            self.acts[ref.off:ref.off + n].zero_()

    def __del__(self):
        # The arenas may have been used on streams other than the one they were allocated under (Pipeline slot
        # streams, hipGraph replays): the caching allocator would hand the blocks out again while such work is
        # still queued.  Dropping a bound program is rare (shape / precision change), so drain the device first.
        try:
            if torch.cuda.is_available():
                torch.cuda.synchronize(self.device)
        except Exception:                                 # noqa: BLE001 - interpreter shutdown
            pass

    def capture(self, in_shape, out_shape, in2_shape=None) -> bool:
        """Capture the whole op program into a hipGraph bound to static in/out buffers.  Replaying it
        costs one graph launch instead of ~250 kernel launches enqueued by the host.  Returns False
        (and stays on direct launches) if the runtime refuses the capture."""
        if self.graph is not None or self.graph_failed:
            return self.graph is not None
        try:
            self.static_in = torch.empty(in_shape, dtype=torch.float32, device=self.device)
            self.static_out = torch.empty(out_shape, dtype=torch.float32, device=self.device)
            if in2_shape is not None:
                self.static_in2 = torch.empty(in2_shape, dtype=torch.float32, device=self.device)
            self.bind(self.static_in.data_ptr(), self.static_out.data_ptr(),
                      self.static_in2.data_ptr() if in2_shape is not None else None)
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                 # warm-up outside the capture
                self.static_in.zero_()
                if self.static_in2 is not None:
                    self.static_in2.zero_()
                self._launch(side.cuda_stream, 0, len(self.prog.ops))      # (program order on one stream: no lanes to set up)
            torch.cuda.current_stream().wait_stream(side)
            # one single-stream hipGraph per lane segment (graphs.py: a hipGraph with internal branches can crash the HIP
            # runtime at replay, depending on streams created elsewhere in the process); programs without parallel
            # branches are one segment, i.e. one graph
            lg = LaneGraphs(self.device, self._plan(), self._launch)
            lg.capture()
            self.graph = lg
        except Exception as e:                            # noqa: BLE001 - any capture failure -> direct launches
            import warnings
            warnings.warn(f"eabnet_amd: hipGraph capture failed ({e!r}); using direct launches")
            self.graph, self.graph_failed = None, True
            self._in_ptr = None
        return self.graph is not None

    def update_weights(self, flat: np.ndarray) -> None:
        self.weights.copy_(torch.from_numpy(flat), non_blocking=False)

    def _addr(self, ref: Optional[prg.Ref], bases) -> Optional[int]:
        if ref is None:
            return None
        return bases[ref.arena] + 4 * ref.off

    def bind(self, in_ptr: int, out_ptr: int, in2_ptr: Optional[int] = None) -> None:
        if (in_ptr, out_ptr, in2_ptr) == self._in_ptr:
            return
        bases = {"w": self.weights.data_ptr(), "a": self.acts.data_ptr(), "in": in_ptr, "out": out_ptr, "in2": in2_ptr}
        A = lambda r: self._addr(r, bases)  # noqa: E731
        for k, op in enumerate(self.prog.ops):
            o = self.ops[k]
            o.kind = op.kind
            if getattr(op, "win", False):
                w = o.conv.win if op.kind == prg.OP_CONV else o.win
                w.pos, w.count = self.t_pos.data_ptr(), self.prog.chunk
            if op.kind == prg.OP_CONV:
                d = o.conv
                for f in ("src0", "src1", "xf0", "xf1", "slope0", "slope1", "w", "bias", "aux", "dst", "dst_acc",
                          "stats", "stat_slope0", "stat_slope1", "fin_stats", "fin_gamma0", "fin_beta0",
                          "fin_gamma1", "fin_beta1", "fz_counter", "fz_gamma0", "fz_beta0", "fz_xf0", "fz_gamma1", "fz_beta1",
                          "fz_xf1"):
                    setattr(d, f, A(getattr(op, f)))
                d.fz_eps = float(op.fz_eps)
                for f in ("C0", "C1", "xf_mode", "N", "Kpad", "B", "T", "Fin", "Fout", "No", "ostride", "ophase",
                          "istride", "epi", "Cout", "nsets", "stat_tiles", "stat_tile0", "bm", "fin_tiles",
                          "fin_nsets", "fin_count", "precision", "korder", "p2_mask1"):
                    setattr(d, f, int(getattr(op, f)))
                d.fin_eps = float(op.fin_eps)
                d.ntaps = len(op.dt)
                for j in range(_lib.MAX_TAPS):
                    d.dt[j] = op.dt[j] if j < len(op.dt) else 0
                    d.ioff[j] = op.ioff[j] if j < len(op.ioff) else 0
                # second output-column phase served by the same launch (small-tile kernel only)
                d.ph1_w = A(op.ph1_w)
                d.ph1_No, d.ph1_ophase, d.ph1_Kpad, d.ph1_ntaps = int(op.ph1_No), int(op.ph1_ophase), int(op.ph1_Kpad), len(op.ph1_dt)
                for j in range(_lib.MAX_TAPS):
                    d.ph1_dt[j] = op.ph1_dt[j] if j < len(op.ph1_dt) else 0
                    d.ph1_ioff[j] = op.ph1_ioff[j] if j < len(op.ph1_ioff) else 0
                # fused second 1x1 convolution (out_conv of one S-TCM + in_conv of the next)
                for f in ("f2_w", "f2_dst", "f2_stats", "f2_stat_slope0", "f2_stat_slope1"):
                    setattr(d, f, A(getattr(op, f)))
                d.f2_N, d.f2_nsets, d.f2_stat_tiles = int(op.f2_N), int(op.f2_nsets), int(op.f2_stat_tiles)
            elif op.kind == prg.OP_IN_FINALIZE:
                o.i[0:5] = [op.B, op.C, op.nsets, op.stat_tiles, op.count]
                o.f[0] = op.eps
                for j, r in enumerate((op.stats, op.gamma0, op.beta0, op.xf0, op.gamma1, op.beta1, op.xf1)):
                    o.p[j] = A(r)
            elif op.kind == prg.OP_NORM_ACT:
                o.i[0:4] = [op.B, op.P, op.C, op.T]
                for j, r in enumerate((op.a, op.xfa, op.slopea, op.b, op.xfb, op.slopeb, op.out)):
                    o.p[j] = A(r)
            elif op.kind == prg.OP_LSTM64:
                o.i[0:4] = [op.B, op.T, op.F, op.precision]
                o.f[0] = op.ln_eps
                for j, r in enumerate((op.x, op.ln_g, op.ln_b, op.wcat, op.bias, op.h_out, op.c_state)):
                    o.p[j] = A(r)
            elif op.kind == prg.OP_BFW_FS:
                o.i[0:4] = [op.B, op.T, op.F, op.M]
                for j, r in enumerate((op.y1, op.w2, op.b2, op.x, op.out, op.bfw, op.w1, op.b1)):
                    o.p[j] = A(r)
            elif op.kind == prg.OP_MEMSET0:
                nbytes = 4 * op.nfloats
                o.i[0] = C.c_int32(nbytes & 0xFFFFFFFF).value
                o.i[1] = nbytes >> 32
                o.i[2:5] = [op.B, op.T, op.row]
                o.p[0] = A(op.ptr)
            elif op.kind == prg.OP_CLN_STATS:
                o.i[0:4] = [op.B, op.T, op.P, op.C]
                o.f[0] = op.eps
                for j, r in enumerate((op.x, op.slope, op.sums, op.state, op.mr)):
                    o.p[j] = A(r)
            elif op.kind == prg.OP_CLN_APPLY:
                o.i[0:5] = [op.B, op.T, op.P, op.C, op.mode]
                for j, r in enumerate((op.x, op.mr, op.gain, op.bias, op.slope, op.add, op.out)):
                    o.p[j] = A(r)
            elif op.kind == prg.OP_GATE_ROWS:
                o.i[0:3] = [op.B, op.T, op.row]
                for j, r in enumerate((op.a, op.r, op.z)):
                    o.p[j] = A(r)
            elif op.kind == prg.OP_GAG_PACK:
                o.i[0:4] = [op.B, op.T, op.F, prg.GAG_PRE_LD]
                for j, r in enumerate((op.inpt, op.pre_x, op.enc_in, op.pre)):
                    o.p[j] = A(r)
            elif op.kind == prg.OP_GAG_CRM:
                o.i[0:6] = [op.B, op.T, op.F, prg.GAG_PRE_LD, prg.GAG_LIN_LD, op.act]
                for j, r in enumerate((op.pre, op.g, op.r, op.i, op.pre_out, op.planar)):
                    o.p[j] = A(r)
            else:
                raise ValueError(op.kind)
        self._in_ptr, self._out_ptr = (in_ptr, out_ptr, in2_ptr), None
        self._plan_chains()

    def _plan_chains(self) -> None:
        """Streaming programs: runs of consecutive small-tile convolutions that give every utterance ONE tile (the S-TCN of a
        frame-synchronous step) are executed by ONE launch (eab_conv_st_chain_run, csrc/conv_st.hip): one workgroup per
        utterance walks the run's descriptors, which are uploaded to device memory here.  self.exec_ops = the op array of a
        whole-program run with every such run replaced by its chain op (self.ops keeps one entry per program op for
        single-op debug launches).  EAB_ST_CHAIN=0 keeps separate launches."""
        import os
        self.exec_ops, self.n_exec, self.chains = None, 0, []
        if not self.prog.chunk:
            return
        lib = _lib.load()
        ops = self.prog.ops
        n = len(ops)
        use_chain = os.environ.get("EAB_ST_CHAIN", "1") != "0"
        # cLN, one frame per step: the statistics / scan / apply launches of a unit (eab_cln_stats_f32 = two launches, then
        # eab_cln_apply_f32) become ONE launch (eab_cln_step_f32: one workgroup per utterance sums the new frame, advances the
        # running sums and normalises the frame -- the same code paths, so the same bits).  EAB_CLN_STEP=0 keeps three launches.
        fuse_cln = self.prog.chunk == 1 and os.environ.get("EAB_CLN_STEP", "1") != "0"
        def plan(first, cnt):
            descs = (_lib.ConvDesc * cnt)(*[self.ops[first + t].conv for t in range(cnt)])
            codes = (C.c_int * cnt)()
            lds, bf = C.c_int(0), C.c_int(0)
            if lib.eab_conv_st_chain_plan(descs, cnt, codes, C.byref(lds), C.byref(bf)) != 0:
                return None
            return first, cnt, descs, codes, lds.value, bf.value
        # every op on its own first (is it a single-tile launch of a form the chain kernel carries, and in which precision),
        # then maximal runs of such ops, planned as a whole
        single = [plan(k, 1) if (use_chain and ops[k].kind == prg.OP_CONV and ops[k].korder == prg.KORDER_FRAG) else None
                  for k in range(n)]
        runs = []                                       # (first, count, descs, codes, lds, bf)
        k = 0
        while k < n:
            if single[k] is None:
                k += 1
                continue
            j = k
            while j < n and single[j] is not None and single[j][5] == single[k][5]:
                j += 1
            got = plan(k, j - k) if j - k >= 2 else None
            if got:
                runs.append(got)
            k = j
        def cln_pair(k):
            a, b = ops[k], ops[k + 1] if k + 1 < n else None
            return (fuse_cln and b is not None and a.kind == prg.OP_CLN_STATS and b.kind == prg.OP_CLN_APPLY and a.x == b.x
                    and a.mr == b.mr and (a.B, a.T, a.P, a.C) == (b.B, b.T, b.P, b.C) and a.win and b.win and a.state is not None)
        if not runs and not any(cln_pair(k) for k in range(n - 1)):
            return
        exec_list = []
        k = 0
        ri = 0
        while k < n:
            if cln_pair(k):
                a, b = self.ops[k], self.ops[k + 1]
                o = _lib.Op()
                o.kind = prg.OP_CLN_STEP
                for j in range(5):
                    o.p[j] = a.p[j]                           # x, slope of the statistics, sums, state, mr
                for j, src in enumerate((2, 3, 4, 5, 6)):
                    o.p[5 + j] = b.p[src]                     # gain, bias, slope, add, y
                o.i[0:5] = [b.i[0], b.i[1], b.i[2], b.i[3], b.i[4]]
                o.f[0] = a.f[0]
                o.win.pos, o.win.count = a.win.pos, a.win.count
                exec_list.append(o)
                k += 2
                continue
            if ri < len(runs) and runs[ri][0] == k:
                first, cnt, descs, codes, lds, bf = runs[ri]
                raw = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(self.device)
                dev_codes = torch.tensor(list(codes), dtype=torch.int32, device=self.device)
                self.chains.append((first, cnt, raw, dev_codes))          # keep the device copies alive
                o = _lib.Op()
                o.kind = prg.OP_CONV_CHAIN
                o.p[0], o.p[1] = raw.data_ptr(), dev_codes.data_ptr()
                o.i[0:4] = [cnt, int(self.prog.B), lds, bf]
                exec_list.append(o)
                k += cnt
                ri += 1
            else:
                exec_list.append(self.ops[k])
                k += 1
        self.exec_ops = (_lib.Op * len(exec_list))(*exec_list)
        self.n_exec = len(exec_list)

    def _launch(self, stream: int, first: int, n: int) -> None:
        if first == 0 and n == len(self.prog.ops) and getattr(self, "exec_ops", None) is not None:
            _lib.check(_lib.load().eab_run_program(self.exec_ops, self.n_exec, C.c_void_p(stream)), "eab_run_program")
            return
        ops = C.cast(C.byref(self.ops, first * C.sizeof(_lib.Op)), C.POINTER(_lib.Op))
        _lib.check(_lib.load().eab_run_program(ops, n, C.c_void_p(stream)), "eab_run_program")

    def _plan(self) -> list:
        n = len(self.prog.ops)
        if self.prog.sync and graph_branches_allowed():
            return plan_segments(n, self.prog.lanes, self.prog.sync)
        return single_lane(n)

    def run(self, stream: int, first: int = 0, count: Optional[int] = None) -> None:
        """Enqueue ops [first, first+count) on ``stream`` (a raw hipStream_t) with direct kernel launches.  Whole programs
        with parallel branches (prog.sync) fork onto side streams with events, exactly as their captured form replays
        (graphs.LaneGraphs)."""
        n = len(self.prog.ops) - first if count is None else count
        if not self.prog.sync or count is not None or not graph_branches_allowed():
            return self._launch(stream, first, n)         # single lane (or a single-op debug launch)
        assert torch.cuda.current_stream().cuda_stream == stream, "multi-lane programs run on torch's current stream"
        if getattr(self, "_direct", None) is None:
            self._direct = LaneGraphs(self.device, self._plan(), self._launch)
        self._direct.run_direct()

    def view(self, act: prg.Act) -> torch.Tensor:
        """Debug view of a named activation as (B, T, F, C)."""
        p = self.prog
        n = p.B * p.T * act.F * act.C
        assert act.ref.arena == "a"
        return self.acts[act.ref.off:act.ref.off + n].view(p.B, p.T, act.F, act.C)


# ----------------------------------------------------------------------------
# the module
# ----------------------------------------------------------------------------
def graph_branches_allowed() -> bool:
    """Parallel branches (side streams) for programs that mark independent chains?  EAB_GRAPH_BRANCHES=0 runs every program
    on one stream.  (Rounds 1-3 also switched them off while a torch.distributed process group was alive: a hipGraph with
    internal branches could crash the HIP runtime at replay.  The cause is an unchecked index in the runtime's stream
    assignment, graphs.py; since branches replay as separate single-stream graphs the condition no longer exists.)"""
    import os
    return os.environ.get("EAB_GRAPH_BRANCHES", "1") != "0"


def _refuse_differentiable(module: nn.Module, what: str, reason: str):
    """A differentiable call the HIP training programs do not cover is refused, never served by another backend: the package
    has one (the PyTorch-ROCm operator evaluation that used to sit here is now test infrastructure, tests/operator_path.py)."""
    return NotImplementedError(
        f"{type(module).__name__}: {what} ({reason}).  eabnet_amd runs differentiable calls on its HIP training programs "
        "(eabnet_amd/train.py, train_gag.py) and has no operator fallback; call under torch.no_grad() for inference.")


class _HipModule(nn.Module):
    """Shared machinery of the two networks: reference-keyed parameters, the lowered-program cache
    (one resident shape, re-packed when a parameter or buffer changes) and the replay knobs."""

    def _init_params(self, specs) -> None:
        self._specs = specs
        for key, spec in specs.items():
            t = _default_init(spec)
            _attach(self, key, t if spec.is_buffer else nn.Parameter(t))
        self._bound: Dict[tuple, _Bound] = {}
        self._packed_version: Dict[tuple, tuple] = {}
        self.dump_bfw = False                         # tests: also emit the (B,T,F,M,2) beam-forming weights
        # replay the lowered program as ONE hipGraph launch (static internal in/out buffers, one
        # device copy of the input and of the output per call); False = direct kernel launches
        self.use_graph = True
        # arithmetic of the MFMA contractions: "f32" = exact fp32 MFMA; "f16x3" = error-compensated
        # fp16 split on the f16 matrix cores, same end-to-end error class as fp32;
        # "bf16" = operands rounded to bf16, fp32 accumulate / norms / activations (BASELINE configs[3]/[4];
        # outside the 1e-4 bar: ~1e-2)
        self.precision = "f32"
        # "hip" once a differentiable forward has run (train.py / train_gag.py programs -- the only backend); None before
        self.training_backend = None

    def _param_fingerprint(self) -> tuple:
        """Change detector for the packed weights (runs on every forward).  Every parameter / buffer slot
        contributes (storage address, version counter): in-place updates bump the version, ``.data =``
        re-assignment, a device move or ``load_state_dict(assign=True)`` change the address or the object in the
        slot.  The slot list (owning module, name) is cached -- walking the ~800-node module tree costs more
        than the rest of the host side of a step; looking the ~500 slots up does not."""
        slots = self.__dict__.get("_slot_list")
        if slots is None:
            slots = []
            for mod in self.modules():
                slots += [(mod._parameters, n) for n in mod._parameters] + [(mod._buffers, n) for n in mod._buffers]
            self.__dict__["_slot_list"] = slots
        fp = []
        for d, n in slots:
            t = d[n]
            fp.append((t.data_ptr(), t._version))
        return tuple(fp)

    def _numpy_params(self) -> Dict[str, np.ndarray]:
        sd = self.state_dict()
        return {k: sd[k].detach().to("cpu", torch.float32).numpy() for k, s in self._specs.items()
                if s.kind != "bn_count"}

    def _program(self, B: int, T: int, F: int, device: torch.device) -> _Bound:
        chains = bool(self.__dict__.get("parallel_chains", True))
        key = (B, T, F, str(device), self.precision, chains)
        fp = self._param_fingerprint()
        bound = self._bound.get(key)
        if bound is None or ("bf_w" in bound.prog.taps) != self.dump_bfw:
            prog = prg.lower(self.cfg, self._numpy_params(), B, T, F, dump_bfw=self.dump_bfw,
                             precision=self.precision, parallel_chains=chains)
            bound = _Bound(prog, device)
            self._bound = {key: bound}                # keep one shape resident (activations can be GBs)
            self._packed_version = {key: fp}
        elif self._packed_version.get(key) != fp:
            prog = prg.lower(self.cfg, self._numpy_params(), B, T, F, dump_bfw=self.dump_bfw,
                             precision=self.precision, parallel_chains=chains)
            bound.update_weights(prog.weights)
            self._packed_version[key] = fp
        return bound

    # -- streaming ------------------------------------------------------------------
    def stream_begin(self, B: int, T_max: int, chunk: int = 1, F: int = 161, device=None) -> EaBNetStream:
        """Frame-synchronous inference (BASELINE config 5): returns a stream object whose
        ``step`` takes ``chunk`` new frames and returns the matching output frames (see EaBNetStream.step).
        Needs a configuration in which the network really is causal -- ``norm_type="BN"`` in eval mode (running
        statistics) or ``norm_type="cLN"`` (cumulative statistics), and ``is_causal=True`` -- and raises
        NotImplementedError otherwise."""
        if self.training:
            raise RuntimeError("stream_begin: call .eval() first (BatchNorm must use its running statistics)")
        _lib.load()
        device = torch.device(device) if device is not None else next(self.parameters()).device
        if device.type != "cuda":
            raise _lib.EabError("streaming inference runs on MI355X only: move the module to 'cuda'")
        if chunk < 1 or T_max < 1:
            raise ValueError("chunk and T_max must be positive")
        with torch.cuda.device(device):
            prog = prg.lower(self.cfg, self._numpy_params(), B, T_max, F, precision=self.precision, chunk=chunk)
            return EaBNetStream(self, _Bound(prog, device), B, T_max, F, chunk)

    def _needs_graph(self, *inputs) -> bool:
        needs = torch.is_grad_enabled() and (any(x.requires_grad for x in inputs)
                                             or any(p.requires_grad for p in self.parameters()))
        return needs or (self.norm_type == "BN" and self.training)


class EaBNet(_HipModule):
    """MI355X implementation of the reference ``EaBNet`` (EaBNet.py:9-125).

    Same constructor keywords and defaults, same ``forward`` signature:
    ``inpt`` (B, T, F, M, 2) [or (B, T, F, 2) for one microphone] ->
    (B, 2, T, F), same state-dict keys.  ``torch.no_grad()`` / ``requires_grad=False`` calls run
    the hand-written HIP inference program; a call that must be differentiable (training: train.py /
    train_distributed.py:218-230) runs the HIP training programs (eabnet_amd/train.py: forward and
    backward as two static op programs behind one autograd node), so ``loss.backward()``, the optimiser,
    ``clip_grad_norm_`` and DistributedDataParallel work on the ordinary ``nn.Parameter``s.  CUDA
    tensors only; there is no second backend.  Refused (NotImplementedError): a gradient w.r.t. the input
    spectrogram, BatchNorm in eval mode under autograd, more than 32 microphones.
    """

    def __init__(self, k1: tuple = (2, 3), k2: tuple = (1, 3), c: int = 64, M: int = 9, embed_dim: int = 64,
                 kd1: int = 5, cd1: int = 64, d_feat: int = 256, p: int = 6, q: int = 3, is_causal: bool = True,
                 is_u2: bool = True, bf_type: str = "lstm", topo_type: str = "mimo", intra_connect: str = "cat",
                 norm_type: str = "IN"):
        super().__init__()
        self.k1, self.k2, self.c, self.M, self.embed_dim = tuple(k1), tuple(k2), c, M, embed_dim
        self.kd1, self.cd1, self.d_feat, self.p, self.q = kd1, cd1, d_feat, p, q
        self.is_causal, self.is_u2, self.bf_type = is_causal, is_u2, bf_type
        self.topo_type, self.intra_connect, self.norm_type = topo_type, intra_connect, norm_type
        self.cfg = NetConfig(k1=tuple(k1), k2=tuple(k2), c=c, M=M, embed_dim=embed_dim, kd1=kd1, cd1=cd1,
                             d_feat=d_feat, p=p, q=q, is_causal=is_causal, is_u2=is_u2, bf_type=bf_type,
                             topo_type=topo_type, intra_connect=intra_connect, norm_type=norm_type)
        self._init_params(param_specs(self.cfg))     # raises NotImplementedError for unsupported topologies

    # -- forward -------------------------------------------------------------------
    def forward(self, inpt: torch.Tensor) -> torch.Tensor:
        """:param inpt: (B, T, F, M, 2) compressed multichannel spectrogram
        :return: beamformed estimate (B, 2, T, F)   (reference EaBNet.py:88-117)"""
        if inpt.ndim == 4:
            inpt = inpt.unsqueeze(-2)
        if inpt.ndim != 5 or inpt.shape[-1] != 2 or inpt.shape[-2] != self.M:
            raise ValueError(f"expected (B,T,F,{self.M},2), got {tuple(inpt.shape)}")
        if self._needs_graph(inpt):
            # training: forward AND backward on the hand-written kernels (train.py: two static op programs behind one
            # autograd node) for every constructor branch; BatchNorm = train mode (batch statistics + buffer update)
            from . import train
            if not (inpt.is_cuda and next(self.parameters()).is_cuda):
                raise _lib.EabError("eabnet_amd.EaBNet trains on MI355X only: move the input and the module to 'cuda'. "
                                    "There is no CPU fallback by design.")
            if inpt.requires_grad:
                raise _refuse_differentiable(self, "the input requires grad", "the training programs produce parameter "
                                             "gradients only, as the reference's training loop needs")
            if self.norm_type == "BN" and not self.training:
                raise _refuse_differentiable(self, "BatchNorm in eval mode under autograd", "the training programs implement "
                                             "BatchNorm's train mode; call .train(), or torch.no_grad() for inference")
            if not train.supported(self.cfg):
                raise _refuse_differentiable(self, "topology outside the training programs", train.unsupported_reason(self.cfg))
            self.training_backend = "hip"
            out = train.forward_train(self, inpt)
            return out.sum(dim=-1) if self.topo_type == "miso" else out      # (EaBNet.py:122-123, as in inference below)
        if not inpt.is_cuda:
            raise _lib.EabError("eabnet_amd.EaBNet inference runs on MI355X only: move the input (and module) to "
                                "'cuda'. There is no CPU fallback by design.")
        _lib.load()
        B, T, F, M, _ = inpt.shape
        x = inpt.detach().to(torch.float32).contiguous()
        with torch.cuda.device(x.device):
            bound = self._program(B, T, F, x.device)
            if self.use_graph and not self.dump_bfw and not torch.cuda.is_current_stream_capturing() \
                    and bound.capture((B, T, F, M, 2), (B, 2, T, F)):
                bound.static_in.copy_(x, non_blocking=True)
                bound.graph.replay()
                out = bound.static_out.clone()
            else:
                out = torch.empty((B, 2, T, F), dtype=torch.float32, device=x.device)
                bound.bind(x.data_ptr(), out.data_ptr())
                bound.run(torch.cuda.current_stream().cuda_stream)
        self._last = (bound, x)                       # keep the input alive until the stream has consumed it
        if self.topo_type == "miso":
            # the reference reduces the masked reference-mic spectrum over frequency (EaBNet.py:122-123:
            # ``.sum(dim=-1)`` on a (B,T,F) tensor) and returns (B,2,T); kept as is
            out = out.sum(dim=-1)
        return out.to(inpt.dtype)


class EaBNetStream:
    """Frame-synchronous inference state of one batch of utterances (``EaBNet.stream_begin``,
    ``GaGNet.stream_begin``).

    The activations of the whole utterance stay resident in HBM ([B][T_max][..] per layer; 288 GB make
    that cheap) and are the state: each ``step`` appends ``chunk`` frames of input, replays the captured
    program restricted to those time rows (eab_time_window: every kernel reads the frame position from
    device memory) and returns the new output frames.  Results are bit-identical to one offline call
    on the concatenated input, because every kernel computes a row independently of the tile or launch
    it falls into and the LSTM state is carried exactly."""

    def __init__(self, net, bound: _Bound, B: int, T_max: int, F: int, chunk: int):
        self.net, self.bound, self.B, self.T_max, self.F, self.chunk = net, bound, B, T_max, F, chunk
        self.pos = 0
        self._closed = False
        self.post = isinstance(net, GaGNet)           # post-filter: two planar inputs, q stage outputs
        if self.post:
            in_shape, out_shape, in2_shape = (B, 2, T_max, F), (net.q, B, 2, T_max, F), (B, 2, T_max, F)
        else:
            in_shape, out_shape, in2_shape = (B, T_max, F, net.M, 2), (B, 2, T_max, F), None
        if not (net.use_graph and bound.capture(in_shape, out_shape, in2_shape)):
            self._in = torch.zeros(in_shape, dtype=torch.float32, device=bound.device)
            self._out = torch.zeros(out_shape, dtype=torch.float32, device=bound.device)
            self._in2 = torch.zeros(in2_shape, dtype=torch.float32, device=bound.device) if in2_shape else None
            bound.bind(self._in.data_ptr(), self._out.data_ptr(), self._in2.data_ptr() if in2_shape else None)
        else:
            self._in, self._out, self._in2 = bound.static_in, bound.static_out, bound.static_in2

    def reset(self) -> None:
        """Start a new batch of utterances (no buffer needs clearing: position 0 ignores all state)."""
        self.pos, self._closed = 0, False

    def step(self, x: torch.Tensor, pre_x: Optional[torch.Tensor] = None):
        """Beam-former: x (B, n, F, M, 2) new frames -> (B, 2, n, F) [(B, 2, n) for topo_type='miso'].
        Post-filter: x, pre_x (B, 2, n, F) -> list of q (B, 2, F, n).  n == chunk, except for the LAST step
        of an utterance (n < chunk): the recurrent state then sits past the end, so the stream accepts no
        further frames until ``reset``."""
        tdim = 2 if self.post else 1
        n = x.shape[tdim]
        if self._closed:
            raise RuntimeError("the previous step was a short final chunk: call reset() before the next utterance")
        want = (self.B, 2, n, self.F) if self.post else (self.B, n, self.F, self.net.M, 2)
        if tuple(x.shape) != want or not 0 < n <= self.chunk or (self.post and (pre_x is None or pre_x.shape != x.shape)):
            raise ValueError(f"expected {want} with n <= {self.chunk}, got {tuple(x.shape)}")
        if self.pos + n > self.T_max:
            raise ValueError(f"utterance longer than the T_max={self.T_max} given to stream_begin")
        if not x.is_cuda:
            raise _lib.EabError("stream.step needs CUDA (ROCm) tensors; there is no CPU fallback by design.")
        lo, hi, end = self.pos, self.pos + n, self.pos + self.chunk
        with torch.cuda.device(x.device), torch.no_grad():
            for buf, src in ((self._in, x), (self._in2, pre_x)):
                if buf is None:
                    continue
                buf.narrow(tdim, lo, n).copy_(src, non_blocking=True)
                if n < self.chunk and end > hi:       # rows the kernels touch beyond the new frames must be defined
                    buf.narrow(tdim, hi, min(end, self.T_max) - hi).zero_()
            self.bound.t_pos.fill_(lo)
            if self.bound.graph is not None:
                self.bound.graph.replay()
            else:
                self.bound.run(torch.cuda.current_stream().cuda_stream)
            out = self._out[..., lo:hi, :].clone()
        self.pos += n
        self._closed = n < self.chunk
        if self.post:
            out = out.to(x.dtype)
            return [out[j].permute(0, 1, 3, 2) for j in range(self.net.q)]
        if self.net.topo_type == "miso":
            out = out.sum(dim=-1)
        return out.to(x.dtype)


def _replica(module: nn.Module) -> nn.Module:
    """A second handle on the same parameters with its own lowered-program cache (own activation arena, boundary
    buffers and hipGraph): shallow copies down the module tree, ``nn.Parameter`` objects shared."""
    import copy
    rep = copy.copy(module)
    rep._modules = {k: (_replica(v) if v is not None else None) for k, v in module._modules.items()}
    if isinstance(rep, _HipModule):
        rep._bound, rep._packed_version = {}, {}
        rep.__dict__.pop("_slot_list", None)
    return rep


_PIPE_STREAMS: Dict[tuple, list] = {}


def _sync_knobs(src: nn.Module, rep: nn.Module) -> None:
    for k in ("precision", "use_graph", "training"):
        if k in src.__dict__:
            rep.__dict__[k] = src.__dict__[k]
    for name, child in src._modules.items():
        if child is not None:
            _sync_knobs(child, rep._modules[name])


class Pipeline:
    """Throughput executor: keeps ``depth`` batches in flight on ``depth`` HIP streams, each through its own
    replica of the model (own captured program, activations and boundary buffers; shared parameters).  One
    program alone leaves capacity idle -- the LSTM occupies 161 of 256 CUs for a fifth of the step, the S-TCM
    launches 112, every kernel ends in a partial round of workgroups -- and a second, independent batch fills
    it: 9.06 -> 7.48 ms per 16-utterance step in exact fp32, 5.74 -> 4.49 ms in f16x3 (deeper pipelines add
    nothing).  Results come back in submission order and are bit-identical to calling the model directly.
    Works for ``EaBNet``, ``GaGNet`` (two inputs) and ``EaBNetWithPostNet``; inference only.

        pipe = Pipeline(net, depth=2, front_end=(320, 160, torch.hann_window(320)))
        for wav in batches:            # (B, M, L) waves; without front_end: whatever the model takes
            if pipe.outstanding == pipe.depth:
                y = pipe.collect()     # what model(x) returns, ordered on the current stream
            pipe.submit(wav)
        while pipe.outstanding: y = pipe.collect()
    """

    def __init__(self, model: nn.Module, depth: int = 2, front_end: Optional[tuple] = None, prepare=None):
        """front_end = (fft_num, hop, window): ``submit(wav)`` takes (B, M, L) waves -- on the device, or in HOST memory: the
        upload then goes through the pinned staging ring on its own copy stream and overlaps the batches already in flight.
        prepare = the ``args`` of ``prepare_data`` (train_distributed.py:68-95): ``submit(x, target)`` takes what the
        reference's loader yields (host or device), runs ``prepare_data`` on the slot's stream and ``collect()`` returns
        ``(model(noisy_stft), target_stft)`` -- the reference's step with its host-to-device copies hidden behind compute."""
        if depth < 1:
            raise ValueError("depth must be >= 1")
        if front_end is not None and prepare is not None:
            raise ValueError("front_end and prepare are alternatives")
        self.model, self.depth, self.front_end, self.prepare = model, depth, front_end, prepare
        self._replicas = [_replica(model) for _ in range(depth)] if depth > 1 else [model]
        if depth > 1:
            # with several batches in flight the post-filter's three S-TCM chains run back to back: the
            # parallelism comes from the other batch, and hipGraphs with internal branches do not overlap
            # each other (two-stage step 13.99 -> 12.13 ms; with branches kept: no gain at all)
            for rep in self._replicas:
                for m in rep.modules():
                    if isinstance(m, GaGNet):
                        m.parallel_chains = False
        self._streams: list = []
        self._pending: list = []
        self._n = 0

    @property
    def outstanding(self) -> int:
        return len(self._pending)

    @staticmethod
    def _tensors(obj):
        if torch.is_tensor(obj):
            yield obj
        elif isinstance(obj, dict):
            for v in obj.values():
                yield from Pipeline._tensors(v)
        elif isinstance(obj, (list, tuple)):
            for v in obj:
                yield from Pipeline._tensors(v)

    def submit(self, *inputs: torch.Tensor) -> None:
        """Enqueue one batch (the model's positional inputs, or the front end's: see __init__); returns immediately.  At most
        ``depth`` batches may be outstanding.  Host tensors are accepted where a front end takes them (waves): they are staged
        through the pinned ring, so the caller may refill its buffer as soon as submit returns."""
        if self.outstanding >= self.depth:
            raise RuntimeError("collect() a result before submitting more than `depth` batches")
        if not inputs:
            raise ValueError("nothing to submit")
        takes_host = self.front_end is not None or self.prepare is not None
        if not takes_host and not all(x.is_cuda for x in inputs):
            raise _lib.EabError("Pipeline.submit needs CUDA (ROCm) tensors (host waves only with front_end= / prepare=); "
                                "there is no CPU fallback by design.")
        dev_in = next((x.device for x in inputs if x.is_cuda), None)
        device = dev_in if dev_in is not None else next(self.model.parameters()).device
        if device.type != "cuda":
            raise _lib.EabError("Pipeline runs on MI355X only: move the module to 'cuda'. There is no CPU fallback by design.")
        with torch.cuda.device(device):
            if not self._streams:
                # one set of streams per (device, depth) for the whole process: HIP maps streams onto a few
                # hardware queues in creation order, and two streams that land on the same queue do not overlap
                # (observed: a later Pipeline with freshly created streams gained nothing)
                key = (str(device), self.depth)
                if key not in _PIPE_STREAMS:
                    from .graphs import overlapping_streams
                    _PIPE_STREAMS[key] = overlapping_streams(device, self.depth)   # (probed: on distinct hardware queues)
                self._streams = _PIPE_STREAMS[key]
            if takes_host and not all(x.is_cuda for x in inputs):
                _stager(device).run_beside(self._streams)              # the upload stream on a hardware queue of its own
            slot = self._n % self.depth
            self._n += 1
            st = self._streams[slot]
            st.wait_stream(torch.cuda.current_stream())            # device inputs were produced on the caller's stream
            extra = None
            with torch.cuda.stream(st), torch.no_grad():
                for x in inputs:
                    if x.is_cuda:
                        x.record_stream(st)
                if self.prepare is not None:
                    if len(inputs) != 2:
                        raise ValueError("prepare=: submit(x (B,M,L), target (B,1,L))")
                    noisy, extra = prepare_data(inputs[0], inputs[1], device, self.prepare)     # uploads + both STFTs on `st`
                    inputs = (noisy,)
                elif self.front_end is not None:
                    fft_num, hop, window = self.front_end
                    wav, consumed = _upload(inputs[0], device)     # (device tensors pass through)
                    spec = stft_compress(wav, fft_num, hop, window)
                    if consumed is not None:
                        consumed.record(st)                        # the staged wave has no reader after the STFT
                    inputs = (spec,) + tuple(inputs[1:])
                rep = self._replicas[slot]
                if rep is not self.model:
                    _sync_knobs(self.model, rep)                   # precision / use_graph / train-eval follow the model
                out = rep(*inputs)
                ev = torch.cuda.Event()
                ev.record(st)
            self._pending.append((out if extra is None else (out, extra), ev, inputs))

    def calibrate(self, *inputs: torch.Tensor, tries: int = 4, steps: int = 6) -> float:
        """Optional, once per process and depth: whether two streams overlap depends on how the runtime maps them
        onto hardware queues.  Times ``steps`` pipelined batches on up to ``tries`` candidate stream sets and keeps
        the fastest (also for later pipelines of this depth).  Returns its seconds per batch."""
        import time
        if self.depth == 1:
            return 0.0
        device = next((x.device for x in inputs if x.is_cuda), None) or next(self.model.parameters()).device
        key = (str(device), self.depth)
        best, best_streams = None, None
        for attempt in range(tries):
            if attempt > 0 or not self._streams:
                with torch.cuda.device(device):
                    torch.cuda.synchronize(device)
                    self._streams = [torch.cuda.Stream(device=device) for _ in range(self.depth)]
            for _ in range(self.depth + 1):                  # programs lowered / captured, caches warm
                self.submit(*inputs)
                self.collect()
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for _ in range(steps):
                if self.outstanding == self.depth:
                    self.collect()
                self.submit(*inputs)
            while self._pending:
                self.collect()
            torch.cuda.synchronize(device)
            dt = (time.perf_counter() - t0) / steps
            if best is None or dt < best:
                best, best_streams = dt, self._streams
            if attempt == 0:
                first = dt
            elif best < 0.9 * first or attempt >= 1 and abs(dt - first) < 0.03 * first:
                break                                        # clearly better found, or nothing changes: stop
        self._streams = _PIPE_STREAMS[key] = best_streams
        return best

    def collect(self):
        """The oldest outstanding result; the caller's current stream is ordered behind its computation."""
        if not self._pending:
            raise RuntimeError("nothing outstanding")
        out, ev, _ = self._pending.pop(0)
        first = next(self._tensors(out))
        cur = torch.cuda.current_stream(first.device)
        cur.wait_event(ev)
        for t in self._tensors(out):
            t.record_stream(cur)
        return out

    def map(self, batches):
        """Generator over the results of ``batches`` in order, with ``depth`` of them in flight."""
        for x in batches:
            if self.outstanding == self.depth:
                yield self.collect()
            self.submit(*(x if isinstance(x, (tuple, list)) else (x,)))
        while self._pending:
            yield self.collect()


class GaGNet(_HipModule):
    """MI355X implementation of the reference post-filter ``GaGNet`` (GaGNet.py:5-90): same
    constructor keywords, same state-dict keys, ``forward(inpt, pre_x)`` with both (B, 2, T, F)
    returning the list of the q stage estimates, each (B, 2, F, T).  Inference runs the HIP program
    (U2-encoder on the conv kernels, 24 single-branch S-TCMs per stage, fused gain/residual tail);
    a call that needs autograd runs the HIP training programs (eabnet_amd/train_gag.py)."""

    def __init__(self, cin: int = 2, k1: tuple = (2, 3), k2: tuple = (1, 3), c: int = 64, kd1: int = 3, cd1: int = 64,
                 d_feat: int = 256, p: int = 2, q: int = 3, dilas=(1, 2, 5, 9), fft_num: int = 320, is_u2: bool = True,
                 is_causal: bool = True, is_squeezed: bool = False, acti_type: str = "sigmoid",
                 intra_connect: str = "cat", norm_type: str = "IN"):
        super().__init__()
        self.cin, self.k1, self.k2, self.c, self.kd1, self.cd1 = cin, tuple(k1), tuple(k2), c, kd1, cd1
        self.d_feat, self.p, self.q, self.dilas, self.fft_num = d_feat, p, q, list(dilas), fft_num
        self.is_u2, self.is_causal, self.is_squeezed = is_u2, is_causal, is_squeezed
        self.acti_type, self.intra_connect, self.norm_type = acti_type, intra_connect, norm_type
        self.cfg = GagConfig(cin=cin, k1=tuple(k1), k2=tuple(k2), c=c, kd1=kd1, cd1=cd1, d_feat=d_feat, p=p, q=q,
                             dilas=tuple(dilas), fft_num=fft_num, is_u2=is_u2, is_causal=is_causal,
                             is_squeezed=is_squeezed, acti_type=acti_type, intra_connect=intra_connect,
                             norm_type=norm_type)
        self._init_params(gag_param_specs(self.cfg))

    def forward(self, inpt: torch.Tensor, pre_x: torch.Tensor) -> list:
        """:param inpt, pre_x: (B, 2, T, F) noisy reference-microphone spectrum and previous estimate
        :return: list of q estimates (B, 2, F, T)   (reference GaGNet.py:76-90)"""
        if inpt.ndim != 4 or inpt.shape[1] != 2 or inpt.shape[3] != self.cfg.freq or pre_x.shape != inpt.shape:
            raise ValueError(f"expected two (B,2,T,{self.cfg.freq}) tensors, got {tuple(inpt.shape)} and {tuple(pre_x.shape)}")
        if self._needs_graph(inpt, pre_x):
            # training: forward AND backward on the hand-written kernels (train_gag.py).  No gradient flows to the inputs:
            # the reference feeds the detached beam-former estimate (EaBNet.py:142)
            from . import train_gag
            if not (inpt.is_cuda and pre_x.is_cuda and next(self.parameters()).is_cuda):
                raise _lib.EabError("eabnet_amd.GaGNet trains on MI355X only: move the inputs and the module to 'cuda'. "
                                    "There is no CPU fallback by design.")
            if inpt.requires_grad or pre_x.requires_grad:
                raise _refuse_differentiable(self, "an input requires grad", "the reference detaches the beam-former's estimate, "
                                             "EaBNet.py:142; the training programs produce parameter gradients only")
            if self.norm_type == "BN" and not self.training:
                raise _refuse_differentiable(self, "BatchNorm in eval mode under autograd", "the training programs implement "
                                             "BatchNorm's train mode; call .train(), or torch.no_grad() for inference")
            if not train_gag.supported(self.cfg):
                raise _refuse_differentiable(self, "post-filter topology outside the training programs", "see train_gag.supported")
            self.training_backend = "hip"
            return train_gag.forward_train(self, inpt, pre_x)
        if not (inpt.is_cuda and pre_x.is_cuda):
            raise _lib.EabError("eabnet_amd.GaGNet inference runs on MI355X only: move the inputs (and module) to "
                                "'cuda'. There is no CPU fallback by design.")
        _lib.load()
        B, _, T, F = inpt.shape
        a = inpt.detach().to(torch.float32).contiguous()
        b = pre_x.detach().to(torch.float32).contiguous()
        with torch.cuda.device(a.device):
            bound = self._program(B, T, F, a.device)
            if self.use_graph and not torch.cuda.is_current_stream_capturing() \
                    and bound.capture((B, 2, T, F), (self.q, B, 2, T, F), (B, 2, T, F)):
                bound.static_in.copy_(a, non_blocking=True)
                bound.static_in2.copy_(b, non_blocking=True)
                bound.graph.replay()
                out = bound.static_out.clone()
            else:
                out = torch.empty((self.q, B, 2, T, F), dtype=torch.float32, device=a.device)
                bound.bind(a.data_ptr(), out.data_ptr(), b.data_ptr())
                bound.run(torch.cuda.current_stream().cuda_stream)
        self._last = (bound, a, b)
        out = out.to(inpt.dtype)
        return [out[j].permute(0, 1, 3, 2) for j in range(self.q)]


def make_gag_net(args) -> GaGNet:
    """Reference GaGNet.py:651-671 (reads the ``gagnet_*`` fields of ``args``), on cuda:current."""
    return GaGNet(cin=2, k1=args.gagnet_k1, k2=args.gagnet_k2, c=args.gagnet_c, kd1=args.gagnet_kd1, cd1=args.gagnet_cd1,
                  d_feat=args.gagnet_d_feat, p=args.gagnet_p, q=args.gagnet_q, dilas=args.gagnet_dilas,
                  fft_num=args.gagnet_fft_num, is_u2=args.gagnet_is_u2, is_causal=args.gagnet_is_causal,
                  is_squeezed=args.gagnet_is_squeezed, acti_type=args.gagnet_acti_type,
                  intra_connect=args.gagnet_intra_connect, norm_type=args.gagnet_norm_type).cuda()


class EaBNetWithPostNet(nn.Module):
    """Reference EaBNet.py:127-155: beam-former, then the GaGNet post-filter on (reference microphone,
    detached beam-former estimate).  Same ``args`` fields, ``eabnet.`` / ``postnet.`` key prefixes and
    output dictionary."""

    def __init__(self, args):
        super().__init__()
        self.eabnet = EaBNet(k1=args.k1, k2=args.k2, c=args.c, M=args.M, embed_dim=args.embed_dim, kd1=args.kd1,
                             cd1=args.cd1, d_feat=args.d_feat, p=args.p, q=args.q, is_causal=args.is_causal,
                             is_u2=args.is_u2, bf_type=args.bf_type, topo_type=args.topo_type,
                             intra_connect=args.intra_connect, norm_type=args.norm_type)
        self.ref_mic = args.ref_mic
        self.postnet = make_gag_net(args)
        if args.freeze_eabnet:
            self.freeze_eabnet()

    def forward(self, noisy_stft: torch.Tensor) -> dict:
        esti0_stft = self.eabnet(noisy_stft)
        inpt = noisy_stft[..., self.ref_mic, :].permute(0, 3, 1, 2)            # 'b t f c -> b c t f'
        esti1_stft_list = self.postnet(inpt, esti0_stft.detach())
        return {"esti0_stft": esti0_stft, "esti1_stft_list": esti1_stft_list,
                "esti_stft": esti1_stft_list[-1].permute(0, 1, 3, 2)}

    def freeze_eabnet(self) -> None:
        for p in self.eabnet.parameters():
            p.requires_grad = False

    def stream_begin(self, B: int, T_max: int, chunk: int = 1) -> "TwoStageStream":
        """Frame-synchronous two-stage inference: both stages need BatchNorm norms and causal S-TCMs."""
        return TwoStageStream(self, self.eabnet.stream_begin(B, T_max, chunk), self.postnet.stream_begin(B, T_max, chunk))


class TwoStageStream:
    """EaBNetWithPostNet.forward (EaBNet.py:138-148) chunk by chunk: beam-former step, then post-filter
    step on (reference microphone, estimate)."""

    def __init__(self, net: EaBNetWithPostNet, first: EaBNetStream, second: EaBNetStream):
        self.net, self.first, self.second = net, first, second

    def reset(self) -> None:
        self.first.reset()
        self.second.reset()

    def step(self, noisy: torch.Tensor) -> dict:
        esti0 = self.first.step(noisy)
        inpt = noisy[..., self.net.ref_mic, :].permute(0, 3, 1, 2)
        lst = self.second.step(inpt, esti0)
        return {"esti0_stft": esti0, "esti1_stft_list": lst, "esti_stft": lst[-1].permute(0, 1, 3, 2)}


class StreamingEnhancer:
    """enhance.py:45-62 as a real-time loop on the device: ``push`` takes the next ``chunk`` hops of
    multichannel samples (B, M, chunk*hop) and returns the enhanced samples that became final.

    Frame t of the centred STFT needs samples up to (t+1)*hop and output segment k of the ISTFT needs frames
    k and k+1, so the enhanced wave trails the input by two hops (20 ms at 16 kHz) plus the compute time of a
    step.  Front and back end are the offline kernels on short windows (a frame / segment is computed
    identically wherever its window starts), so the streamed wave is bit-identical to the offline
    wave -> prepare_data -> model -> istft chain."""

    def __init__(self, model, B: int, seconds: float, chunk: int = 1, sr: int = 16000, fft_num: int = 320, hop: int = 160):
        self.model, self.B, self.chunk, self.fft, self.hop = model, B, chunk, fft_num, hop
        self.T_max = 1 + int(seconds * sr) // hop
        self.stream = model.stream_begin(B, self.T_max, chunk)
        self.window = torch.hann_window(fft_num)
        self.reset()

    def reset(self) -> None:
        self.stream.reset()
        self._tail = None          # the last hop samples (B, M, hop): left half of the next frame
        self._hold = None          # samples of a first push too short to form frame 0
        self._frames = 0           # frames handed to the model so far
        self._spec_prev = None     # last estimate frame, for the overlap-add with the next one

    def push(self, samples: torch.Tensor, last: bool = False) -> torch.Tensor:
        """samples (B, M, n*hop), n == chunk (any 0 <= n <= chunk with last=True).  Returns (B, k*hop)
        enhanced samples, k = frames that became final in this call."""
        B, M, L = samples.shape
        if L % self.hop or L // self.hop > self.chunk or (L // self.hop != self.chunk and not last):
            raise ValueError(f"push takes chunk*hop = {self.chunk * self.hop} samples per call (fewer only with last=True)")
        first = self._tail is None
        if first and self._hold is not None:
            samples, self._hold = torch.cat((self._hold, samples), dim=2), None
        if first and samples.shape[2] <= self.fft // 2 and not last:
            self._hold = samples                      # frame 0 reflects about sample 0 and needs sample fft_num/2
            return samples.new_zeros((B, 0))
        buf = samples if first else torch.cat((self._tail, samples), dim=2)
        # window of `buf`: as an utterance of its own, its interior frames are exact; its first frame is exact only
        # at the true start (reflection), its last one only at the true end
        if buf.shape[2] <= self.fft // 2:
            raise ValueError("an utterance must be longer than fft_num/2 samples")
        spec = stft_compress(buf, self.fft, self.hop, self.window)                # (B, T', F, M, 2)
        lo = 0 if first else 1
        hi = spec.shape[1] if last else spec.shape[1] - 1
        new = spec[:, lo:hi]
        self._tail = buf[:, :, buf.shape[2] - self.hop:]       # the next window starts one hop before its first new frame
        outs = []
        for a in range(0, new.shape[1], self.chunk):
            blk = new[:, a:a + self.chunk]
            o = self.stream.step(blk)
            outs.append(o["esti_stft"] if isinstance(o, dict) else o)
        if not outs:
            return samples.new_zeros((B, 0))
        est = torch.cat(outs, dim=2)                                                # (B, 2, k, F)
        self._frames += est.shape[2]
        seq = est if self._spec_prev is None else torch.cat((self._spec_prev, est), dim=2)
        self._spec_prev = est[:, :, -1:].clone()
        if seq.shape[2] < 2:
            return samples.new_zeros((B, 0))
        return istft(seq, self.fft, self.hop, self.window)


def make_eabnet_with_postnet(args) -> EaBNetWithPostNet:
    """Reference EaBNet.py:815-816."""
    return EaBNetWithPostNet(args)


# ----------------------------------------------------------------------------
# front end and helpers
# ----------------------------------------------------------------------------
_TWIDDLE: Dict[Tuple[int, str], torch.Tensor] = {}
_WINDOW: Dict[tuple, torch.Tensor] = {}


def _device_window(window: torch.Tensor, device: torch.device) -> torch.Tensor:
    """The analysis/synthesis window on the device.  prepare_data builds it on the CPU on every call, as the
    reference does (train_distributed.py:83), and a pageable host-to-device copy blocks the host until the
    stream has drained -- one full pipeline bubble per step.  CPU windows are therefore cached by content."""
    if window.is_cuda:
        return window.to(device=device, dtype=torch.float32).contiguous()
    w = window.detach().to(torch.float32).contiguous()
    key = (str(device), w.numpy().tobytes())
    hit = _WINDOW.get(key)
    if hit is None:
        if len(_WINDOW) > 16:
            _WINDOW.clear()
        hit = _WINDOW[key] = w.to(device)
    return hit


def _twiddle(n_fft: int, device: torch.device) -> torch.Tensor:
    key = (n_fft, str(device))
    if key not in _TWIDDLE:
        k = np.arange(n_fft, dtype=np.float64)
        tw = np.stack([np.cos(2 * np.pi * k / n_fft), np.sin(2 * np.pi * k / n_fft)], axis=1).astype(np.float32)
        _TWIDDLE[key] = torch.from_numpy(tw).to(device)
    return _TWIDDLE[key]


def stft_compress(wav: torch.Tensor, n_fft: int, hop: int, window: torch.Tensor, layout: int = 0) -> torch.Tensor:
    """(B, M, L) -> (B, T, F, M, 2)  [layout 0]  or (B, 1, L) -> (B, 2, T, F)  [layout 1]."""
    if not wav.is_cuda:
        raise _lib.EabError("stft_compress needs a CUDA (ROCm) tensor; there is no CPU fallback by design.")
    lib = _lib.load()
    B, M, L = wav.shape
    T, F = 1 + L // hop, n_fft // 2 + 1
    wav = wav.to(torch.float32).contiguous()
    window = _device_window(window, wav.device)
    out = torch.empty((B, T, F, M, 2) if layout == 0 else (B, 2, T, F), dtype=torch.float32, device=wav.device)
    with torch.cuda.device(wav.device):
        tw = _twiddle(n_fft, wav.device)
        _lib.check(lib.eab_stft_compress_f32(wav.data_ptr(), window.data_ptr(), tw.data_ptr(), out.data_ptr(),
                                             B, M, L, n_fft, hop, layout,
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                   "eab_stft_compress_f32")
    return out


_COPY_POOL = None
_COPY_THREADS = 4


def _host_copy(dst: torch.Tensor, src: torch.Tensor) -> None:
    """dst.copy_(src) on the host; a large contiguous fp32 tensor is copied as _COPY_THREADS slices on worker threads (the copy
    is one memcpy per call and releases the GIL): the 33 MB of a batch of waves are 3.3 ms of ONE core otherwise, and on a busy host
    that single memcpy is what a pipelined step waits for."""
    global _COPY_POOL
    n = src.numel()
    if n < (1 << 21) or not (src.is_contiguous() and dst.is_contiguous() and src.dtype == dst.dtype):
        dst.copy_(src)
        return
    if _COPY_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _COPY_POOL = ThreadPoolExecutor(max_workers=_COPY_THREADS, thread_name_prefix="eab-stage")
    d, s_ = dst.view(-1), src.view(-1)
    step = (n + _COPY_THREADS - 1) // _COPY_THREADS
    futs = [_COPY_POOL.submit(d[a:a + step].copy_, s_[a:a + step]) for a in range(0, n, step)]
    for f in futs:
        f.result()


class _HostStager:
    """Host -> device path of ``prepare_data`` (train_distributed.py:76-77 does two blocking ``.to(device)``): a ring of
    pinned staging buffers and resident device buffers plus a dedicated copy stream.  The upload of batch k+1 is
    enqueued while batch k still computes and nothing on the path allocates or synchronises the device; a slot is
    reused only after the kernels that read its device buffer have been enqueued AND its previous copy completed."""

    # True: every host tensor is copied into the ring's own pinned slot first.  The direct path (the copy kernel reads the
    # caller's pinned buffer) measured SLOWER on MI355X (20 ms vs 10-12 ms per 32.8 MB batch, bench.py next_rows.pcie_inclusive)
    # and has to block until the read has finished, because nothing else protects a caller's pinned buffer from being refilled.
    always_stage = True

    def __init__(self, device: torch.device, depth: int = 4):
        # (one slot more than the deepest Pipeline keeps in flight: a slot is refilled only when its batch has been consumed)
        self.device, self.depth, self.k = device, depth, {}
        self.slots: Dict[tuple, list] = {}
        # high priority: the copy kernel is a few dozen workgroups that must get their CU slots while other batches compute --
        # on a normal-priority stream its workgroups queue behind the convolutions' and every pipeline slot waits for its upload
        self.stream = torch.cuda.Stream(device=device, priority=-1)
        self._beside: tuple = ()

    def run_beside(self, streams) -> None:
        """The copy stream must not share a hardware queue with a stream whose batches it feeds: a queue is in order, so an
        upload behind a whole program of another batch arrives a step late (measured: 7-9 ms per step became 17-22 whenever the
        creation-order lottery put the two together).  Picks -- once per set of streams -- a high-priority stream that a spin-kernel
        probe shows overlapping with all of them (graphs.overlapping_streams)."""
        key = tuple(s.cuda_stream for s in streams)
        if key == self._beside:
            return
        from .graphs import overlapping_streams
        torch.cuda.synchronize(self.device)
        self.stream = overlapping_streams(self.device, 1, beside=list(streams), priority=-1)[0]
        self._beside = key

    def upload(self, t: torch.Tensor, role: int = 0) -> Tuple[torch.Tensor, "torch.cuda.Event"]:
        """CPU tensor -> contiguous fp32 device tensor, ordered on the CURRENT stream.  Returns (tensor, consumed):
        record `consumed` on the current stream after the last kernel that reads the tensor has been enqueued.
        role: which input of the step this is (x = 0, target = 1) -- inputs of equal shape (one microphone) keep their own rings."""
        key = (role,) + tuple(t.shape)
        ring = self.slots.get(key)
        if ring is None:
            if len(self.slots) > 8:
                torch.cuda.synchronize(self.device)
                self.slots.clear()
                self.k.clear()
            ring = self.slots[key] = [dict(pin=torch.empty(key[1:], dtype=torch.float32).pin_memory(),
                                           dev=torch.empty(key[1:], dtype=torch.float32, device=self.device),
                                           copied=torch.cuda.Event(), consumed=torch.cuda.Event(), used=False)
                                      for _ in range(self.depth)]
        # (one counter per ring: x and target of a step use different rings -- a shared counter walked each of them with a
        # stride of two, i.e. gave a pipeline of three batches TWO slots per ring and made submit() wait for the GPU)
        self.k[key] = self.k.get(key, 0) + 1
        slot = ring[self.k[key] % self.depth]
        if slot["used"]:
            slot["copied"].synchronize()                 # the pinned buffer is free again (depth steps ago)
            self.stream.wait_event(slot["consumed"])     # the device buffer's readers were enqueued before this
        src = t
        if self.always_stage or not (t.is_pinned() and t.is_contiguous() and t.dtype == torch.float32):
            _host_copy(slot["pin"], t)                   # pageable / strided / other dtype: one host pass into pinned memory
            src = slot["pin"]
        with torch.cuda.stream(self.stream):
            # a copy KERNEL reading the pinned buffer over the bus (device-visible under unified addressing), not
            # hipMemcpyAsync: DMA copies submitted between compute kernels cost milliseconds each on this stack
            _lib.check(_lib.load().eab_copy_f32(src.data_ptr(), slot["dev"].data_ptr(), src.numel(), C.c_void_p(self.stream.cuda_stream)),
                       "eab_copy_f32")
            slot["copied"].record(self.stream)
        torch.cuda.current_stream(self.device).wait_event(slot["copied"])
        slot["used"] = True
        if src is t:
            # the copy kernel reads the CALLER's pinned buffer: nothing ties that buffer's contents or lifetime to our side
            # stream (a reused pinned batch, a DataLoader(pin_memory=True) block recycled for the next batch), so return only
            # when the read has finished -- what the reference's blocking x.to(device) gives (train_distributed.py:76-77)
            slot["copied"].synchronize()
        return slot["dev"], slot["consumed"]


_STAGERS: Dict[str, _HostStager] = {}


def _stager(device: torch.device) -> _HostStager:
    st = _STAGERS.get(str(device))
    if st is None:
        st = _STAGERS[str(device)] = _HostStager(device)
    return st


def _upload(t: torch.Tensor, device: torch.device, role: int = 0):
    """(device tensor, event to record after its consumers) for any input of prepare_data"""
    if t.is_cuda:
        return t.to(device), None
    return _stager(device).upload(t, role)


def prepare_data(x: torch.Tensor, target: torch.Tensor, device, args):
    """Reference train_distributed.py:68-95.  ``args`` provides mics, sr,
    wav_len, win_size, win_shift (seconds) and fft_num.
    x (B, M, L), target (B, 1, L) -> noisy_stft (B, T, F, M, 2), target_stft (B, 2, T, F)."""
    sr = args.sr
    win_size = int(args.win_size * sr)
    win_shift = int(args.win_shift * sr)
    fft_num = args.fft_num
    if win_size > fft_num:
        raise RuntimeError(f"win_size ({win_size} samples) must not exceed fft_num ({fft_num}), as in torch.stft")
    batch_size = x.shape[0]
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    with torch.cuda.device(device):
        xd, ev_x = _upload(x, device)
        td, ev_t = _upload(target, device, 1)
        noisy_wav = xd.contiguous().view(batch_size, args.mics, -1)
        target_wav = td.reshape(batch_size, 1, -1)
        window = torch.hann_window(win_size)
        if win_size < fft_num:                        # torch.stft(win_length < n_fft): zero-padded on both sides, centred
            left = (fft_num - win_size) // 2
            window = torch.nn.functional.pad(window, (left, fft_num - win_size - left))
        noisy_stft = stft_compress(noisy_wav, fft_num, win_shift, window, 0)
        target_stft = stft_compress(target_wav, fft_num, win_shift, window, 1)
        for ev in (ev_x, ev_t):
            if ev is not None:
                ev.record(torch.cuda.current_stream(device))         # the staged waves have no reader after this point
    return noisy_stft, target_stft


_NOLA: Dict[tuple, bool] = {}


def _check_nola(window: torch.Tensor, fft_num: int, hop: int, T: int) -> None:
    """torch.istft's window-overlap condition, on the host, before the kernel divides by the envelope: the squared-window
    overlap-add over the frames that exist must stay above 1e-11 at every output sample (a short zero-padded window with a
    large hop leaves gaps).  The envelope's two rims and its interior (periodic in the hop, whether or not the hop divides
    fft_num) are those of a signal of 2R+2 frames, R = ceil(fft_num / hop)."""
    w = window.detach().to("cpu", torch.float64).numpy()                     # (already zero-padded to fft_num by the caller)
    key = (w.tobytes(), fft_num, hop, min(T, 2 * -(-fft_num // hop) + 2))   # by content: a pointer can be reused by another window
    ok = _NOLA.get(key)
    if ok is None:
        w2 = w ** 2
        Te = key[-1]
        env = np.zeros(fft_num + hop * (Te - 1))
        for t in range(Te):
            env[t * hop:t * hop + fft_num] += w2
        trimmed = env[fft_num // 2:fft_num // 2 + hop * (Te - 1)]
        ok = bool(trimmed.size == 0 or np.abs(trimmed).min() > 1e-11)
        if len(_NOLA) > 64:
            _NOLA.clear()
        _NOLA[key] = ok
    if not ok:
        raise RuntimeError("istft: window overlap add min is below 1e-11 (the NOLA condition of torch.istft fails for this "
                           f"window / hop: fft_num={fft_num}, win_shift={hop})")


def istft(esti_stft: torch.Tensor, fft_num: int, win_shift: int, window: torch.Tensor) -> torch.Tensor:
    """The reference's back end (enhance.py:59-62, test.py:189-191, train_distributed.py:128-130)
    ``torch.istft(view_as_complex(esti.permute(0,3,2,1)), fft_num, win_shift, win_size, hann)`` in one
    HIP kernel: (B, 2, T, F) -> (B, win_shift*(T-1)).  Like the reference it inverts the estimate as
    it is (compressed domain)."""
    if not esti_stft.is_cuda:
        raise _lib.EabError("istft needs a CUDA (ROCm) tensor; there is no CPU fallback by design.")
    if esti_stft.ndim != 4 or esti_stft.shape[1] != 2 or esti_stft.shape[3] != fft_num // 2 + 1:
        raise ValueError(f"expected (B,2,T,{fft_num // 2 + 1}), got {tuple(esti_stft.shape)}")
    if window.numel() > fft_num:
        raise RuntimeError(f"window ({window.numel()} samples) must not exceed fft_num ({fft_num}), as in torch.istft")
    if not 0 < win_shift <= window.numel():
        raise RuntimeError(f"istft: expected 0 < win_shift <= win_length, got {win_shift} and {window.numel()} (as torch.istft)")
    if window.numel() < fft_num:                  # torch.istft(win_length < n_fft): zero-padded on both sides, centred
        left = (fft_num - window.numel()) // 2
        window = torch.nn.functional.pad(window, (left, fft_num - window.numel() - left))
    if -(-fft_num // win_shift) > 8:
        raise NotImplementedError("the HIP back end implements any hop with at most 8 overlapping frames "
                                  "(ceil(fft_num / win_shift) in 1..8; the reference's 320/160 is 2)")
    lib = _lib.load()
    B, _, T, _ = esti_stft.shape
    _check_nola(window, fft_num, win_shift, T)
    x = esti_stft.detach().to(torch.float32).contiguous()
    window = _device_window(window, x.device)
    wav = torch.empty((B, win_shift * (T - 1)), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.eab_istft_f32(x.data_ptr(), window.data_ptr(), _twiddle(fft_num, x.device).data_ptr(),
                                     wav.data_ptr(), B, T, fft_num, win_shift,
                                     C.c_void_p(torch.cuda.current_stream().cuda_stream)), "eab_istft_f32")
    return wav


def filter_and_sum(w: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """Stand-alone K13 (reference EaBNet.py:114-117): (B,T,F,M,2)^2 -> (B,2,T,F)."""
    if not (w.is_cuda and x.is_cuda):
        raise _lib.EabError("filter_and_sum needs CUDA (ROCm) tensors; there is no CPU fallback by design.")
    lib = _lib.load()
    B, T, F, M, _ = x.shape
    w = w.to(torch.float32).contiguous()
    x = x.to(torch.float32).contiguous()
    y = torch.empty((B, 2, T, F), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.eab_filter_sum_f32(w.data_ptr(), x.data_ptr(), y.data_ptr(), B, T, F, M,
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream)), "eab_filter_sum_f32")
    return y


def numParams(net: nn.Module) -> int:
    """Reference EaBNet.py:653-659."""
    return sum(int(np.prod(p.size())) for p in net.parameters() if p.requires_grad)


class _ComMagMseLoss(torch.autograd.Function):
    """Loss value and d loss / d esti from one fused HIP pass (csrc/loss.hip); the backward is a scale."""

    @staticmethod
    def forward(ctx, esti: torch.Tensor, label: torch.Tensor, frames: tuple) -> torch.Tensor:
        lib = _lib.load()
        B, _, T, F = esti.shape
        e = esti.detach().to(torch.float32).contiguous()
        lab = label.detach().to(torch.float32).contiguous()
        grad = torch.empty_like(e) if ctx.needs_input_grad[0] else None
        nblocks = 1024
        partial = torch.empty(2 * nblocks, dtype=torch.float32, device=e.device)
        loss = torch.empty((), dtype=torch.float32, device=e.device)
        fr = (C.c_int32 * B)(*[int(n) for n in frames])
        with torch.cuda.device(e.device):
            _lib.check(lib.eab_com_mag_mse_loss_f32(e.data_ptr(), lab.data_ptr(), fr, B, T, F, partial.data_ptr(), nblocks,
                                                    loss.data_ptr(), grad.data_ptr() if grad is not None else None,
                                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                       "eab_com_mag_mse_loss_f32")
        ctx.save_for_backward(grad)
        ctx.in_dtype = esti.dtype
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (g * grad).to(ctx.in_dtype), None, None


def _loss_on_device(esti: torch.Tensor, label: torch.Tensor, frame_list) -> bool:
    return (esti.is_cuda and label.is_cuda and esti.ndim == 4 and esti.shape == label.shape and esti.shape[1] == 2
            and esti.shape[0] <= 64 and len(frame_list) == esti.shape[0] and not label.requires_grad
            and all(0 <= int(n) <= esti.shape[2] for n in frame_list) and sum(int(n) for n in frame_list) > 0)


def com_mag_mse_loss(esti: torch.Tensor, label: torch.Tensor, frame_list) -> torch.Tensor:
    """Reference EaBNet.py:627-640: 0.5*(masked magnitude MSE + masked complex MSE), esti/label (B,2,T,F).
    CUDA tensors: one fused HIP pass produces the value and the gradient w.r.t. ``esti``
    (eab_com_mag_mse_loss_f32); anything else (CPU tensors, > 64 utterances, a label that needs a gradient)
    is evaluated with the reference's tensor expressions."""
    if _loss_on_device(esti, label, frame_list):
        return _ComMagMseLoss.apply(esti, label, tuple(int(n) for n in frame_list))
    B, _, T, F = esti.shape
    mask = torch.zeros((B, T, F), dtype=esti.dtype, device=esti.device)
    for i, n in enumerate(frame_list):
        mask[i, :n] = 1.0
    com_mask = torch.stack((mask, mask), dim=1)
    mag_e, mag_l = torch.norm(esti, dim=1), torch.norm(label, dim=1)
    loss1 = (((mag_e - mag_l) ** 2.0) * mask).sum() / mask.sum()
    loss2 = (((esti - label) ** 2.0) * com_mask).sum() / com_mask.sum()
    return 0.5 * (loss1 + loss2)


def stagewise_com_mag_mse_loss(esti_list, label: torch.Tensor, frame_list) -> torch.Tensor:
    """Reference GaGNet.py:601-619: the complex + magnitude MSE of every stage, weight 0.1 (1 for the
    last stage).  esti (B,2,F,T) each, label (B,2,F,T)."""
    lab_t = label.permute(0, 1, 3, 2)
    if all(_loss_on_device(e.permute(0, 1, 3, 2), lab_t, frame_list) for e in esti_list):
        # every stage is the same masked loss on (B,2,T,F) views; the stage outputs of eabnet_amd.GaGNet are
        # already laid out that way, so nothing is copied
        total = 0.0
        for i, e in enumerate(esti_list):
            alpha = 1.0 if i == len(esti_list) - 1 else 0.1
            total = total + alpha * com_mag_mse_loss(e.permute(0, 1, 3, 2), lab_t, frame_list)
        return total
    B, _, Fq, T = label.shape
    mask = torch.zeros((B, Fq, T), dtype=label.dtype, device=label.device)
    for i, n in enumerate(frame_list):
        mask[i, :, :n] = 1.0
    com_mask = torch.stack((mask, mask), dim=1)
    loss1 = loss2 = 0.0
    mag_label = torch.norm(label, dim=1)
    for i, e in enumerate(esti_list):
        alpha = 1.0 if i == len(esti_list) - 1 else 0.1
        loss1 = loss1 + alpha * (((e - label) ** 2.0) * com_mask).sum() / com_mask.sum()
        loss2 = loss2 + alpha * (((torch.norm(e, dim=1) - mag_label) ** 2.0) * mask).sum() / mask.sum()
    return 0.5 * (loss1 + loss2)


def eabnet_with_postnet_loss(output: dict, label: torch.Tensor, frame_list) -> dict:
    """Reference EaBNet.py:642-650 (called at train_distributed.py:225): the beam-former's loss on
    ``esti0_stft`` plus the stage-wise loss of the post-filter estimates; label (B,2,T,F)."""
    loss0 = com_mag_mse_loss(output["esti0_stft"], label, frame_list)
    loss1 = stagewise_com_mag_mse_loss(output["esti1_stft_list"], label.permute(0, 1, 3, 2), frame_list)
    return {"eabnet": loss0, "postnet": loss1, "final": loss0 + loss1}
