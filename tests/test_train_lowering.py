"""Host-side invariants of the training lowering (eabnet_amd/train.py); no GPU.

The backward program emits every weight gradient at its END (sorted by geometry so that eab_run_program can serve runs of
identical geometry with one launch).  That is only correct if a weight gradient's operands are final when their producer
has run and are never written again: checked here on the op lists themselves, for the benchmark shape and a small one."""
from dataclasses import replace

import numpy as np
import pytest

from eabnet_amd import program as prg
from eabnet_amd import spec, train


def _writes(op):
    """Refs an op writes (arena, offset)"""
    k = op.kind
    if k == prg.OP_CONV:
        out = [op.dst, op.dst_acc, op.stats, getattr(op, "glu_dump", None)]
    elif k == prg.OP_IN_FINALIZE:
        out = [op.xf0, op.xf1, getattr(op, "mr0", None), getattr(op, "mr1", None)]
    elif k == prg.OP_MEMSET0:
        out = [op.ptr]
    elif k == train.OP_WGRAD:
        out = [op.dw, op.dbias]
    else:
        p = list(op.p) + [None] * 12
        out = {train.OP_GATHER: [p[3]], train.OP_IN_STATS: [p[4], p[5], p[6]], train.OP_TR_NORM_ACT: [p[4]],
               train.OP_NORM_BWD: [p[6], p[8], p[9], p[10], p[11]], train.OP_GLU_BWD: [p[2]], train.OP_GATE_FWD: [p[2]],
               train.OP_GATE_BWD: [p[3], p[4]], train.OP_ADD: [p[2]], train.OP_RELU_BWD: [p[2]], train.OP_COLSUM: [p[1]],
               train.OP_FILTER_SUM: [p[2]], train.OP_FS_BWD: [p[2]], train.OP_LN_FWD: [p[3], p[4]],
               train.OP_LN_BWD: [p[4], p[5], p[6]], train.OP_LSTM_TRAIN: [p[3], p[4]], train.OP_LSTM_BWD: [p[3]]}[k]
    res = [(r.arena, r.off) for r in out if r is not None]
    if k == train.OP_IN_STATS and len(op.i) > 3 and op.i[3] and op.p[6] is not None:
        B, P, C, xC = op.i[:4]                               # several views, one contiguous tensor each behind p[6]
        res += [(op.p[6].arena, op.p[6].off + v * B * P * xC) for v in range(1, C // xC)]
    return res


@pytest.mark.parametrize("M,B,T,pq", [(8, 6, 601, (6, 3)), (9, 2, 24, (2, 2))])
def test_deferred_weight_gradients_read_final_buffers(M, B, T, pq):
    cfg = replace(spec.NetConfig(), M=M, p=pq[0], q=pq[1])
    prog = train.lower_train(cfg, B, T)
    bwd = prog.bwd
    first_w = next(k for k, o in enumerate(bwd) if o.kind == train.OP_WGRAD)
    assert all(o.kind == train.OP_WGRAD for o in bwd[first_w:]), "weight gradients are the tail of the backward program"
    assert all(o.kind != train.OP_WGRAD for o in bwd[:first_w])
    fwd_written = {w for o in prog.fwd for w in _writes(o)}
    # last writer of every buffer the backward program writes (activation arena; accumulate-in-place counts as a write)
    writes = {}
    for k, o in enumerate(bwd[:first_w]):
        for w in _writes(o):
            writes.setdefault(w, []).append(k)
    for o in bwd[first_w:]:
        dz = (o.dz.arena, o.dz.off)
        assert dz in writes or dz[0] == "dout", f"{o.name}: its output gradient is produced by nobody"
        # the op that the lowering emitted right behind the gradient's producer would have read dz after writes[dz][-1] only
        # if nobody writes it again: every write of dz must come from ONE producer chain (first write, then in-place
        # accumulations by later contributions), all of them before any reader -- i.e. no write after the first reader.
        readers = [k for k, q in enumerate(bwd[:first_w])
                   if q.kind == prg.OP_CONV and (q.src0.arena, q.src0.off) == dz]          # its dgrad launches read dz too
        if readers and dz in writes:
            assert max(writes[dz]) < min(readers), f"{o.name}: dz is written after its dgrad has read it"
        elif dz in writes:                                  # no input gradient wanted (first convolution): one producer, one write
            assert len(writes[dz]) == 1, f"{o.name}: dz has several writers and no reader to order them against"
        for src in (o.src0, o.src1):
            if src is None:
                continue
            s = (src.arena, src.off)
            assert s not in writes, f"{o.name}: forward activation {s} is written by the backward program"
            assert s in fwd_written or s[0] == "in", f"{o.name}: operand {s} is written by nobody"


def test_weight_gradients_batch_into_few_launches():
    """identical geometries are adjacent (what eab_wgrad_batchable looks for): 153 descriptors -> 38 launches at the
    benchmark shape"""
    prog = train.lower_train(replace(spec.NetConfig(), M=8), 6, 601)

    def geo(o):
        return (o.N, o.C0, o.C1, o.Kpad, o.Fin, o.Fz, o.No, o.ostride, o.ophase, o.istride, tuple(o.dt), tuple(o.ioff),
                o.src1 is None, o.dbias is None, o.precision)
    w = [o for o in prog.bwd if o.kind == train.OP_WGRAD]
    runs, prev, run = 0, None, 0
    seen = set()
    for o in w:
        g = geo(o)
        if g == prev and run < 24:
            run += 1
        else:
            assert g not in seen or run == 24, "a geometry appears in two separate runs"
            runs, run = runs + 1, 1
        seen.add(g)
        prev = g
    assert len(w) == 153 and runs == 38
    # every parameter element receives exactly one gradient entry (the inverse table is total on the trained parameters)
    assert (np.asarray(prog.inv) >= 0).all()


def _variants(fname):
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", fname)) as f:
        return json.load(f)


@pytest.mark.parametrize("name", sorted(_variants("keys_variants.json")))
def test_every_constructor_variant_lowers_to_training_programs(name):
    """train.supported() is true for every constructor branch of tests/golden/keys_variants.json, and the lowering
    of each holds the same invariants as the default topology: every trained parameter element gets exactly one gradient
    entry, BatchNorm variants record one (mean, rstd) table per norm for the running-buffer update."""
    e = _variants("keys_variants.json")[name]
    cfg = spec.NetConfig(M=e["M"], **dict(e["kwargs"], p=2, q=2))
    assert train.supported(cfg) and train.unsupported_reason(cfg) == ""
    prog = train.lower_train(cfg, 2, 20)
    assert (np.asarray(prog.inv) >= 0).all()
    specs = spec.param_specs(cfg)
    assert set(prog.keys) == {k for k, s in specs.items() if not s.kind.startswith("bn_")}
    n_bn = sum(1 for s in specs.values() if s.kind == "bn_mean")
    assert len(prog.bn_layers) == n_bn and len({k for k, *_ in prog.bn_layers}) == n_bn
    if n_bn:
        # one virtual utterance: every norm-side op of the programs runs with B = 1
        for o in list(prog.fwd) + list(prog.bwd):
            if o.kind in (train.OP_IN_STATS, train.OP_TR_NORM_ACT, train.OP_NORM_BWD):
                assert o.i[0] == 1
            if o.kind == prg.OP_IN_FINALIZE:
                assert o.B == 1


@pytest.mark.parametrize("name", sorted(_variants("keys_gagnet.json")))
def test_every_post_filter_variant_lowers_to_training_programs(name):
    import eabnet_amd
    from eabnet_amd import train_gag
    kw = dict(_variants("keys_gagnet.json")[name]["kwargs"])
    if name == "default":
        kw.update(p=1, q=2, dilas=[1, 2])
    cfg = eabnet_amd.GaGNet(**kw).cfg
    assert train_gag.supported(cfg)
    prog = train_gag.lower_train(cfg, 2, 14, 161, "f32")
    assert (np.asarray(prog.inv) >= 0).all()
    specs = eabnet_amd.gag_param_specs(cfg)
    assert len(prog.bn_layers) == sum(1 for s in specs.values() if s.kind == "bn_mean")


def test_bf16_storage_is_assigned_only_where_nothing_but_bf16_contractions_read(monkeypatch):
    """bf16 training programs store as bf16 exactly the tensors whose every reader is a bf16 contraction (the gather of a bf16
    convolution, the bf16 weight gradient): each such tensor has ONE writer, flagged EAB_STORE_BF16; every reader of it is
    flagged for that operand; no other op touches it; and nothing is flagged in an fp32 program or with EAB_BF16_STORE=0."""
    from dataclasses import replace
    from eabnet_amd import program as prg
    cfg = replace(spec.NetConfig(), M=4, p=2, q=1)

    def flagged(prog):
        writes, reads = {}, {}
        for op in prog.fwd + prog.bwd:
            if isinstance(op, prg.ConvOp):
                for bit, ref in ((1, op.src0), (2, op.src1)):
                    if getattr(op, "src_bf16", 0) & bit:
                        reads.setdefault(ref.off, []).append(op)
            elif isinstance(op, train.WgradOp):
                for bit, ref in ((1, op.dz), (2, op.src0), (4, op.src1)):
                    if op.bf16_mask & bit:
                        reads.setdefault(ref.off, []).append(op)
            elif isinstance(op, train.GenOp) and len(op.i) > 3 and (op.i[3] & train.STORE_BF16):
                out = {train.OP_TR_NORM_ACT: 4, train.OP_NORM_BWD: 8, train.OP_GLU_BWD: 2}[op.kind]
                assert op.p[out].off not in writes, "one writer per bf16 tensor"
                writes[op.p[out].off] = op
        return writes, reads

    prog = train.lower_train(cfg, 2, 20, 161, "bf16")
    writes, reads = flagged(prog)
    assert len(writes) >= 20 and set(writes) == set(reads), "every bf16 tensor has its flagged writer and flagged readers"
    # nobody else reads or writes those regions
    offs = set(writes)
    for op in prog.fwd + prog.bwd:
        refs = []
        if isinstance(op, prg.ConvOp):
            refs = [(r, getattr(op, "src_bf16", 0) & b) for b, r in ((1, op.src0), (2, op.src1))] + \
                   [(getattr(op, f, None), 0) for f in ("aux", "dst", "dst_acc", "stats", "glu_dump")]
            assert not getattr(op, "src_bf16", 0) or (op.precision == prg.PREC_BF16 and op.korder == prg.KORDER_TAP and op.Fin > 1
                                                      and op.xf_mode == prg.XF_NONE and op.C0 % 16 == 0 and op.C1 % 16 == 0)
        elif isinstance(op, train.WgradOp):
            refs = [(r, op.bf16_mask & b) for b, r in ((1, op.dz), (2, op.src0), (4, op.src1))]
            assert not op.bf16_mask or op.precision == prg.PREC_BF16
        elif isinstance(op, train.GenOp):
            out = {train.OP_TR_NORM_ACT: 4, train.OP_NORM_BWD: 8, train.OP_GLU_BWD: 2}.get(op.kind)
            st = len(op.i) > 3 and (op.i[3] & train.STORE_BF16) and out is not None
            refs = [(r, 1 if (st and j == out) else 0) for j, r in enumerate(op.p)]
            if op.kind == train.OP_NORM_BWD and st:
                assert op.p[7] is None, "a bf16 gradient is never an accumulation target"
        for r, fl in refs:
            if r is not None and r.arena == "a" and r.off in offs:
                assert fl, f"{getattr(op, 'name', op)} touches a bf16 tensor without knowing it"
    # the S-TCN (1-D, small-tile kernel), the LSTM and the head stay fp32
    assert not any(getattr(op, "src_bf16", 0) for op in prog.fwd + prog.bwd if isinstance(op, prg.ConvOp) and op.Fin == 1)
    w32, r32 = flagged(train.lower_train(cfg, 2, 20, 161, "f32"))
    assert not w32 and not r32
    monkeypatch.setenv("EAB_BF16_STORE", "0")
    w0, r0 = flagged(train.lower_train(cfg, 2, 20, 161, "bf16"))
    assert not w0 and not r0


def test_bf16_programs_run_the_lstm_products_in_bf16(monkeypatch):
    """precision word of the two LSTM ops (eab_lstm64_train_fwd_prec_f32 / eab_lstm64_bwd_prec_f32): bf16 in a bf16 program,
    fp32 in an fp32 program and with EAB_BF16_LSTM=0 (the rounds 2-3 arithmetic, kept as the A/B switch)."""
    cfg = replace(spec.NetConfig(), M=4, p=2, q=1)

    def precs(prog):
        ops = [op for op in prog.fwd + prog.bwd if isinstance(op, train.GenOp) and op.kind in (train.OP_LSTM_TRAIN, train.OP_LSTM_BWD)]
        assert len(ops) == 4 and all(len(op.i) == 4 for op in ops)
        return {op.i[3] for op in ops}
    monkeypatch.delenv("EAB_BF16_LSTM", raising=False)
    assert precs(train.lower_train(cfg, 2, 20, 161, "bf16")) == {prg.PREC_BF16}
    assert precs(train.lower_train(cfg, 2, 20, 161, "f32")) == {prg.PREC_F32}
    monkeypatch.setenv("EAB_BF16_LSTM", "0")
    assert precs(train.lower_train(cfg, 2, 20, 161, "bf16")) == {prg.PREC_F32}
