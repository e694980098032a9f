"""Co-run matrix (GPU box): does a kernel of program A compute the same result while a kernel of program B runs
on another stream?  Victim op replayed N times on stream 1 (its output snapshotted each time), aggressor op looped
on stream 2.  Two independent bound programs (own arenas)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import paramgen  # noqa: E402
import eabnet_amd  # noqa: E402
from eabnet_amd import program as prg  # noqa: E402
from eabnet_amd.model import _Bound  # noqa: E402
from eabnet_amd.spec import NetConfig, param_specs  # noqa: E402

dev = torch.device("cuda:0")
B, T, M = 16, 401, 8


def make(precision, seed):
    cfg = NetConfig(M=M)
    P = paramgen.make_params(param_specs(cfg), 5)
    prog = prg.lower(cfg, P, B, T, 161, precision=precision)
    bound = _Bound(prog, dev)
    x = torch.from_numpy(paramgen.make_spec_input(B, T, 161, M, seed)).to(dev)
    out = torch.empty(B, 2, T, 161, device=dev)
    bound.bind(x.data_ptr(), out.data_ptr())
    bound.run(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return bound, x, out


def out_region(bound, out, op):
    if op.kind == prg.OP_CONV:
        return bound.acts[op.dst.off:op.dst.off + op.B * op.T * op.Fout * op.Cout]
    if op.kind == prg.OP_LSTM64:
        return bound.acts[op.h_out.off:op.h_out.off + op.B * op.T * op.F * 64]
    if op.kind == prg.OP_BFW_FS:
        return out.view(-1)
    if op.kind == prg.OP_IN_FINALIZE:
        return bound.acts[op.xf0.off:op.xf0.off + op.B * op.C * 2]
    if op.kind == prg.OP_NORM_ACT:
        return bound.acts[op.out.off:op.out.off + op.B * op.P * op.C]
    raise ValueError(op.kind)


def find(bound, name):
    for k, op in enumerate(bound.prog.ops):
        if op.name == name:
            return k
    raise KeyError(name)


def corun(A, B_, vname, aname, reps=24, detail=False):
    (ba, xa, oa), (bb, xb, ob) = A, B_
    kv, ka = find(ba, vname), find(bb, aname)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    reg = out_region(ba, oa, ba.prog.ops[kv])
    n = min(reg.numel(), 1 << 24)
    ref = reg.clone()
    snaps = torch.empty(reps, n, device=dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(s2):
        for _ in range(reps * 6):
            bb.run(s2.cuda_stream, ka, 1)
    with torch.cuda.stream(s1):
        for r in range(reps):
            ba.run(s1.cuda_stream, kv, 1)
            snaps[r].copy_(reg[:n])
    torch.cuda.synchronize()
    bad = [(r, int((snaps[r] != ref[:n]).sum())) for r in range(reps) if not torch.equal(snaps[r], ref[:n])]
    print(f"victim {vname:28s} | aggressor {aname:28s}: {len(bad)}/{reps} runs differ {bad[:6]}", flush=True)
    if bad and detail and ba.prog.ops[kv].kind == prg.OP_BFW_FS:
        r = bad[0][0]
        idx = (snaps[r] != ref[:n]).nonzero().flatten()
        F_, T_ = 161, T
        f = idx % F_; t = (idx // F_) % T_; pl = (idx // (F_ * T_)) % 2; b = idx // (F_ * T_ * 2)
        bins = (b * T_ + t) * F_ + f
        tiles = bins // 64
        wg = tiles % 1024
        rr = bins % 64
        import collections
        print("   planes:", collections.Counter(pl.tolist()), " rows-in-tile mod 16:", sorted(collections.Counter((rr % 16).tolist()).items()),
              " waves:", sorted(collections.Counter((rr // 16).tolist()).items()))
        print("   distinct WGs:", len(set(wg.tolist())), " distinct tiles:", len(set(tiles.tolist())), " tile passes (tile//1024):",
              sorted(collections.Counter((tiles // 1024).tolist()).items()))
        for j in idx[:6].tolist():
            print(f"   idx {j}: got {float(snaps[r][j]):+.6e} want {float(ref[j]):+.6e}")
    return bad


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "bfw":
    A, B_ = make("f16x3", 11), make("f16x3", 12)
    print("EAB_BFW_VAR =", os.environ.get("EAB_BFW_VAR"))
    for a in ("bf_map.rnn2", "de.last_conv.ph0"):
        corun(A, B_, "bf_map.w_dnn+fs", a, reps=8, detail=True)
    sys.exit(0)

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "mixed":
    A, B_ = make("f32", 11), make("f16x3", 12)        # fp32-MFMA victims, f16-MFMA aggressors
    for v in ("bf_map.w_dnn+fs", "bf_map.rnn2", "de.last_conv.ph0", "en.meta_unet_list.1.in_conv", "en.meta_unet_list.1.enco.0.conv",
              "stcns.0.tcm_list.0.lr_conv", "de.meta_unet_list.3.in_conv.ph0"):
        for a in ("bf_map.rnn2", "de.last_conv.ph0"):
            corun(A, B_, v, a, reps=8, detail=True)
    A2 = make("f32", 13)
    for a in ("bf_map.rnn2", "de.last_conv.ph0"):
        corun(A, A2, "bf_map.w_dnn+fs", a, reps=8, detail=True)      # fp32 aggressors
    sys.exit(0)

if __name__ == "__main__":
    prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
    A, B_ = make(prec, 11), make(prec, 12)
    names = [op.name for op in A[0].prog.ops]
    victims = ["bf_map.w_dnn+fs", "bf_map.rnn2", "bf_map.rnn1", "de.last_conv.ph0", "en.meta_unet_list.1.in_conv",
               "en.meta_unet_list.1.enco.0.conv", "stcns.0.tcm_list.0.lr_conv", "en.meta_unet_list.0.in_conv.in", "de.last_conv"]
    aggr = ["bf_map.rnn2", "de.last_conv.ph0", "en.meta_unet_list.1.enco.0.conv", "bf_map.w_dnn+fs", "stcns.0.tcm_list.0.lr_conv"]
    for v in victims:
        for a in aggr:
            if v in names and a in names:
                corun(A, B_, v, a)
