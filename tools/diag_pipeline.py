"""Diagnostic for pipelined-vs-direct mismatches (GPU box): different inputs per step so that stale reads are
visible, error geometry per bad result, stale-output test, graph vs direct launches, per-kernel-class precision."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import paramgen  # noqa: E402
import eabnet_amd  # noqa: E402
from util import torch_params  # noqa: E402

dev = torch.device("cuda:0")
win = torch.hann_window(320)


def run(tag, precision, use_graph, steps=14, ncycle=3):
    net = eabnet_amd.EaBNet(M=8)
    net.load_state_dict(torch_params(8, 5), strict=True)
    net = net.to(dev).eval()
    net.precision, net.use_graph = precision, use_graph
    wavs = [torch.from_numpy(paramgen.make_wave(16, 8, 64000, 900 + i)).to(dev) for i in range(ncycle)]
    with torch.no_grad():
        wants = [net(eabnet_amd.stft_compress(w, 320, 160, win)).clone() for w in wavs]
        pipe = eabnet_amd.Pipeline(net, depth=2, front_end=(320, 160, win))
        got = []
        for k in range(steps):
            if pipe.outstanding == 2:
                got.append(pipe.collect())
            pipe.submit(wavs[k % ncycle])
        while pipe.outstanding:
            got.append(pipe.collect())
        torch.cuda.synchronize()
    bad = [k for k, y in enumerate(got) if not torch.equal(y, wants[k % ncycle])]
    print(f"[{tag}] precision={precision} graph={use_graph} lstm={os.environ.get('EAB_LSTM_PREC', '-')}: bad {bad} of {len(got)}", flush=True)
    for k in bad[:4]:
        w = wants[k % ncycle]
        d = got[k] != w
        bins = d.any(dim=1)                                     # (B, T, F)
        seqs = bins.any(dim=1).nonzero()                        # (b, f) pairs
        nb = int(bins.sum())
        msg = f"   result {k} (slot {k % 2}): {int(d.sum())} values, {nb} bins, {seqs.shape[0]} (b,f) sequences; max-rel {float((got[k] - w).abs().max() / w.abs().max()):.2e}"
        for b, f in seqs[:5].tolist():
            tt = bins[b, :, f].nonzero().flatten()
            msg += f"\n      (b={b}, f={f}): {tt.numel()} frames, t {int(tt.min())}..{int(tt.max())}"
        if k >= 2:
            stale = got[k][d] == got[k - 2][d]                  # values of the same slot's previous output?
            msg += f"\n      equal to the slot's previous output at the bad positions: {int(stale.sum())} of {int(d.sum())}"
            prevw = wants[(k - 2) % ncycle]
            msg += f"; equal to the previous step's CORRECT output: {int((got[k][d] == prevw[d]).sum())}"
        print(msg, flush=True)
    return bad


if __name__ == "__main__":
    which = sys.argv[1:] or ["a", "b", "c", "d", "e"]
    if "a" in which:
        run("a", "f16x3", True)
    if "b" in which:
        run("b", "f16x3", False)
    if "c" in which:
        os.environ["EAB_LSTM_PREC"] = "f32"
        run("c", "f16x3", True)
        os.environ.pop("EAB_LSTM_PREC")
    if "d" in which:
        run("d", "f32", True)
    if "e" in which:
        os.environ["EAB_LSTM_PREC"] = "f16x3"
        run("e", "f32", True)
        os.environ.pop("EAB_LSTM_PREC")
