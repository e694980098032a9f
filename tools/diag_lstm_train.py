"""The training LSTM kernels at configs[3] size (6 x 161 = 966 sequences, 601 steps) and at 16 utterances: microseconds per layer
and per step, forward (gates stored) and reverse time, fp32 and bf16 recurrent products."""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from eabnet_amd import _lib      # noqa: E402


def timed(fn, n=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    lib = _lib.load()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for B, T, F in ((6, 601, 161), (16, 401, 161)):
        g = torch.Generator().manual_seed(1)
        x = torch.randn(B, T, F, 64, generator=g).cuda()
        dh = torch.randn(B, T, F, 64, generator=g).cuda()
        w = (torch.randn(256, 128, generator=g) * 0.1).cuda()
        b = torch.zeros(256).cuda()
        h, gates, dg = torch.empty_like(x), torch.empty(B * F, T, 5, 64).cuda(), torch.empty(B, T, F, 256).cuda()
        out = []
        for prec, name in ((0, "f32"), (2, "bf16")):
            tf = timed(lambda: lib.eab_lstm64_train_fwd_prec_f32(x.data_ptr(), w.data_ptr(), b.data_ptr(), h.data_ptr(), gates.data_ptr(),
                                                                 B, T, F, prec, st))
            tb = timed(lambda: lib.eab_lstm64_bwd_prec_f32(gates.data_ptr(), dh.data_ptr(), w.data_ptr(), dg.data_ptr(), B, T, F, prec, st))
            out.append(f"{name}: fwd {tf:7.1f} us ({tf / T:5.2f}/step)  bwd {tb:7.1f} us ({tb / T:5.2f}/step)")
        # the same layers without the gate store (the inference entry: same kernels, DUMP = false)
        for prec, name in ((0, "f32"), (2, "bf16")):
            ti = timed(lambda: lib.eab_lstm64_prec_f32(x.data_ptr(), None, None, 0.0, w.data_ptr(), b.data_ptr(), h.data_ptr(), B, T, F, prec, st))
            out.append(f"{name} no gate store: fwd {ti:7.1f} us ({ti / T:5.2f}/step)")
        print(f"B={B} T={T} S={B * F}: " + " | ".join(out), flush=True)


if __name__ == "__main__":
    main()
