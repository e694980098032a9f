"""wgrad_bf_kernel with fp32- vs bf16-STORED operands at a training-size geometry: which operand's storage costs time?
mask bit 0 = dz, bits 1|2 = x (both sources).  Prints ms per launch (HIP events, 20 launches each)."""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from eabnet_amd import _lib      # noqa: E402


def run(lib, dz, x0, x1, N, C0, C1, taps, B, T, Fin, Fz, No, ostride, istride, mask):
    d = _lib.WgradDesc()
    upt = (C0 + C1 + 15) // 16
    Kpad = len(taps) * upt * 16
    dw = torch.zeros(N, Kpad, device="cuda:0")
    d.dz, d.src0, d.src1, d.dw = dz.data_ptr(), x0.data_ptr(), (x1.data_ptr() if x1 is not None else None), dw.data_ptr()
    d.N, d.C0, d.C1, d.Kpad = N, C0, C1, Kpad
    d.B, d.T, d.Fin, d.Fz, d.No, d.ostride, d.ophase, d.istride = B, T, Fin, Fz, No, ostride, 0, istride
    d.ntaps, d.precision, d.bf16_mask = len(taps), 2, mask
    for j, (a, c) in enumerate(taps):
        d.dt[j], d.ioff[j] = a, c
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        _lib.check(lib.eab_wgrad_f32(C.byref(d), st), "wgrad")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lib.eab_wgrad_f32(C.byref(d), st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20


def main():
    lib = _lib.load()
    B, T = 6, 601
    cases = [("unit conv 64<-64, (1,3)/2, Fin 79", 64, 64, 0, [(0, 0), (0, 1), (0, 2)], 79, 39, 39, 1, 2),
             ("gated conv 128<-64, (2,3)/2, Fin 79", 128, 64, 0, [(-1, 0), (-1, 1), (-1, 2), (0, 0), (0, 1), (0, 2)], 79, 39, 39, 1, 2),
             ("deconv unit ph0 64<-64+64, Fin 39", 64, 64, 64, [(0, 0), (0, -1)], 39, 79, 40, 2, 1),
             ("gated deconv ph0 128<-64+64 (2,3), Fin 79", 128, 64, 64, [(0, 0), (0, -1), (-1, 0), (-1, -1)], 79, 161, 80, 2, 1)]
    for name, N, C0, C1, taps, Fin, Fz, No, ostride, istride in cases:
        g = torch.Generator().manual_seed(1)
        dz = torch.randn(B, T, Fz, N, generator=g).cuda()
        x0 = torch.randn(B, T, Fin, C0, generator=g).cuda()
        x1 = torch.randn(B, T, Fin, C1, generator=g).cuda() if C1 else None
        h = lambda t: None if t is None else t.bfloat16().contiguous()      # noqa: E731
        out = []
        for mask in (0, 1, 6 if C1 else 2, 7 if C1 else 3):
            a = h(dz) if mask & 1 else dz
            b0 = h(x0) if mask & 2 else x0
            b1 = h(x1) if mask & 2 else x1
            out.append((mask, run(lib, a, b0, b1, N, C0, C1, taps, B, T, Fin, Fz, No, ostride, istride, mask)))
        print(name, " ".join(f"mask {m}: {ms * 1e3:.1f} us" for m, ms in out), flush=True)


if __name__ == "__main__":
    main()
