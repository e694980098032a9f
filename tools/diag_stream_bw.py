"""The HBM-side training kernels at configs[3] sizes (B = 6, T = 601, C = 64): microseconds per call and the algorithmic bytes
they move per second.  norm_bwd = reduce + apply (two launches), bytes = 2 reads of (dy, x) + 1 write of dx."""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from eabnet_amd import _lib      # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
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
    B, T, Cc = 6, 601, 64
    for F in (161, 79, 39, 19, 9, 4):
        P = T * F
        g = torch.Generator().manual_seed(F)
        x, dy = torch.randn(B, P, Cc, generator=g).cuda(), torch.randn(B, P, Cc, generator=g).cuda()
        gam, bet, slp = torch.rand(Cc).cuda() + 0.5, torch.randn(Cc).cuda(), torch.rand(Cc).cuda() * 0.3
        xf, mr = torch.empty(B, Cc, 2).cuda(), torch.empty(B, Cc, 2).cuda()
        _lib.check(lib.eab_train_in_stats_f32(x.data_ptr(), None, B, P, Cc, 1e-5, gam.data_ptr(), bet.data_ptr(), xf.data_ptr(),
                                              mr.data_ptr(), st), "stats")
        y, dx = torch.empty_like(x), torch.empty_like(x)
        sums = torch.zeros(8, B, Cc, 4).cuda()
        dg, db, ds = (torch.zeros(Cc).cuda() for _ in range(3))
        nbytes = x.numel() * 4
        t_act = timed(lambda: lib.eab_train_norm_act_f32(x.data_ptr(), xf.data_ptr(), slp.data_ptr(), None, y.data_ptr(), B, P, Cc, 1, st))
        t_bwd = timed(lambda: lib.eab_train_norm_bwd_f32(dy.data_ptr(), x.data_ptr(), mr.data_ptr(), gam.data_ptr(), bet.data_ptr(),
                                                         slp.data_ptr(), sums.data_ptr(), None, dx.data_ptr(), dg.data_ptr(),
                                                         db.data_ptr(), ds.data_ptr(), B, P, Cc, 1 | 0x100, st))
        dump, dz = torch.randn(B * P, 2 * Cc, generator=g).cuda(), torch.empty(B * P, 2 * Cc).cuda()
        t_glu = timed(lambda: lib.eab_glu_bwd_f32(dy.data_ptr(), dump.data_ptr(), dz.data_ptr(), B * P, 2 * Cc, st))
        t_na = timed(lambda: lib.eab_norm_act_f32(x.data_ptr(), xf.data_ptr(), slp.data_ptr(), None, None, None, y.data_ptr(), B, P, Cc, st))
        print(f"F={F:3d} ({nbytes / 1e6:6.1f} MB/tensor)  norm_act {t_act:6.1f} us {2 * nbytes / t_act / 1e6:5.2f} TB/s | "
              f"norm_bwd {t_bwd:6.1f} us {5 * nbytes / t_bwd / 1e6:5.2f} TB/s | glu_bwd {t_glu:6.1f} us {5 * nbytes / t_glu / 1e6:5.2f} TB/s | "
              f"norm_act(inference) {t_na:6.1f} us {2 * nbytes / t_na / 1e6:5.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
