"""The HBM-side kernels of the path alone, same protocol as bench.py's `hbm_kernels` block (HIP events around back-to-back
launches on the current stream) but with more repetitions: us per call and fraction of the 8 TB/s HBM peak on algorithmic bytes.
python tools/bench_hbm_kernels.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import eabnet_amd

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda:0")
B, M, L, T, F = 16, 8, 64000, 401, 161
torch.manual_seed(0)
wav = 0.05 * torch.randn(B, M, L, device=dev)
window = torch.hann_window(320)


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e-3)
    return best


with torch.no_grad():
    ns = eabnet_amd.stft_compress(wav, 320, 160, window)
    wts = torch.randn_like(ns)
    est = torch.randn(B, 2, T, F, device=dev)
    rows = [("stft_compress", lambda: eabnet_amd.stft_compress(wav, 320, 160, window), B * T * (160 * M * 4 + 161 * M * 2 * 4)),
            ("filter_sum", lambda: eabnet_amd.filter_and_sum(wts, ns), B * T * F * (4 * M + 2) * 4),
            ("istft", lambda: eabnet_amd.istft(est, 320, 160, window), B * T * (2 * F + 160) * 4)]
    # the head of the beam-former as the program runs it: w_dnn (Linear 64 -> 64 + ReLU, Linear 64 -> 2M) + filter-and-sum on the
    # LSTM output h, one launch (eab_mlp_bfw_filter_sum_f32); algorithmic bytes: h + X read, the estimate written
    import ctypes as C
    from eabnet_amd import _lib
    lib = _lib.load()
    h = torch.randn(B * T * F, 64, device=dev)
    w1, b1 = 0.1 * torch.randn(64, 64, device=dev), 0.1 * torch.randn(64, device=dev)
    w2, b2 = 0.1 * torch.randn(2 * M, 64, device=dev), 0.1 * torch.randn(2 * M, device=dev)
    out = torch.empty(B, 2, T, F, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def head():
        _lib.check(lib.eab_mlp_bfw_filter_sum_f32(h.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                                  ns.data_ptr(), out.data_ptr(), None, B, T, F, M, _lib.TimeWindow(None, 0), st),
                   "eab_mlp_bfw_filter_sum_f32")
    rows.append(("mlp_bfw_fs", head, B * T * F * (64 + 2 * M + 2) * 4))
    for name, fn, by in rows:
        t = timed(fn)
        print(f"{name:14s} {1e6 * t:8.2f} us  {by / t / 1e12:6.3f} TB/s  frac {by / t / 8e12:.3f}", flush=True)
