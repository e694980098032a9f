"""STFT kernel timing (HIP events) and a statement-by-statement timing of prepare_data from host memory (GPU box)."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eabnet_amd
dev = torch.device("cuda:0")
win = torch.hann_window(320)
x = 0.05 * torch.randn(16, 8, 64000)
xd = x.to(dev)
def ev(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
us = ev(lambda: eabnet_amd.stft_compress(xd, 320, 160, win))
by = 16 * 401 * (160 * 8 * 4 + 161 * 8 * 2 * 4)
print(f"EAB_STFT_FR={os.environ.get('EAB_STFT_FR')}: stft_compress {us:.1f} us = {by / us / 1e3:.0f} GB/s = {by / us / 1e3 / 8000:.3f} of 8 TB/s")
if os.environ.get("EAB_STFT_FR"):
    sys.exit(0)
def wall(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
xp = x.pin_memory()
args = argparse.Namespace(mics=8, sr=16000, wav_len=4.0, win_size=0.020, win_shift=0.010, fft_num=320)
for name, src in (("pageable", x), ("pinned", xp)):
    tg = src[:, :1]
    print(name, "x.to(dev).contiguous().view:", f"{wall(lambda: src.to(dev).contiguous().view(16, 8, -1)):.2f} ms")
    print(name, "target.to(dev).reshape:", f"{wall(lambda: tg.to(dev).reshape(16, 1, -1)):.2f} ms")
    nw = src.to(dev); tw_ = tg.to(dev).reshape(16, 1, -1)
    print(name, "stft noisy:", f"{wall(lambda: eabnet_amd.stft_compress(nw, 320, 160, win, 0)):.2f} ms",
          " stft target:", f"{wall(lambda: eabnet_amd.stft_compress(tw_, 320, 160, win, 1)):.2f} ms")
    print(name, "prepare_data:", f"{wall(lambda: eabnet_amd.prepare_data(src, tg, dev, args)):.2f} ms")
