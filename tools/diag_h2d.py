"""Where does prepare_data from host memory spend its time? (GPU box)"""
import time, torch, argparse, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eabnet_amd
dev = torch.device("cuda:0")
x = 0.05 * torch.randn(16, 8, 64000)
xp = x.pin_memory()
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
mb = x.numel() * 4 / 1e6
for name, src in (("pageable", x), ("pinned", xp)):
    ms = t(lambda: src.to(dev))
    print(f"{name}: .to(device) {ms:.2f} ms = {mb / ms:.2f} GB/s")
    ms = t(lambda: src.to(dev, non_blocking=True))
    print(f"{name}: .to(device, non_blocking) {ms:.2f} ms = {mb / ms:.2f} GB/s")
buf = torch.empty_like(x, device=dev)
ms = t(lambda: buf.copy_(xp, non_blocking=True)); print(f"pinned copy_ into resident buffer: {ms:.2f} ms = {mb/ms:.2f} GB/s")
for sz in (1, 4, 16, 64):
    a = torch.empty(sz * 1024 * 1024 // 4).pin_memory(); b = torch.empty_like(a, device=dev)
    ms = t(lambda: b.copy_(a, non_blocking=True)); print(f"pinned {sz} MB: {ms:.3f} ms = {sz*1.048576/ms:.2f} GB/s")
args = argparse.Namespace(mics=8, sr=16000, wav_len=4.0, win_size=0.020, win_shift=0.010, fft_num=320)
for name, src in (("pageable", x), ("pinned", xp)):
    ms = t(lambda: eabnet_amd.prepare_data(src, src[:, :1], dev, args))
    print(f"prepare_data from {name}: {ms:.2f} ms")
xd = x.to(dev)
ms = t(lambda: eabnet_amd.prepare_data(xd, xd[:, :1], dev, args)); print(f"prepare_data resident: {ms:.3f} ms")
ms = t(lambda: torch.hann_window(320)); print(f"hann_window cpu: {ms:.3f} ms")
tgt = xp[:, :1]
ms = t(lambda: tgt.to(dev)); print(f"target slice (non-contiguous pinned view) .to: {ms:.3f} ms")
