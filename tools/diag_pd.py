import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eabnet_amd
from eabnet_amd import model as mdl
dev = torch.device("cuda:0")
x = 0.05 * torch.randn(16, 8, 64000)
tg = x[:, :1]
args = argparse.Namespace(mics=8, sr=16000, wav_len=4.0, win_size=0.020, win_shift=0.010, fft_num=320)
def tick(label, t0):
    torch.cuda.synchronize(); t1 = time.perf_counter(); print(f"  {label}: {(t1 - t0) * 1e3:.2f} ms"); return time.perf_counter()
for it in range(3):
    print("iteration", it)
    t0 = time.perf_counter()
    noisy_wav = x.to(dev).contiguous().view(16, 8, -1); t0 = tick("x.to", t0)
    target_wav = tg.to(dev).reshape(16, 1, -1); t0 = tick("target.to", t0)
    window = torch.hann_window(320); t0 = tick("hann", t0)
    wd = mdl._device_window(window, noisy_wav.device); t0 = tick("_device_window", t0)
    tw = mdl._twiddle(320, noisy_wav.device); t0 = tick("_twiddle", t0)
    a = eabnet_amd.stft_compress(noisy_wav, 320, 160, window, 0); t0 = tick("stft noisy", t0)
    b = eabnet_amd.stft_compress(target_wav, 320, 160, window, 1); t0 = tick("stft target", t0)
    t0 = time.perf_counter(); r = eabnet_amd.prepare_data(x, tg, dev, args); t0 = tick("prepare_data whole", t0)
    del r
    t0 = time.perf_counter(); r = eabnet_amd.prepare_data(x, tg, dev, args); del r; r = eabnet_amd.prepare_data(x, tg, dev, args); t0 = tick("two calls back to back", t0)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5): r = eabnet_amd.prepare_data(x, tg, dev, args)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
