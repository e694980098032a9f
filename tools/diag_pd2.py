import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eabnet_amd
from eabnet_amd import model as mdl
dev = torch.device("cuda:0")
x = 0.05 * torch.randn(16, 8, 64000)
xp = x.pin_memory()
args = argparse.Namespace(mics=8, sr=16000, wav_len=4.0, win_size=0.020, win_shift=0.010, fft_num=320)
def wall(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    return (t1 - t0) / n * 1e3, (time.perf_counter() - t0) / n * 1e3
tgc = xp[:, :1].contiguous().pin_memory()
for label, src, tg, stage in (("pageable", x, x[:, :1], False), ("pinned direct, strided target", xp, xp[:, :1], False),
                              ("pinned direct, contiguous pinned target", xp, tgc, False), ("pinned staged", xp, xp[:, :1], True),
                              ("pageable again", x, x[:, :1], False)):
    mdl._HostStager.always_stage = stage
    h, w = wall(lambda: eabnet_amd.prepare_data(src, tg, dev, args))
    print(f"{label}: host {h:.2f} ms/call, wall {w:.2f} ms/call")
# pieces for the pinned-direct case
st = mdl._STAGERS[str(dev)]
mdl._HostStager.always_stage = False
h, w = wall(lambda: st.upload(xp)); print(f"upload(xp) alone: host {h:.2f} wall {w:.2f}")
h, w = wall(lambda: st.upload(xp[:, :1])); print(f"upload(strided pinned target) alone: host {h:.2f} wall {w:.2f}")
h, w = wall(lambda: st.upload(x[:, :1])); print(f"upload(strided pageable target) alone: host {h:.2f} wall {w:.2f}")
t = torch.empty(16, 1, 64000).pin_memory()
h, w = wall(lambda: t.copy_(xp[:, :1])); print(f"host copy pinned strided -> pinned: {h:.2f}")
h, w = wall(lambda: t.copy_(x[:, :1])); print(f"host copy pageable strided -> pinned: {h:.2f}")
