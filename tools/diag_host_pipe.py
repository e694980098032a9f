"""Where does a host-input step go?  (bench.py next_rows.pcie_inclusive; eabnet_amd.Pipeline(prepare=args))
Times, at the headline size (16 x 4 s x 8 mics): the host memcpy into the pinned ring alone, the copy kernel alone, the blocking
prepare_data + network step, and the pipelined loop with the host time spent inside submit() / collect()."""
import argparse
import sys
import time

import torch

sys.path.insert(0, ".")
import eabnet_amd                                    # noqa: E402
from eabnet_amd import model as mdl                  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    B, M, L = 16, 8, 64000
    net = eabnet_amd.EaBNet(M=M).to(dev).eval()
    args = argparse.Namespace(mics=M, sr=16000, wav_len=4.0, win_size=0.020, win_shift=0.010, fft_num=320)
    wav = 0.05 * torch.randn(B, M, L)
    tgt = wav[:, :1].contiguous()
    pin = torch.empty_like(wav).pin_memory()
    for name, src in (("pageable", wav), ("pinned", wav.clone().pin_memory())):
        t0 = time.perf_counter()
        for _ in range(10):
            pin.copy_(src)
        print(f"host memcpy {name} -> pinned, 32.8 MB: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms", flush=True)
    print("torch threads", torch.get_num_threads(), flush=True)
    with torch.no_grad():
        for _ in range(3):
            net(eabnet_amd.prepare_data(wav, tgt, dev, args)[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            net(eabnet_amd.prepare_data(wav, tgt, dev, args)[0])
        torch.cuda.synchronize()
        print(f"blocking step: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms", flush=True)
        xd = wav.to(dev)
        td = tgt.to(dev)
        for _ in range(3):
            net(eabnet_amd.prepare_data(xd, td, dev, args)[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            net(eabnet_amd.prepare_data(xd, td, dev, args)[0])
        torch.cuda.synchronize()
        print(f"resident step (one at a time): {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms", flush=True)
        for depth in (2, 3):
            for label, a, b in (("host", wav, tgt), ("resident", xd, td)):
                pipe = eabnet_amd.Pipeline(net, depth=depth, prepare=args)
                for _ in range(depth + 2):
                    pipe.submit(a, b)
                    pipe.collect()
                torch.cuda.synchronize()
                ts, tc = 0.0, 0.0
                t0 = time.perf_counter()
                for _ in range(12):
                    if pipe.outstanding == depth:
                        q = time.perf_counter()
                        pipe.collect()
                        tc += time.perf_counter() - q
                    q = time.perf_counter()
                    pipe.submit(a, b)
                    ts += time.perf_counter() - q
                while pipe.outstanding:
                    pipe.collect()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 12
                print(f"pipelined depth {depth}, {label} input: {dt * 1e3:.2f} ms / step; host time in submit {ts / 12 * 1e3:.2f} ms, "
                      f"in collect {tc / 12 * 1e3:.2f} ms", flush=True)
                pipe = None
        # the staging pieces alone
        st = mdl._STAGERS[str(dev)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            d_, ev = st.upload(wav)
            ev.record(torch.cuda.current_stream())
        torch.cuda.synchronize()
        print(f"stager.upload alone (memcpy + copy kernel): {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms", flush=True)


if __name__ == "__main__":
    main()
