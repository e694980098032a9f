"""Whole-network step time for several batch sizes under EAB_LSTM_G (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, eabnet_amd
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = eabnet_amd.EaBNet(M=8).to(dev).eval()
for B in (1, 2, 4, 8, 12, 16):
    x = 0.3 * torch.randn(B, 401, 161, 8, 2, device=dev)
    with torch.no_grad():
        for _ in range(3):
            net(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            net(x)
        torch.cuda.synchronize()
    print(f"G={os.environ.get('EAB_LSTM_G', 'auto')} B={B}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms", flush=True)
