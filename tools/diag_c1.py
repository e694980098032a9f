"""One single-utterance (C1: B = 1, M = 8, 4 s) forward per iteration, for kernel timelines:
rocprofv3 --kernel-trace -- python3 tools/diag_c1.py; prints ms per utterance."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eabnet_amd
dev = torch.device("cuda:0")
torch.manual_seed(2)
net = eabnet_amd.EaBNet(M=8).to(dev).eval()
x = 0.3 * torch.randn(1, 401, 161, 8, 2, device=dev)
with torch.no_grad():
    for _ in range(3):
        net(x)
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        y = net(x)
    torch.cuda.synchronize()
print(f"C1: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per utterance", flush=True)
