"""Kernel timeline of one frame-synchronous streaming step (BASELINE configs[4] shape: B=1, M=16, 8 s, BN norms):
rocprofv3 --kernel-trace -- python3 tools/diag_stream.py [chunk] [precision]; prints ms per step."""
import sys, time
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eabnet_amd

chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 1
prec = sys.argv[2] if len(sys.argv) > 2 else "f32"
dev = torch.device("cuda:0")
torch.manual_seed(2)
net = eabnet_amd.EaBNet(M=16, norm_type="BN").to(dev).eval()
net.precision = prec
st = net.stream_begin(1, T_max=801, chunk=chunk)
x = 0.3 * torch.randn(1, chunk, 161, 16, 2, device=dev)
for _ in range(3):
    st.step(x)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    y = st.step(x)
torch.cuda.synchronize()
print(f"chunk={chunk} {prec}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per step", flush=True)
