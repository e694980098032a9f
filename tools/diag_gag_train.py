"""Per-kernel-family time of the post-filter's two training programs at the benchmark shape (6 x 6 s):
python tools/diag_gag_train.py [f32|bf16]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eabnet_amd
from eabnet_amd import program as prg, train as tr

prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = eabnet_amd.GaGNet().to(dev).train()
net.precision = prec
B, T, F = 6, 601, 161
a, b = 0.3 * torch.randn(B, 2, T, F, device=dev), 0.3 * torch.randn(B, 2, T, F, device=dev)
lab = 0.3 * torch.randn(B, 2, F, T, device=dev)
for _ in range(3):
    net.zero_grad(set_to_none=True)
    loss = eabnet_amd.stagewise_com_mag_mse_loss(net(a, b), lab, [T] * B)
    loss.backward()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    net.zero_grad(set_to_none=True)
    loss = eabnet_amd.stagewise_com_mag_mse_loss(net(a, b), lab, [T] * B)
    loss.backward()
torch.cuda.synchronize()
print(f"{prec}: forward + loss + backward {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")
bound = next(iter(net._train_bound.values()))
prog = bound.prog
stream = torch.cuda.current_stream()
names = {prg.OP_CONV: "conv", tr.OP_WGRAD: "wgrad", tr.OP_NORM_BWD: "norm_bwd", tr.OP_IN_STATS: "in1d", tr.OP_TR_NORM_ACT: "norm_act",
         tr.OP_GLU_BWD: "glu_bwd", prg.OP_IN_FINALIZE: "in_finalize", tr.OP_ADD: "add"}
for which, ops in (("fwd", prog.fwd), ("bwd", prog.bwd)):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(ops) + 1)]
    if which == "bwd":
        bound.g.zero_()
    for k in range(len(ops)):
        evs[k].record(stream)
        bound.run(which, stream.cuda_stream, k, 1)
    evs[-1].record(stream)
    torch.cuda.synchronize()
    ms = np.array([evs[k].elapsed_time(evs[k + 1]) for k in range(len(ops))])
    by, cnt = {}, {}
    for k, o in enumerate(ops):
        n = names.get(o.kind, f"kind{o.kind}")
        by[n] = by.get(n, 0.0) + float(ms[k])
        cnt[n] = cnt.get(n, 0) + 1
    print(which, f"{ms.sum():.2f} ms, {len(ops)} ops:", {k: (round(v, 2), cnt[k]) for k, v in sorted(by.items(), key=lambda kv: -kv[1])})
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if which == "bwd":
        bound.g.zero_()
    e0.record(stream)
    bound.run(which, stream.cuda_stream)
    e1.record(stream)
    torch.cuda.synchronize()
    print(which, f"whole program (wgrads batched): {e0.elapsed_time(e1):.2f} ms")
