"""Time the post-filter (GaGNet) and the two-stage model at the bench shape (16 x 4 s x 8 mics)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import eabnet_amd

dev = torch.device("cuda:0")
B, T, F, M = 16, 401, 161, 8
torch.manual_seed(0)
gag = eabnet_amd.GaGNet().to(dev).eval()
a, b = 0.3 * torch.randn(B, 2, T, F, device=dev), 0.3 * torch.randn(B, 2, T, F, device=dev)


def timed(fn, n=10):
    with torch.no_grad():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for prec in ("f32", "f16x3"):
    gag.precision = prec
    ms = timed(lambda: gag(a, b))
    print(f"GaGNet {prec}: {ms:.3f} ms/step  {B * T / ms * 1e3:.0f} frames/s", flush=True)
if "--per-op" in sys.argv:
    gag.precision = "f32"
    gag.use_graph = False
    with torch.no_grad():
        gag(a, b)
    bound = next(iter(gag._bound.values()))
    ops = bound.prog.ops
    s = torch.cuda.current_stream()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(ops) + 1)]
    ev[0].record()
    for k in range(len(ops)):
        bound.run(s.cuda_stream, k, 1)
        ev[k + 1].record()
    torch.cuda.synchronize()
    by = {}
    for k, op in enumerate(ops):
        nm = op.name.split(".")[-1] if op.kind == 1 else type(op).__name__
        if op.kind == 1 and op.name.startswith("en."):
            nm = "en.conv"
        by.setdefault(nm, [0, 0.0])
        by[nm][0] += 1
        by[nm][1] += ev[k].elapsed_time(ev[k + 1])
    for nm, (n, ms) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        print(f"  {nm:24s} n={n:4d} total {ms:7.3f} ms  avg {1e3 * ms / n:7.1f} us")
