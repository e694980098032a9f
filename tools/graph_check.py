import sys, torch
sys.path.insert(0, ".")
import bench, eabnet_amd
dev = torch.device("cuda:0")
net, state = bench.make_model(8, dev)
wav = bench.synth_waves(16, 8, 64000, 1234).to(dev)
win = torch.hann_window(320)
def step():
    ns = eabnet_amd.stft_compress(wav, 320, 160, win)
    return net(ns)
def rel(a, b): return float((a - b).abs().max() / b.abs().max())
with torch.no_grad():
    for prec in ("f32", "f16x3"):
        net.precision = prec
        net.use_graph = False
        ref = step(); torch.cuda.synchronize()
        net.use_graph = True
        ys = [step() for _ in range(6)]          # no sync in between
        torch.cuda.synchronize()
        print(prec, "graph async vs eager:", ["%.1e" % rel(y, ref) for y in ys])
        ys = []
        for _ in range(4):
            ys.append(step()); torch.cuda.synchronize()
        print(prec, "graph sync  vs eager:", ["%.1e" % rel(y, ref) for y in ys])
