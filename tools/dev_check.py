import sys, os, torch, numpy as np
sys.path.insert(0, ".")
import bench, eabnet_amd
from oracle import eabnet_oracle as orc
dev = torch.device("cuda:0")
torch.set_num_threads(16)
net, state = bench.make_model(8, dev)
B = int(os.environ.get("DEV_B", "4"))
wav = bench.synth_waves(B, 8, 64000, 1234)
win = torch.hann_window(320)
with torch.no_grad():
    ns_ref, _ = orc.prepare_data_oracle(wav, None)
    ref = orc.eabnet_forward(state, ns_ref, fast_lstm=True)
    ref64 = orc.eabnet_forward({k: v.double() for k, v in state.items()}, ns_ref.double(), fast_lstm=True)
    ns = eabnet_amd.stft_compress(wav.to(dev), 320, 160, win)
    outs = {}
    for prec in ("f32", "f16x3"):
        for graph in (True, False):
            net.precision = prec; net.use_graph = graph
            outs[(prec, graph)] = net(ns).cpu()
def err(a, b):
    a = a.double(); b = b.double()
    return float((a - b).abs().max() / b.abs().max())
print("patch env", os.environ.get("EAB_PATCH"))
print("oracle fp32 vs fp64:", err(ref, ref64))
for k, v in outs.items():
    print(k, "vs fp64 oracle: %.2e" % err(v, ref64), " per-utterance:", ["%.1e" % err(v[i], ref64[i]) for i in range(B)])
print("f16x3 vs f32 (graph):", err(outs[("f16x3", True)], outs[("f32", True)]))
