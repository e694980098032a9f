"""In-kernel phase timing of the small-tile convolution launches (csrc/conv_st.hip): every workgroup stamps s_memtime at
start / tables ready / A burst stored / barrier passed / main loop done / stores drained into a debug buffer
(eab_conv_desc.glu_dump, unused by that kernel otherwise).  Needs a diagnostic build of the library:
`make -C eabnet_amd/csrc clean && make -C eabnet_amd/csrc -j8 EXTRA=-DEAB_ST_STAMPS` (the production build refuses a glu_dump here).  Prints per launch kind the median phase durations in cycles.
    python tools/diag_st_stamps.py [B] [T]"""
import os, sys, collections
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eabnet_amd
from eabnet_amd import program as prg

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T = int(sys.argv[2]) if len(sys.argv) > 2 else 401
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = eabnet_amd.EaBNet(M=8).to(dev).eval()
net.use_graph = False
x = 0.3 * torch.randn(B, T, 161, 8, 2, device=dev)
with torch.no_grad():
    y_keep = net(x)          # keep the bound output buffer alive: the program keeps writing to it
    torch.cuda.synchronize()
    bound = net._last[0]
    st_ops = [k for k, o in enumerate(bound.prog.ops) if o.kind == prg.OP_CONV and o.korder == prg.KORDER_FRAG]
    NW = 8192
    buf = torch.zeros(len(st_ops), NW, 8, dtype=torch.int64, device=dev)
    for j, k in enumerate(st_ops):
        bound.ops[k].conv.glu_dump = buf[j].data_ptr()
    for _ in range(3):
        bound.run(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    h = buf.cpu().numpy()
groups = collections.defaultdict(list)
for j, k in enumerate(st_ops):
    o = bound.prog.ops[k]
    Tw = o.T
    n_wg = o.B * (prg.conv_tiles(Tw, o.No, o.bm) + (prg.conv_tiles(Tw, o.ph1_No, o.bm) if o.ph1_No else 0))
    s = h[j][:n_wg]
    if not (s[:, 0] > 0).all():
        print(o.name, "rows with stamps:", int((s[:, 0] > 0).sum()), "of", n_wg)
        print(s[:6]); print(s[450:460])
        nz = np.nonzero((h[j] != 0).any(1))[0]; print("nonzero rows:", nz[:10], nz[-10:], len(nz))
        break
    d = np.diff(s[:, :6].astype(np.float64), axis=1)
    span = (s[:, 5].max() - s[:, 0].min())
    key = (o.N, o.C0 + o.C1, len(o.dt), o.No, o.bm, o.ph1_No)
    groups[key].append((len(s), np.median(d, axis=0), span, o.name))
print("N C taps No bm ph1No | WGs | median cycles: tables, A-burst, barrier, mainloop, epilogue | first-start..last-end cycles")
for key, v in groups.items():
    n = v[0][0]
    med = np.median(np.stack([x[1] for x in v]), axis=0)
    span = np.median([x[2] for x in v])
    print(key, "|", n, "|", " ".join(f"{c:8.0f}" for c in med), "|", f"{span:9.0f} = {span / 2400:.1f} us at 2.4 GHz", v[0][3])
