"""why does (or doesn't) the streaming program chain its S-TCN launches: per-op plan result (debug aid)"""
import sys, os, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eabnet_amd
from eabnet_amd import _lib, program as prg
dev = torch.device("cuda:0")
net = eabnet_amd.EaBNet(M=16, norm_type="BN").to(dev).eval()
net.precision = sys.argv[1] if len(sys.argv) > 1 else "f32"
st = net.stream_begin(1, T_max=801, chunk=1)
x = 0.3 * torch.randn(1, 1, 161, 16, 2, device=dev)
st.step(x)
b = st.bound
print("chains:", [(f, c) for f, c, *_ in b.chains], "exec ops", b.n_exec, "of", len(b.prog.ops))
lib = _lib.load()
for k, op in enumerate(b.prog.ops):
    if op.kind == prg.OP_CONV and op.korder == prg.KORDER_FRAG:
        d = (_lib.ConvDesc * 1)(b.ops[k].conv)
        codes = (C.c_int * 1)(); lds = C.c_int(); bf = C.c_int()
        rc = lib.eab_conv_st_chain_plan(d, 1, codes, C.byref(lds), C.byref(bf))
        print(k, op.name, "N", op.N, "K", op.Kpad, "bm", op.bm, "No", op.No, "xf", op.xf_mode, "epi", op.epi, "rc", rc, "code", codes[0], "lds", lds.value)
