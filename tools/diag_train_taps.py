"""Gradient of every tapped activation (HIP training programs vs fp64 autograd of the oracle) for one configuration:
python tools/diag_train_taps.py M B T p q"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import eabnet_amd
from eabnet_amd.spec import NetConfig, param_specs
from oracle import eabnet_oracle as orc
from util import torch_params
import paramgen

M, B, T, p, q = (int(a) for a in sys.argv[1:6])
dev = torch.device("cuda:0")
kw = dict(p=p, q=q)
P = torch_params(M, 910 + M, **kw)
for k, sp in param_specs(NetConfig(M=M, **kw)).items():
    if sp.kind == "prelu":
        P[k] = torch.ones_like(P[k])
net = eabnet_amd.EaBNet(M=M, **kw)
net.load_state_dict(P, strict=True)
net = net.to(dev).train()
x = torch.from_numpy(paramgen.make_spec_input(B, T, 161, M, 920))
label = torch.from_numpy(paramgen.make_spec_input(B, T, 161, 1, 921)[..., 0, :]).permute(0, 3, 1, 2).contiguous()
y = net(x.to(dev))
loss = eabnet_amd.com_mag_mse_loss(y, label.to(dev), [T] * B)
loss.backward()
Pd = {k: v.double().requires_grad_(True) for k, v in P.items()}
taps = {}
yo = orc.eabnet_forward(Pd, x.double(), taps=taps, **kw)
taps = {k: v for k, v in taps.items() if torch.is_tensor(v) and v.requires_grad}
for v in taps.values():
    v.retain_grad()
orc.com_mag_mse_loss(yo, label.double(), [T] * B).backward()
bound = next(iter(net._train_bound.values()))
for name, (r, Fv, Cv) in bound.prog.grad_taps.items():
    if name not in taps or taps[name].grad is None:
        continue
    g = bound.acts[r.off:r.off + B * T * Fv * Cv].view(B, T, Fv, Cv).cpu().double()
    w = taps[name].grad
    if name == "bf_w":
        w = w.reshape(B, T, Fv, -1)
    elif name.startswith("bf_map."):
        w = w.permute(0, 2, 1, 3)                   # (B,F,T,C) -> (B,T,F,C)
    else:
        w = w.permute(0, 2, 3, 1)
    gg = g[..., :w.shape[-1]]
    print(f"{name:14s} l2-rel {float((gg - w).norm() / w.norm()):.2e}  max-abs {float((gg - w).abs().max()):.2e} of {float(w.abs().max()):.2e}")

# the head again in fp64 with every intermediate tapped (LayerNorm -> LSTM x 2 -> Linear + ReLU -> Linear -> filter-and-sum)
import torch.nn.functional as F
e = taps["de.4"].detach().clone().requires_grad_(True)
Pd2 = {k: v.detach() for k, v in Pd.items()}
Bq, Cq, Tq, Fq = e.shape
x_ln = F.layer_norm(e.permute(0, 3, 2, 1).contiguous(), (Cq,), Pd2["bf_map.norm.weight"], Pd2["bf_map.norm.bias"], 1e-5)
x_ln.retain_grad()
hs = []
h = x_ln.view(Bq * Fq, Tq, Cq)
for nm in ("rnn1", "rnn2"):
    h = orc.lstm_layer(h, Pd2[f"bf_map.{nm}.weight_ih_l0"], Pd2[f"bf_map.{nm}.weight_hh_l0"], Pd2[f"bf_map.{nm}.bias_ih_l0"],
                       Pd2[f"bf_map.{nm}.bias_hh_l0"], False)
    h.retain_grad()
    hs.append(h)
hh = h.view(Bq, Fq, Tq, -1).transpose(1, 2).contiguous()
y1 = F.relu(F.linear(hh, Pd2["bf_map.w_dnn.0.weight"], Pd2["bf_map.w_dnn.0.bias"]))
y1.retain_grad()
w = F.linear(y1, Pd2["bf_map.w_dnn.2.weight"], Pd2["bf_map.w_dnn.2.bias"]).view(Bq, Tq, Fq, M, 2)
out = orc.filter_and_sum(w, x.double())
orc.com_mag_mse_loss(out, label.double(), [T] * B).backward()
refs = {"bf_map.ln": x_ln.grad.permute(0, 2, 1, 3), "bf_map.rnn1": hs[0].grad.view(Bq, Fq, Tq, -1).permute(0, 2, 1, 3),
        "bf_map.rnn2": hs[1].grad.view(Bq, Fq, Tq, -1).permute(0, 2, 1, 3), "bf_map.y1": y1.grad}
for name, w_ in refs.items():
    if name not in bound.prog.grad_taps:
        print(name, "not tapped")
        continue
    r, Fv, Cv = bound.prog.grad_taps[name]
    g = bound.acts[r.off:r.off + B * T * Fv * Cv].view(B, T, Fv, Cv).cpu().double()
    print(f"{name:14s} l2-rel {float((g - w_).norm() / w_.norm()):.2e}  max-abs {float((g - w_).abs().max()):.2e} of {float(w_.abs().max()):.2e}")

# the reference arithmetic's own fp32 floor for this instance (oracle in fp32 vs fp64): a ReLU / PReLU whose sign flips under
# rounding shows up here as well
P32 = {k: v.float().requires_grad_(True) for k, v in P.items()}
orc.com_mag_mse_loss(orc.eabnet_forward(P32, x.float(), **kw), label.float(), [T] * B).backward()
num = sum(float(((P32[k].grad.double() - Pd[k].grad) ** 2).sum()) for k in Pd)
den = sum(float((Pd[k].grad ** 2).sum()) for k in Pd)
got = {k: net.get_parameter(k).grad.cpu().double() for k in Pd}
num2 = sum(float(((got[k] - Pd[k].grad) ** 2).sum()) for k in Pd)
print(f"parameter gradients, global l2-rel vs fp64: oracle in fp32 {np.sqrt(num / den):.2e}, HIP programs {np.sqrt(num2 / den):.2e}")

# ReLU mask of the head's first Linear: HIP forward vs fp64
op = next(o for o in bound.prog.fwd if getattr(o, "name", "") == "bf_map.w_dnn.0")
y1_hip = bound.acts[op.dst.off:op.dst.off + B * T * 161 * 64].view(B, T, 161, 64).cpu().double()
y1_ref = y1.detach()
flips = ((y1_hip > 0) != (y1_ref > 0))
print("ReLU units whose mask differs from fp64:", int(flips.sum()), "of", flips.numel(),
      "| largest fp64 activation among them:", float(y1_ref[flips].abs().max()) if flips.any() else 0.0,
      "| largest HIP activation among them:", float(y1_hip[flips].abs().max()) if flips.any() else 0.0,
      "| forward max-abs diff:", float((y1_hip - y1_ref).abs().max()))
