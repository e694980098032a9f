"""per-parameter gradient error of one constructor variant on the HIP training programs vs fp64 oracle autograd (debug aid):
python tools/diag_variant_grads.py unet [B] [T]"""
import json, os, sys
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [root, os.path.join(root, "tests"), os.path.join(root, "tests", "golden")]
import paramgen, eabnet_amd
from util import torch_params
from eabnet_amd.spec import NetConfig, param_specs
from oracle import eabnet_oracle as orc
name = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 3
T = int(sys.argv[3]) if len(sys.argv) > 3 else 21
e = json.load(open(os.path.join(root, "tests/golden/keys_variants.json")))[name]
kw, M = dict(e["kwargs"], p=2, q=2), e["M"]
P = torch_params(M, 980, **kw)
specs = param_specs(NetConfig(M=M, **kw))
for k, sp in specs.items():
    if sp.kind == "prelu":
        P[k] = torch.ones_like(P[k])
dev = torch.device("cuda:0")
net = eabnet_amd.EaBNet(M=M, **kw); net.load_state_dict(P, strict=True); net = net.to(dev).train()
x = torch.from_numpy(paramgen.make_spec_input(B, T, 161, M, 981))
y = net(x.to(dev))
label = torch.from_numpy(np.random.default_rng(982).standard_normal(tuple(y.shape)).astype(np.float32))
((y - label.to(dev)) ** 2).mean().backward()
is_param = {k for k, sp in specs.items() if not sp.kind.startswith("bn_")}
Pd = {k: (v.double().requires_grad_(True) if k in is_param else v.double()) for k, v in P.items()}
y_ref = orc.eabnet_forward(Pd, x.double(), bn_train=kw.get("norm_type") == "BN", **kw)
((y_ref - label.double()) ** 2).mean().backward()
print("fwd max rel", float((y.detach().cpu().double() - y_ref.detach()).abs().max() / y_ref.detach().abs().max()))
for k in is_param:
    g, r = net.get_parameter(k).grad.cpu().double(), Pd[k].grad
    print(f"{float((g - r).abs().max() / (r.abs().max() + 1e-30)):9.2e}  {float(r.abs().max()):9.2e}  {k}")
