"""The C-ABI library loads on a machine without a GPU and exports every symbol
include/eabnet_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "eabnet_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(eab_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from eabnet_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/eabnet_hip.h but not exported"
    assert sorted(_lib.EXPORTS) == names, "ctypes binding and header disagree on the entry points"


def test_binding_handshake_and_error_strings():
    from eabnet_amd import _lib
    lib = _lib.load()                      # checks ABI version + struct sizes
    assert lib.eab_abi_version() == _lib.ABI_VERSION
    assert b"invalid argument" in lib.eab_error_string(1)
    assert lib.eab_conv_tiles(401, 79, 128) == (401 * 79 + 127) // 128
    assert lib.eab_conv_tiles(401, 79, 100) == -1


def test_argument_validation_happens_before_any_launch():
    """Null pointers / bad shapes are rejected on the host (EAB_EINVAL) -- safe to
    call without a GPU because nothing is launched."""
    from eabnet_amd import _lib
    lib = _lib.load()
    assert lib.eab_filter_sum_f32(None, None, None, 1, 1, 1, 1, None) == 1
    assert lib.eab_stft_compress_f32(None, None, None, None, 1, 1, 1000, 320, 160, 0, None) == 1
    d = _lib.ConvDesc()
    assert lib.eab_conv_f32(ctypes.byref(d), None) == 1
    assert lib.eab_run_program(None, 0, None) == 1
    with pytest.raises(_lib.EabError):
        _lib.check(1, "x")


def test_product_path_refuses_cpu_tensors():
    import torch
    import eabnet_amd
    net = eabnet_amd.EaBNet(M=2)
    with torch.no_grad(), pytest.raises(eabnet_amd._lib.EabError):
        net(torch.zeros(1, 4, 161, 2, 2))
    with pytest.raises(eabnet_amd._lib.EabError):
        eabnet_amd.filter_and_sum(torch.zeros(1, 1, 1, 1, 2), torch.zeros(1, 1, 1, 1, 2))


def test_training_forward_matches_oracle_on_cpu():
    """The differentiable path (training) against the oracle; runs on CPU because it is plain PyTorch."""
    import torch
    import paramgen
    import eabnet_amd
    from eabnet_amd.spec import NetConfig, param_specs
    from oracle import eabnet_oracle as orc
    cfg = NetConfig(M=3, p=2, q=2)
    P = {k: torch.from_numpy(v) for k, v in paramgen.make_params(param_specs(cfg), 160).items()}
    net = eabnet_amd.EaBNet(M=3, p=2, q=2)
    net.load_state_dict(P, strict=True)
    x = torch.from_numpy(paramgen.make_spec_input(2, 9, 161, 3, 161))
    y = net(x)                                           # grad enabled -> autograd path
    assert y.requires_grad
    with torch.no_grad():
        ref = orc.eabnet_forward(P, x, p=2, q=2)
    assert float((y.detach() - ref).abs().max() / ref.abs().max()) < 1e-5
    y.square().mean().backward()
    assert all(p.grad is not None for p in net.parameters())


def test_state_dict_round_trip_with_reference_keys():
    import json
    import torch
    import eabnet_amd
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "keys_M8.json")))
    net = eabnet_amd.EaBNet(M=8)
    sd = net.state_dict()
    assert list(sd.keys()) == [k for k, _ in ref]
    assert all(tuple(sd[k].shape) == tuple(s) for k, s in ref)
    assert eabnet_amd.numParams(net) == 2_835_920
    other = eabnet_amd.EaBNet(M=8)
    other.load_state_dict({k: torch.randn_like(v) for k, v in sd.items()}, strict=True)
    # wrapper prefix used by EaBNetWithPostNet (reference EaBNet.py:130)
    class Wrap(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.eabnet = eabnet_amd.EaBNet(M=8)
    assert all(k.startswith("eabnet.") for k in Wrap().state_dict())
    # reference default is M=9
    assert eabnet_amd.EaBNet().M == 9


def _variants():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "keys_variants.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("name", sorted(_variants()))
def test_constructor_variants_keep_reference_keys_and_train(name):
    """Every constructor branch: the module's state dict (parameters AND BatchNorm buffers, in the
    reference's order) and the differentiable path against the reference's output."""
    import numpy as np
    import torch
    import paramgen
    import eabnet_amd
    from eabnet_amd.spec import NetConfig, param_specs
    e = _variants()[name]
    g = np.load(os.path.join(ROOT, "tests", "golden", f"var_{name}.npz"))
    net = eabnet_amd.EaBNet(M=e["M"], **e["kwargs"])
    assert [[k, list(v.shape)] for k, v in net.state_dict().items()] == e["keys"]
    specs = param_specs(NetConfig(M=e["M"], **e["kwargs"]))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in paramgen.make_params(specs, int(g["param_seed"])).items()},
                        strict=True)
    net.eval()
    x = torch.from_numpy(paramgen.make_spec_input(2, 20, 161, e["M"], int(g["input_seed"])))
    y = net(x)
    ref = torch.from_numpy(g["out"])
    assert y.shape == ref.shape and y.requires_grad
    assert float((y.detach() - ref).abs().max() / ref.abs().max()) < 1e-5
    y.square().mean().backward()
    assert all(p.grad is not None for p in net.parameters())


def test_batchnorm_training_mode_updates_running_statistics():
    """norm_type='BN' in train mode follows nn.BatchNorm: batch statistics, momentum-0.1 update of
    the buffers, step counter; this holds under no_grad too (never the eval-mode HIP tables)."""
    import torch
    import eabnet_amd
    net = eabnet_amd.EaBNet(M=2, p=1, q=1, norm_type="BN").train()
    key = "en.meta_unet_list.0.in_conv.1.norm"
    before = net.state_dict()[f"{key}.running_mean"].clone()
    with torch.no_grad():
        net(torch.randn(2, 6, 161, 2, 2))
    sd = net.state_dict()
    assert int(sd[f"{key}.num_batches_tracked"]) == 1
    assert not torch.equal(sd[f"{key}.running_mean"], before)
    assert eabnet_amd.numParams(net) == sum(p.numel() for p in net.parameters())
