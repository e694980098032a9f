"""Generate tests/golden/*.npz|json by RUNNING THE REFERENCE in the authoring
container.  Not run by the test-suite (the reference does not exist on the GPU
box); committed so the fixtures can be regenerated and audited.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

* Network fixtures import the reference's own ``EaBNet`` class
  (/root/reference/EaBNet.py), load parameters from paramgen.py (key-seeded,
  reference-free) with ``strict=True`` and record outputs + hook taps.
* ``train_distributed.py`` cannot be imported here (torchaudio, tensorboard,
  ... are absent: SURVEY §8c), so the STFT fixtures call ``torch.stft`` with
  the arguments ``prepare_data`` passes (train_distributed.py:83-92) and apply
  the same norm/atan2/cos/sin compression.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

from EaBNet import EaBNet as RefEaBNet, com_mag_mse_loss as ref_loss  # noqa: E402  (reference)
import paramgen  # noqa: E402
from eabnet_amd.spec import NetConfig, param_specs  # noqa: E402

torch.set_num_threads(8)
N_FFT, HOP = 320, 160


VARIANTS = {          # constructor branches away from the default, T=20, B=2
    "bn": dict(norm_type="BN"),
    "unet": dict(is_u2=False),
    "cnn": dict(bf_type="cnn"),
    "miso": dict(topo_type="miso"),
    "add": dict(intra_connect="add"),
    "noncausal": dict(is_causal=False),
    "unet_bn_cnn_noncausal": dict(is_u2=False, norm_type="BN", bf_type="cnn", is_causal=False),
    "add_bn_miso": dict(intra_connect="add", norm_type="BN", topo_type="miso"),
    "cln": dict(norm_type="cLN"),       # needs the reference's NormSwitch constructor fixed: see _fixed_norm_switch()
    # the time extent of the gated convolutions (causal pad / chomp of k_t - 1 rows, EaBNet.py:447-452,477-482)
    "k1_33": dict(k1=(3, 3)),
    "k1_53_bn_add": dict(k1=(5, 3), norm_type="BN", intra_connect="add"),
    # one-frame gated kernels: no pad / chomp module, ".conv.weight" keys (EaBNet.py:452-454,482-484)
    "k1_13": dict(k1=(1, 3)),
}


def _fixed_norm_switch():
    """The reference cannot construct norm_type="cLN": NormSwitch passes the STRING dim_size to
    CumulativeLayerNorm{1,2}d as num_features (EaBNet.py:689,691).  For the cLN fixtures the two calls are given the
    channel count ``c`` instead -- nothing else of the reference changes (its own CumulativeLayerNorm classes run)."""
    import EaBNet as R

    class FixedNormSwitch(R.NormSwitch):
        def __init__(self, norm_type, dim_size, c):
            if norm_type != "cLN":
                super().__init__(norm_type, dim_size, c)
                return
            torch.nn.Module.__init__(self)
            self.norm_type, self.dim_size, self.c = norm_type, dim_size, c
            self.norm = (R.CumulativeLayerNorm1d if dim_size == "1D" else R.CumulativeLayerNorm2d)(c, affine=True)
    R.NormSwitch = FixedNormSwitch


def ref_model(M: int, seed: int, **kw):
    if kw.get("norm_type") == "cLN":
        _fixed_norm_switch()
    specs = param_specs(NetConfig(M=M, **kw))
    net = RefEaBNet(M=M, **kw).eval()
    sd = net.state_dict()
    assert list(sd.keys()) == list(specs.keys()), "key order/name mismatch vs reference"
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(specs[k].shape), (k, v.shape, specs[k].shape)
    params = paramgen.make_params(specs, seed)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    return net, specs


def ref_prepare(x: torch.Tensor, target: torch.Tensor):
    """torch.stft with prepare_data's arguments + its compression."""
    B, M, L = x.shape
    win = torch.hann_window(N_FFT)
    ns = torch.stft(x.contiguous().view(B * M, -1), N_FFT, HOP, N_FFT, win, return_complex=False)
    ts = torch.stft(target.squeeze(1), N_FFT, HOP, N_FFT, win, return_complex=False)
    _, Fq, T, _ = ns.shape
    ns = ns.view(B, M, Fq, T, -1).permute(0, 3, 2, 1, 4)
    ts = ts.permute(0, 3, 2, 1)
    nmag, nph = torch.norm(ns, dim=-1) ** 0.5, torch.atan2(ns[..., -1], ns[..., 0])
    tmag, tph = torch.norm(ts, dim=1) ** 0.5, torch.atan2(ts[:, -1, ...], ts[:, 0, ...])
    ns = torch.stack((nmag * torch.cos(nph), nmag * torch.sin(nph)), dim=-1)
    ts = torch.stack((tmag * torch.cos(tph), tmag * torch.sin(tph)), dim=1)
    return ns.contiguous(), ts.contiguous()


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"{name}: {os.path.getsize(path) / 1e3:.0f} kB")


def main():
    # -- state-dict inventory ------------------------------------------------
    for M in (8, 9):
        sd = RefEaBNet(M=M).state_dict()
        with open(os.path.join(HERE, f"keys_M{M}.json"), "w") as f:
            json.dump([[k, list(v.shape)] for k, v in sd.items()], f)

    # -- STFT front end --------------------------------------------------------
    for (B, M, L, seed) in ((1, 2, 1600, 0), (2, 8, 4000, 1), (1, 3, 2085, 2)):
        x = torch.from_numpy(paramgen.make_wave(B, M, L, seed))
        tgt = x[:, :1].clone()
        with torch.no_grad():
            ns, ts = ref_prepare(x, tgt)
        save(f"stft_B{B}_M{M}_L{L}.npz", noisy=ns.numpy(), target=ts.numpy(), seed=seed)
    # zero-signal bins must map to exactly 0 (atan2(0,0)=0, mag 0)
    x = torch.from_numpy(paramgen.make_wave(1, 2, 1600, 3)); x[:, 1] = 0.0
    with torch.no_grad():
        ns, _ = ref_prepare(x, x[:, :1])
    save("stft_zero_mic.npz", noisy=ns.numpy(), seed=3)

    # -- network, with taps ------------------------------------------------------
    net8, _ = ref_model(8, seed=100)
    taps = {}

    def hook(name):
        def fn(_m, _i, o):
            taps[name] = (o[0] if isinstance(o, tuple) else o).detach().numpy()
        return fn
    hs = []
    for i in range(4):
        hs.append(net8.en.meta_unet_list[i].register_forward_hook(hook(f"en.{i}")))
        hs.append(net8.en.meta_unet_list[i].in_conv.register_forward_hook(hook(f"en.meta_unet_list.{i}.in_conv")))
        hs.append(net8.de.meta_unet_list[i].register_forward_hook(hook(f"de.{i}")))
    hs.append(net8.en.last_conv.register_forward_hook(hook("en.4")))
    hs.append(net8.de.last_conv.register_forward_hook(hook("de.4")))
    hs.append(net8.stcns[0].tcm_list[0].register_forward_hook(hook("stcns.0.0")))
    hs.append(net8.bf_map.register_forward_hook(hook("bf_w")))
    hs.append(net8.bf_map.rnn1.register_forward_hook(hook("rnn1")))
    hs.append(net8.bf_map.rnn2.register_forward_hook(hook("rnn2")))
    x = torch.from_numpy(paramgen.make_spec_input(1, 12, 161, 8, seed=7))
    with torch.no_grad():
        y = net8(x)
    for h in hs:
        h.remove()
    save("e2e_M8_B1_T12_taps.npz", out=y.numpy(), param_seed=100, input_seed=7,
         **{"tap/" + k: v for k, v in taps.items()})

    # -- network, batch 2, ragged loss --------------------------------------------
    x = torch.from_numpy(paramgen.make_spec_input(2, 20, 161, 8, seed=8))
    with torch.no_grad():
        y = net8(x)
        label = torch.from_numpy(paramgen.make_spec_input(2, 20, 161, 1, seed=9)[..., 0, :]).permute(0, 3, 1, 2).contiguous()
        loss_full = ref_loss(y, label, [20, 20])
        # the reference pads masks to the longest entry, so keep one entry at T
        loss_ragged = ref_loss(y, label, [20, 13])
    save("e2e_M8_B2_T20.npz", out=y.numpy(), param_seed=100, input_seed=8, label_seed=9,
         loss_full=loss_full.numpy(), loss_ragged=loss_ragged.numpy())

    # -- 4-D input (B,T,F,2) == single mic (EaBNet.py:93-94) and odd mic count -------
    net1, _ = ref_model(1, seed=101)
    x = torch.from_numpy(paramgen.make_spec_input(1, 10, 161, 1, seed=10))
    with torch.no_grad():
        y5 = net1(x)
        y4 = net1(x[..., 0, :])
    assert torch.equal(y4, y5)
    save("e2e_M1_B1_T10.npz", out=y5.numpy(), param_seed=101, input_seed=10)
    net9, _ = ref_model(9, seed=102)
    x = torch.from_numpy(paramgen.make_spec_input(1, 10, 161, 9, seed=11))
    with torch.no_grad():
        y = net9(x)
    save("e2e_M9_B1_T10.npz", out=y.numpy(), param_seed=102, input_seed=11)

    # -- full C1 size: wave -> prepare_data -> EaBNet (4 s, 8 mics) -----------------
    wav = torch.from_numpy(paramgen.make_wave(1, 8, 64000, seed=12))
    with torch.no_grad():
        ns, ts = ref_prepare(wav, wav[:, :1])
        y = net8(ns)
    probes_t = [0, 1, 200, 400]
    save("c1_M8_T401.npz", out=y.numpy(), param_seed=100, wave_seed=12,
         stft_probe_t=np.array(probes_t), stft_probe=ns[:, probes_t].numpy(),
         stft_l2=np.float64(torch.linalg.vector_norm(ns.double()).item()),
         target_l2=np.float64(torch.linalg.vector_norm(ts.double()).item()))


def main_variants(only=None):
    """One small end-to-end fixture per non-default constructor branch (eval mode)."""
    path = os.path.join(HERE, "keys_variants.json")
    inventory = json.load(open(path)) if (only and os.path.exists(path)) else {}
    for i, (name, kw) in enumerate(VARIANTS.items()):
        if only and name != only:
            continue
        M = 4
        net, specs = ref_model(M, seed=200 + i, **kw)
        inventory[name] = dict(kwargs=kw, M=M, keys=[[k, list(v.shape)] for k, v in net.state_dict().items()])
        x = torch.from_numpy(paramgen.make_spec_input(2, 20, 161, M, seed=300 + i))
        y = net(x)
        save(f"var_{name}.npz", out=y.numpy(), param_seed=200 + i, input_seed=300 + i, M=M)
    with open(path, "w") as f:
        json.dump(inventory, f)


def main_bn_train():
    """norm_type="BN" with the module in TRAIN mode (the mode the reference's trainer runs in): batch statistics in the
    forward, and the running buffers after that one forward (momentum 0.1, unbiased variance)."""
    M, kw = 4, dict(norm_type="BN", p=2, q=2)
    net, specs = ref_model(M, seed=260, **kw)
    net.train()
    x = torch.from_numpy(paramgen.make_spec_input(3, 20, 161, M, seed=360))
    y = net(x)
    sd = net.state_dict()
    arrs = {k: v.numpy() for k, v in sd.items() if k.endswith(("running_mean", "running_var", "num_batches_tracked"))}
    save("bn_train.npz", out=y.numpy(), param_seed=260, input_seed=360, M=M, p=2, q=2, B=3, T=20, **arrs)


def main_istft():
    """Back end: the reference's call (enhance.py:59-62) on seeded (B,2,T,F) estimates; the inputs
    carry non-zero imaginary DC/Nyquist bins, which torch.istft ignores."""
    for (B, T, seed) in ((1, 2, 20), (2, 9, 21), (3, 40, 22)):
        esti = torch.from_numpy(paramgen.make_spec_input(B, T, 161, 1, seed)[..., 0, :]).permute(0, 3, 1, 2).contiguous()
        x = esti.permute(0, 3, 2, 1)
        wav = torch.istft(torch.view_as_complex(x.contiguous()), N_FFT, HOP, N_FFT, torch.hann_window(N_FFT))
        assert wav.shape == (B, HOP * (T - 1))
        save(f"istft_B{B}_T{T}.npz", wav=wav.numpy(), seed=seed)


GAG_BASE = dict(cin=2, k1=(2, 3), k2=(1, 3), c=64, kd1=3, cd1=64, d_feat=256, p=2, q=3, dilas=(1, 2, 5, 9), fft_num=320,
                is_u2=True, is_causal=True, is_squeezed=False, acti_type="sigmoid", intra_connect="cat", norm_type="IN")
GAG_VARIANTS = {
    "default": dict(),
    "bn_squeezed_tanh": dict(norm_type="BN", is_squeezed=True, acti_type="tanh", p=1, q=2),
    "unet_add_noncausal_relu": dict(is_u2=False, intra_connect="add", is_causal=False, acti_type="relu", p=1, q=2,
                                    dilas=(1, 2)),
    "k1_33": dict(k1=(3, 3), p=1, q=2, dilas=(1, 2)),    # gated convolutions three frames long (GaGNet.py's GateConv2d, as EaBNet's)
    "k1_13": dict(k1=(1, 3), p=1, q=2, dilas=(1, 2)),    # ... and one frame long (".conv.weight" keys)
}


def main_gagnet(only=None):
    """Post-filter: the reference's GaGNet on CPU (its factory make_gag_net calls
    .cuda(), the class itself does not) and the two-stage composition of EaBNetWithPostNet.forward
    (EaBNet.py:138-148) spelled out with the reference's two classes."""
    from GaGNet import GaGNet as RefGaGNet, stagewise_com_mag_mse_loss as ref_stage_loss
    from eabnet_amd.spec import GagConfig, gag_param_specs
    inv_path = os.path.join(HERE, "keys_gagnet.json")
    inventory = json.load(open(inv_path)) if (only and os.path.exists(inv_path)) else {}
    for i, (name, kw) in enumerate(GAG_VARIANTS.items()):
        if only and name != only:
            continue
        full = dict(GAG_BASE); full.update(kw)
        net = RefGaGNet(**{**full, "dilas": list(full["dilas"])}).eval()
        specs = gag_param_specs(GagConfig(**full))
        sd = net.state_dict()
        assert list(sd.keys()) == list(specs.keys())
        net.load_state_dict({k: torch.from_numpy(v) for k, v in paramgen.make_params(specs, 500 + i).items()}, strict=True)
        inventory[name] = dict(kwargs={k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()},
                               keys=[[k, list(v.shape)] for k, v in sd.items()])
        B, T = 2, 14
        inpt = torch.from_numpy(paramgen.make_spec_input(B, T, 161, 1, 600 + i)[..., 0, :]).permute(0, 3, 1, 2).contiguous()
        pre = torch.from_numpy(paramgen.make_spec_input(B, T, 161, 1, 700 + i)[..., 0, :]).permute(0, 3, 1, 2).contiguous()
        outs = net(inpt, pre)
        extra = {}
        if name == "default":
            label = torch.from_numpy(paramgen.make_spec_input(B, T, 161, 1, 800)[..., 0, :]).permute(0, 3, 2, 1).contiguous()
            extra["stage_loss"] = ref_stage_loss(outs, label, [T, 9]).numpy()
        save(f"gag_{name}.npz", param_seed=500 + i, inpt_seed=600 + i, pre_seed=700 + i,
             **{f"out{j}": o.numpy() for j, o in enumerate(outs)}, **extra)
    with open(os.path.join(HERE, "keys_gagnet.json"), "w") as f:
        json.dump(inventory, f)
    if only:
        return

    # two-stage wrapper: reference EaBNet -> reference GaGNet, composed as EaBNetWithPostNet.forward does
    M, B, T, ref_mic = 4, 1, 12, 1
    eab, _ = ref_model(M, seed=520)
    gag = RefGaGNet(**{**GAG_BASE, "dilas": list(GAG_BASE["dilas"])}).eval()
    gag.load_state_dict({k: torch.from_numpy(v) for k, v in
                         paramgen.make_params(gag_param_specs(GagConfig(**GAG_BASE)), 521).items()}, strict=True)
    noisy = torch.from_numpy(paramgen.make_spec_input(B, T, 161, M, 522))
    esti0 = eab(noisy)
    lst = gag(noisy[..., ref_mic, :].permute(0, 3, 1, 2), esti0)
    save("postnet_M4_T12.npz", esti0=esti0.numpy(), esti=lst[-1].permute(0, 1, 3, 2).numpy(), eab_seed=520, gag_seed=521,
         input_seed=522, ref_mic=ref_mic, **{f"stage{j}": o.numpy() for j, o in enumerate(lst)})


def _fill(module, seed, recipe):
    """deterministic, non-default values for every parameter of a stand-alone reference block; `recipe`
    receives (key, shape, kind, fan_in, seed) so that the test can regenerate them without the reference"""
    sd = {}
    for k, v in module.state_dict().items():
        kind = ("prelu" if v.ndim == 1 and k.endswith(("0.weight", "2.weight")) and "norm" not in k and "conv" not in k.split(".")[-2:]
                else "norm_w" if "norm.weight" in k else "norm_b" if "norm.bias" in k
                else "bias" if k.endswith("bias") or "bias_" in k else "conv_w")
        fan = int(np.prod(v.shape[1:])) if v.ndim > 1 else int(v.shape[0])
        sd[k] = torch.from_numpy(paramgen.make_param("block/" + k, tuple(v.shape), kind, fan, seed))
        recipe.append([k, list(v.shape), kind, fan, seed])
    module.load_state_dict(sd, strict=True)


def main_blocks():
    """Per-block input/output pairs of the reference's own classes: each block alone, with
    its parameters stored next to the result (they are small), T = 10, B = 1."""
    import EaBNet as R
    rng = np.random.default_rng(77)
    out, recipes = {}, {}

    def run(name, module, x, seed):
        module.eval()
        recipes[name] = []
        _fill(module, seed, recipes[name])
        y = module(torch.from_numpy(x))
        out[f"{name}/x"] = x
        out[f"{name}/y"] = (y[0] if isinstance(y, tuple) else y).numpy()

    g = lambda *shape: (0.5 * rng.standard_normal(shape)).astype(np.float32)  # noqa: E731
    run("gateconv_2x5", R.GateConv2d(16, 64, (2, 5), (1, 2)), g(1, 16, 10, 161), 1)
    run("gateconv_2x3", R.GateConv2d(64, 64, (2, 3), (1, 2)), g(1, 64, 10, 39), 2)
    run("gatedeconv_2x3", R.GateConvTranspose2d(128, 64, (2, 3), (1, 2)), g(1, 128, 10, 19), 3)
    run("gatedeconv_2x5", R.GateConvTranspose2d(128, 64, (2, 5), (1, 2)), g(1, 128, 10, 79), 4)
    run("conv2dunit", R.Conv2dunit((1, 3), 64, "IN"), g(1, 64, 10, 39), 5)
    run("deconv2dunit_add", R.Deconv2dunit((1, 3), 64, "add", "IN"), g(1, 64, 10, 9), 6)
    run("deconv2dunit_cat", R.Deconv2dunit((1, 3), 64, "cat", "IN"), g(1, 128, 10, 9), 7)
    run("unet_module_enc_s3", R.En_unet_module(64, 64, (2, 3), (1, 3), "cat", "IN", 3, False), g(1, 64, 10, 79), 8)
    run("unet_module_dec_s2", R.En_unet_module(128, 64, (2, 3), (1, 3), "cat", "IN", 2, True), g(1, 128, 10, 9), 9)
    run("stcm_d1", R.SqueezedTCM(5, 64, 1, 256, True, "IN"), g(1, 256, 40), 10)
    run("stcm_d32", R.SqueezedTCM(5, 64, 32, 256, True, "IN"), g(1, 256, 40), 11)
    run("lstm_bf", R.LSTM_BF(64, 8), g(1, 64, 10, 21), 12)
    save("blocks.npz", **out)
    with open(os.path.join(HERE, "blocks_params.json"), "w") as f:
        json.dump(recipes, f)


def main_cln():
    """The reference's own CumulativeLayerNorm1d / 2d on random inputs with random gain / bias (pins the restatement of
    the norm itself), and the cLN variant of the network (constructor fix above)."""
    import EaBNet as R
    rng = np.random.default_rng(91)
    out = {}
    for name, cls, shape in (("1d", R.CumulativeLayerNorm1d, (2, 64, 37)), ("2d", R.CumulativeLayerNorm2d, (2, 64, 11, 19))):
        m = cls(shape[1], affine=True)
        gshape = tuple(m.gain.shape)
        g, b = (1.0 + 0.3 * rng.standard_normal(gshape)).astype(np.float32), (0.2 * rng.standard_normal(gshape)).astype(np.float32)
        m.gain.data, m.bias.data = torch.from_numpy(g), torch.from_numpy(b)
        x = (0.7 * rng.standard_normal(shape) + 0.2).astype(np.float32)
        out[f"{name}/x"], out[f"{name}/gain"], out[f"{name}/bias"] = x, g, b
        out[f"{name}/y"] = m(torch.from_numpy(x)).numpy()
    save("cln_classes.npz", **out)


def main_losses():
    """eabnet_with_postnet_loss (EaBNet.py:642-650, called at train_distributed.py:225) on the two-stage
    fixture's reference outputs."""
    from EaBNet import eabnet_with_postnet_loss as ref_two_stage_loss
    g = np.load(os.path.join(HERE, "postnet_M4_T12.npz"))
    output = {"esti0_stft": torch.from_numpy(g["esti0"]),
              "esti1_stft_list": [torch.from_numpy(g[f"stage{j}"]) for j in range(3)]}
    label = torch.from_numpy(paramgen.make_spec_input(1, 12, 161, 1, 810)[..., 0, :]).permute(0, 3, 1, 2).contiguous()
    arrs = {"label_seed": 810}
    for tag, frames in (("full", [12]),):        # (the reference's pad_sequence masks need one full-length utterance per batch)
        l = ref_two_stage_loss(output, label, frames)
        for k, v in l.items():
            arrs[f"{tag}/{k}"] = v.numpy()
    save("loss_postnet.npz", **arrs)


if __name__ == "__main__":
    with torch.no_grad():
        if sys.argv[1:] == ["losses"]:
            main_losses()
        elif sys.argv[1:] == ["cln"]:
            main_cln()
            main_variants(only="cln")
        elif sys.argv[1:] == ["blocks"]:
            main_blocks()
        elif sys.argv[1:] == ["gagnet"]:
            main_gagnet()
        elif sys.argv[1:] == ["variants"]:
            main_variants()
        elif len(sys.argv) == 3 and sys.argv[1] == "variant":
            main_variants(only=sys.argv[2])
        elif len(sys.argv) == 3 and sys.argv[1] == "gag_variant":
            main_gagnet(only=sys.argv[2])
        elif sys.argv[1:] == ["istft"]:
            main_istft()
        elif sys.argv[1:] == ["bn_train"]:
            main_bn_train()
        else:
            main()
            main_variants()
            main_istft()
            main_gagnet()
            main_blocks()
            main_losses()
            main_bn_train()
