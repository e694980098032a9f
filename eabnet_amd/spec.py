"""Parameter inventory of the EaBNet hot path, as data.

The drop-in boundary requires that checkpoints written by the reference load
unchanged (``load_state_dict(strict=True)``, reference enhance.py:22), so every
tensor the reference registers must exist here under the same dotted key and
with the same shape.  This file lists them without mirroring the reference's
class tree: one flat, ordered table ``key -> ParamSpec`` generated from the
network hyper-parameters.  Both the ``nn.Module`` boundary (model.py) and the
weight packer for the HIP program (program.py) read this table.

Key layout follows the reference constructors (file:line into
/root/reference/EaBNet.py):
  * U2Net_Encoder      157-189   ``en.meta_unet_list.{i}``, ``en.last_conv``
  * U2Net_Decoder      241-271   ``de.meta_unet_list.{i}``, ``de.last_conv``
  * En_unet_module     331-370   ``in_conv`` / ``enco.{j}`` / ``deco.{j}``
  * GateConv2d         434-453   ``conv.1`` (index 1: index 0 is the causal pad)
  * GateConvTranspose2d 463-483  ``conv.0`` (index 1 is the chomp)
  * Conv2dunit 391-405, Deconv2dunit 410-428, NormSwitch 662-686
  * SqueezedTCM        532-571   ``stcns.{g}.tcm_list.{i}``
  * LSTM_BF            581-598   ``bf_map``
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import Tuple


@dataclass(frozen=True)
class NetConfig:
    """Hyper-parameters of EaBNet.__init__ (reference EaBNet.py:10-27)."""
    k1: Tuple[int, int] = (2, 3)
    k2: Tuple[int, int] = (1, 3)
    c: int = 64
    M: int = 9
    embed_dim: int = 64
    kd1: int = 5
    cd1: int = 64
    d_feat: int = 256
    p: int = 6
    q: int = 3
    is_causal: bool = True
    is_u2: bool = True
    bf_type: str = "lstm"
    topo_type: str = "mimo"
    intra_connect: str = "cat"
    norm_type: str = "IN"

    # The encoder's first and the decoder's last gated conv use a fixed
    # (2, 5) kernel and the bottleneck is fixed at 64 channels
    # (reference EaBNet.py:173-174, 250-251).
    k_beg: Tuple[int, int] = (2, 5)
    c_end: int = 64
    hid_node: int = 64

    def check_supported(self) -> None:
        """The HIP path covers the reference's default topology (SURVEY §2.1:
        the other branches are marked out of scope)."""
        bad = []
        if not self.is_u2:
            bad.append("is_u2=False")
        if self.bf_type != "lstm":
            bad.append(f"bf_type={self.bf_type!r}")
        if self.topo_type != "mimo":
            bad.append(f"topo_type={self.topo_type!r}")
        if self.intra_connect != "cat":
            bad.append(f"intra_connect={self.intra_connect!r}")
        if self.norm_type != "IN":
            bad.append(f"norm_type={self.norm_type!r}")
        if not self.is_causal:
            bad.append("is_causal=False")
        if tuple(self.k1) != (2, 3) or tuple(self.k2) != (1, 3):
            bad.append(f"k1={self.k1} k2={self.k2}")
        if self.c != 64 or self.embed_dim != 64 or self.cd1 != 64:
            bad.append("c/embed_dim/cd1 != 64")
        if self.d_feat != self.c_end * 4:
            bad.append("d_feat != 4*64")
        if self.kd1 < 1 or self.p < 1 or self.q < 1 or self.M < 1:
            bad.append("kd1/p/q/M < 1")
        if bad:
            raise NotImplementedError(
                "eabnet_amd implements the reference's default EaBNet topology "
                "on MI355X; unsupported option(s): " + ", ".join(bad))


@dataclass(frozen=True)
class ParamSpec:
    shape: Tuple[int, ...]
    kind: str       # conv_w | convT_w | bias | norm_w | norm_b | prelu | lstm | lin_w | ln_w | ln_b
    fan_in: int     # for the default initialiser


def _norm_prelu(tab, prefix_norm: str, prefix_act: str, c: int) -> None:
    tab[f"{prefix_norm}.norm.weight"] = ParamSpec((c,), "norm_w", c)
    tab[f"{prefix_norm}.norm.bias"] = ParamSpec((c,), "norm_b", c)
    tab[f"{prefix_act}.weight"] = ParamSpec((c,), "prelu", c)


def _gate_conv(tab, prefix: str, cin: int, cout: int, k, transposed: bool) -> None:
    kt, kf = k
    if transposed:
        # ConvTranspose2d weight is (cin, 2*cout, kt, kf); PyTorch's default
        # initialiser takes fan_in from dim 1.
        idx = 0
        tab[f"{prefix}.conv.{idx}.weight"] = ParamSpec((cin, 2 * cout, kt, kf), "convT_w", 2 * cout * kt * kf)
        tab[f"{prefix}.conv.{idx}.bias"] = ParamSpec((2 * cout,), "bias", 2 * cout * kt * kf)
    else:
        idx = 1 if kt > 1 else None
        name = f"{prefix}.conv.{idx}" if idx is not None else f"{prefix}.conv"
        tab[f"{name}.weight"] = ParamSpec((2 * cout, cin, kt, kf), "conv_w", cin * kt * kf)
        tab[f"{name}.bias"] = ParamSpec((2 * cout,), "bias", cin * kt * kf)


def _unet_module(tab, prefix: str, cin: int, cout: int, k1, k2, scale: int, is_deconv: bool) -> None:
    _gate_conv(tab, f"{prefix}.in_conv.0", cin, cout, k1, is_deconv)
    _norm_prelu(tab, f"{prefix}.in_conv.1", f"{prefix}.in_conv.2", cout)
    kt, kf = k2
    for j in range(scale):
        p = f"{prefix}.enco.{j}.conv"
        tab[f"{p}.0.weight"] = ParamSpec((cout, cout, kt, kf), "conv_w", cout * kt * kf)
        tab[f"{p}.0.bias"] = ParamSpec((cout,), "bias", cout * kt * kf)
        _norm_prelu(tab, f"{p}.1", f"{p}.2", cout)
    for j in range(scale):
        p = f"{prefix}.deco.{j}.deconv"
        cin_j = cout if j == 0 else 2 * cout          # first inner deconv has no skip
        tab[f"{p}.0.weight"] = ParamSpec((cin_j, cout, kt, kf), "convT_w", cout * kt * kf)
        tab[f"{p}.0.bias"] = ParamSpec((cout,), "bias", cout * kt * kf)
        _norm_prelu(tab, f"{p}.1", f"{p}.2", cout)


def param_specs(cfg: NetConfig) -> "OrderedDict[str, ParamSpec]":
    """Ordered ``key -> ParamSpec`` for the supported topology."""
    cfg.check_supported()
    tab: "OrderedDict[str, ParamSpec]" = OrderedDict()
    c, M = cfg.c, cfg.M

    # encoder: scales 4,3,2,1 then a gated conv down to F=4
    en_k = [cfg.k_beg, cfg.k1, cfg.k1, cfg.k1]
    en_cin = [2 * M, c, c, c]
    for i in range(4):
        _unet_module(tab, f"en.meta_unet_list.{i}", en_cin[i], c, en_k[i], cfg.k2, 4 - i, False)
    _gate_conv(tab, "en.last_conv.0", c, cfg.c_end, cfg.k1, False)
    _norm_prelu(tab, "en.last_conv.1", "en.last_conv.2", cfg.c_end)

    # decoder: scales 1,2,3,4 on cat(skip) inputs, then the (2,5) gated deconv
    for i in range(4):
        cin = 2 * cfg.c_end if i == 0 else 2 * c
        _unet_module(tab, f"de.meta_unet_list.{i}", cin, c, cfg.k1, cfg.k2, i + 1, True)
    _gate_conv(tab, "de.last_conv.0", 2 * c, cfg.embed_dim, cfg.k_beg, True)
    _norm_prelu(tab, "de.last_conv.1", "de.last_conv.2", cfg.embed_dim)

    # beamformer head
    E, H = cfg.embed_dim, cfg.hid_node
    for name, isz in (("rnn1", E), ("rnn2", H)):
        tab[f"bf_map.{name}.weight_ih_l0"] = ParamSpec((4 * H, isz), "lstm", H)
        tab[f"bf_map.{name}.weight_hh_l0"] = ParamSpec((4 * H, H), "lstm", H)
        tab[f"bf_map.{name}.bias_ih_l0"] = ParamSpec((4 * H,), "lstm", H)
        tab[f"bf_map.{name}.bias_hh_l0"] = ParamSpec((4 * H,), "lstm", H)
    tab["bf_map.w_dnn.0.weight"] = ParamSpec((H, H), "lin_w", H)
    tab["bf_map.w_dnn.0.bias"] = ParamSpec((H,), "bias", H)
    tab["bf_map.w_dnn.2.weight"] = ParamSpec((2 * M, H), "lin_w", H)
    tab["bf_map.w_dnn.2.bias"] = ParamSpec((2 * M,), "bias", H)
    tab["bf_map.norm.weight"] = ParamSpec((E,), "ln_w", E)
    tab["bf_map.norm.bias"] = ParamSpec((E,), "ln_b", E)

    # squeezed-TCN bottleneck
    D, cd, kd = cfg.d_feat, cfg.cd1, cfg.kd1
    for g in range(cfg.q):
        for i in range(cfg.p):
            p = f"stcns.{g}.tcm_list.{i}"
            tab[f"{p}.in_conv.weight"] = ParamSpec((cd, D, 1), "conv_w", D)
            for side in ("left_conv", "right_conv"):
                tab[f"{p}.{side}.0.weight"] = ParamSpec((cd,), "prelu", cd)
                tab[f"{p}.{side}.1.norm.weight"] = ParamSpec((cd,), "norm_w", cd)
                tab[f"{p}.{side}.1.norm.bias"] = ParamSpec((cd,), "norm_b", cd)
                tab[f"{p}.{side}.3.weight"] = ParamSpec((cd, cd, kd), "conv_w", cd * kd)
            tab[f"{p}.out_conv.0.weight"] = ParamSpec((cd,), "prelu", cd)
            tab[f"{p}.out_conv.1.norm.weight"] = ParamSpec((cd,), "norm_w", cd)
            tab[f"{p}.out_conv.1.norm.bias"] = ParamSpec((cd,), "norm_b", cd)
            tab[f"{p}.out_conv.2.weight"] = ParamSpec((D, cd, 1), "conv_w", cd)
    return tab
