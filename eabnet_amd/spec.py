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
  * UNet_Encoder 199-238 / UNet_Decoder 282-328 (``is_u2=False``)  ``en|de.unet_list.{i}``
  * pointwise beam-former heads (``bf_type="cnn"``, ``topo_type="miso"``) 78-81  ``bf_map``
  * BatchNorm branch of NormSwitch 677-681: adds the three running-statistics buffers per norm
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
        """Every switch of the reference constructor is implemented (U2/plain U-Net, lstm/cnn
        beam-former, mimo/miso, cat/add, IN/BN, causal or not) and the time extent of the gated convolutions
        (k1 = (1..5, 3)).  The rest of the kernel geometry (64-channel layers, 3-bin kernels, (1,3) units) is fixed; ``cLN`` cannot be constructed in the reference
        either (EaBNet.py:689-691 pass the string dim_size as num_features)."""
        bad = []
        if self.bf_type not in ("lstm", "cnn"):
            bad.append(f"bf_type={self.bf_type!r}")
        if self.topo_type not in ("mimo", "miso"):
            bad.append(f"topo_type={self.topo_type!r}")
        if self.intra_connect not in ("cat", "add"):
            bad.append(f"intra_connect={self.intra_connect!r}")
        if self.norm_type not in ("IN", "BN", "cLN"):
            bad.append(f"norm_type={self.norm_type!r}")
        if self.norm_type == "cLN" and not (self.is_u2 and self.intra_connect == "cat" and self.is_causal):
            bad.append("norm_type='cLN' is built for the default topology (U2, 'cat' skips, causal)")
        if not self.is_causal and (self.kd1 - 1) % 2:
            bad.append("is_causal=False with even kd1 (the reference's residual add fails on the shorter branch)")
        # the gated convolutions' time extent is free (causal pad / chomp of k_t - 1 rows, EaBNet.py:447-452,477-482; k_t * 3 taps
        # must fit the kernels' tap table); k_t = 1 has no pad / chomp module and with it other state-dict keys (".conv.weight"
        # instead of ".conv.1.weight" / ".conv.0.weight", EaBNet.py:452-454,482-484): _gate_conv and gate_key() follow that.
        # The frequency extent is tied to the 161 -> 79 -> 39 -> 19 -> 9 -> 4 chain (other widths do not meet their skip
        # connections in the reference either), and a unit kernel with k_t > 1 shortens the utterance (Conv2dunit has no pad).
        if len(tuple(self.k1)) != 2 or self.k1[1] != 3 or not 1 <= self.k1[0] <= 5 or tuple(self.k2) != (1, 3):
            bad.append(f"k1={self.k1} (supported: (1..5, 3)) k2={self.k2} (supported: (1, 3))")
        if self.c != 64 or self.embed_dim != 64 or self.cd1 != 64:
            bad.append("c/embed_dim/cd1 != 64")
        if self.d_feat != self.c_end * 4:
            bad.append("d_feat != 4*64")
        if self.kd1 < 1 or self.p < 1 or self.q < 1 or self.M < 1:
            bad.append("kd1/p/q/M < 1")
        if bad:
            raise NotImplementedError(
                "eabnet_amd implements EaBNet with the reference's layer geometry "
                "on MI355X; unsupported option(s): " + ", ".join(bad))


@dataclass(frozen=True)
class ParamSpec:
    shape: Tuple[int, ...]
    kind: str       # conv_w | convT_w | bias | norm_w | norm_b | prelu | lstm | lin_w | ln_w | ln_b
    #                 | bn_mean | bn_var | bn_count   (buffers of the BatchNorm branch)
    fan_in: int     # for the default initialiser

    @property
    def is_buffer(self) -> bool:
        return self.kind in ("bn_mean", "bn_var", "bn_count")


def _norm(tab, prefix_norm: str, c: int, bn, dim: int = 2) -> None:
    """NormSwitch (EaBNet.py:662-694): affine InstanceNorm, BatchNorm with its buffers (bn=True), or -- bn == "cLN" -- the
    cumulative LayerNorm of EaBNet.py:696-769 with its (1,C,1[,1]) gain / bias (the reference's constructor passes the
    string dim_size as num_features, :689,691, and cannot be built; this is the class with ``c`` passed instead)."""
    if bn == "cLN":
        shape = (1, c, 1, 1) if dim == 2 else (1, c, 1)
        tab[f"{prefix_norm}.norm.gain"] = ParamSpec(shape, "norm_w", c)
        tab[f"{prefix_norm}.norm.bias"] = ParamSpec(shape, "norm_b", c)
        return
    tab[f"{prefix_norm}.norm.weight"] = ParamSpec((c,), "norm_w", c)
    tab[f"{prefix_norm}.norm.bias"] = ParamSpec((c,), "norm_b", c)
    if bn:
        tab[f"{prefix_norm}.norm.running_mean"] = ParamSpec((c,), "bn_mean", c)
        tab[f"{prefix_norm}.norm.running_var"] = ParamSpec((c,), "bn_var", c)
        tab[f"{prefix_norm}.norm.num_batches_tracked"] = ParamSpec((), "bn_count", c)


def _norm_prelu(tab, prefix_norm: str, prefix_act: str, c: int, bn: bool = False) -> None:
    _norm(tab, prefix_norm, c, bn)
    tab[f"{prefix_act}.weight"] = ParamSpec((c,), "prelu", c)


def gate_key(params, wkey: str) -> str:
    """Parameter prefix of a gated convolution: ``wkey`` ("...conv.1" behind the causal pad, "...conv.0" in front of the chomp)
    or, for a one-frame kernel, the bare "...conv" the reference then registers (EaBNet.py:452-454,482-484)."""
    return wkey if f"{wkey}.weight" in params else wkey.rsplit(".", 1)[0]


def _gate_conv(tab, prefix: str, cin: int, cout: int, k, transposed: bool) -> None:
    kt, kf = k
    if transposed:
        # ConvTranspose2d weight is (cin, 2*cout, kt, kf); PyTorch's default
        # initialiser takes fan_in from dim 1.
        # (k_t = 1: no Chomp_T behind the transposed convolution, hence no nn.Sequential and no index in the key, EaBNet.py:482-484)
        name = f"{prefix}.conv.0" if kt > 1 else f"{prefix}.conv"
        tab[f"{name}.weight"] = ParamSpec((cin, 2 * cout, kt, kf), "convT_w", 2 * cout * kt * kf)
        tab[f"{name}.bias"] = ParamSpec((2 * cout,), "bias", 2 * cout * kt * kf)
    else:
        idx = 1 if kt > 1 else None
        name = f"{prefix}.conv.{idx}" if idx is not None else f"{prefix}.conv"
        tab[f"{name}.weight"] = ParamSpec((2 * cout, cin, kt, kf), "conv_w", cin * kt * kf)
        tab[f"{name}.bias"] = ParamSpec((2 * cout,), "bias", cin * kt * kf)


def _unet_module(tab, prefix: str, cin: int, cout: int, k1, k2, scale: int, is_deconv: bool,
                 bn: bool = False, add: bool = False) -> None:
    _gate_conv(tab, f"{prefix}.in_conv.0", cin, cout, k1, is_deconv)
    _norm_prelu(tab, f"{prefix}.in_conv.1", f"{prefix}.in_conv.2", cout, bn)
    kt, kf = k2
    for j in range(scale):
        p = f"{prefix}.enco.{j}.conv"
        tab[f"{p}.0.weight"] = ParamSpec((cout, cout, kt, kf), "conv_w", cout * kt * kf)
        tab[f"{p}.0.bias"] = ParamSpec((cout,), "bias", cout * kt * kf)
        _norm_prelu(tab, f"{p}.1", f"{p}.2", cout, bn)
    for j in range(scale):
        p = f"{prefix}.deco.{j}.deconv"
        cin_j = cout if (j == 0 or add) else 2 * cout   # first inner deconv has no skip; 'add' sums the skip in
        tab[f"{p}.0.weight"] = ParamSpec((cin_j, cout, kt, kf), "convT_w", cout * kt * kf)
        tab[f"{p}.0.bias"] = ParamSpec((cout,), "bias", cout * kt * kf)
        _norm_prelu(tab, f"{p}.1", f"{p}.2", cout, bn)


def param_specs(cfg: NetConfig) -> "OrderedDict[str, ParamSpec]":
    """Ordered ``key -> ParamSpec`` (parameters and buffers) for the configured topology."""
    cfg.check_supported()
    tab: "OrderedDict[str, ParamSpec]" = OrderedDict()
    c, M = cfg.c, cfg.M
    bn, add = {"BN": True, "IN": False, "cLN": "cLN"}[cfg.norm_type], cfg.intra_connect == "add"

    if cfg.is_u2:
        # encoder: scales 4,3,2,1 then a gated conv down to F=4
        en_k = [cfg.k_beg, cfg.k1, cfg.k1, cfg.k1]
        en_cin = [2 * M, c, c, c]
        for i in range(4):
            _unet_module(tab, f"en.meta_unet_list.{i}", en_cin[i], c, en_k[i], cfg.k2, 4 - i, False, bn, add)
        _gate_conv(tab, "en.last_conv.0", c, cfg.c_end, cfg.k1, False)
        _norm_prelu(tab, "en.last_conv.1", "en.last_conv.2", cfg.c_end, bn)

        # decoder: scales 1,2,3,4 on cat(skip) inputs, then the (2,5) gated deconv
        for i in range(4):
            cin = 2 * cfg.c_end if i == 0 else 2 * c
            _unet_module(tab, f"de.meta_unet_list.{i}", cin, c, cfg.k1, cfg.k2, i + 1, True, bn, add)
        _gate_conv(tab, "de.last_conv.0", 2 * c, cfg.embed_dim, cfg.k_beg, True)
        _norm_prelu(tab, "de.last_conv.1", "de.last_conv.2", cfg.embed_dim, bn)
    else:
        # plain U-Net (EaBNet.py:199-238, 282-328): five gated convs down, five gated deconvs up;
        # encoder layers 1 and 2 have a PReLU but no norm
        for i, (cin, cout, k, has_norm) in enumerate(unet_encoder_layers(cfg)):
            _gate_conv(tab, f"en.unet_list.{i}.0", cin, cout, k, False)
            if has_norm:
                _norm_prelu(tab, f"en.unet_list.{i}.1", f"en.unet_list.{i}.2", cout, bn)
            else:
                tab[f"en.unet_list.{i}.1.weight"] = ParamSpec((cout,), "prelu", cout)
        for i, (cin, cout, k) in enumerate(unet_decoder_layers(cfg)):
            _gate_conv(tab, f"de.unet_list.{i}.0", cin, cout, k, True)
            _norm_prelu(tab, f"de.unet_list.{i}.1", f"de.unet_list.{i}.2", cout, bn)

    # beamformer head
    E, H = cfg.embed_dim, cfg.hid_node
    if cfg.topo_type == "mimo" and cfg.bf_type == "lstm":
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
    else:
        # pointwise Conv2d head (EaBNet.py:78-81): 2M weight planes (mimo/cnn) or one complex mask (miso)
        n_out = 2 * M if cfg.topo_type == "mimo" else 2
        tab["bf_map.weight"] = ParamSpec((n_out, E, 1, 1), "conv_w", E)
        tab["bf_map.bias"] = ParamSpec((n_out,), "bias", E)

    # squeezed-TCN bottleneck
    D, cd, kd = cfg.d_feat, cfg.cd1, cfg.kd1
    for g in range(cfg.q):
        for i in range(cfg.p):
            p = f"stcns.{g}.tcm_list.{i}"
            tab[f"{p}.in_conv.weight"] = ParamSpec((cd, D, 1), "conv_w", D)
            for side in ("left_conv", "right_conv"):
                tab[f"{p}.{side}.0.weight"] = ParamSpec((cd,), "prelu", cd)
                _norm(tab, f"{p}.{side}.1", cd, bn, dim=1)
                tab[f"{p}.{side}.3.weight"] = ParamSpec((cd, cd, kd), "conv_w", cd * kd)
            tab[f"{p}.out_conv.0.weight"] = ParamSpec((cd,), "prelu", cd)
            _norm(tab, f"{p}.out_conv.1", cd, bn, dim=1)
            tab[f"{p}.out_conv.2.weight"] = ParamSpec((D, cd, 1), "conv_w", cd)
    return tab


def unet_encoder_layers(cfg: NetConfig):
    """(cin, cout, kernel, has_norm) of UNet_Encoder's five gated convs (EaBNet.py:213-232)."""
    c = cfg.c
    return [(2 * cfg.M, c, cfg.k_beg, True), (c, c, cfg.k1, False), (c, c, cfg.k1, False),
            (c, c, cfg.k1, True), (c, cfg.c_end, cfg.k1, True)]


def unet_decoder_layers(cfg: NetConfig):
    """(cin, cout, kernel) of UNet_Decoder's five gated deconvs on cat(skip) inputs (EaBNet.py:297-321)."""
    c = cfg.c
    return [(2 * cfg.c_end, c, cfg.k1), (2 * c, c, cfg.k1), (2 * c, c, cfg.k1), (2 * c, c, cfg.k1),
            (2 * c, cfg.embed_dim, cfg.k_beg)]


# ----------------------------------------------------------------------------
# GaGNet post-filter (reference GaGNet.py), the second stage of EaBNetWithPostNet
# ----------------------------------------------------------------------------
@dataclass(frozen=True)
class GagConfig:
    """Hyper-parameters of GaGNet.__init__ (reference GaGNet.py:6-24; defaults = the ``gagnet_*``
    arguments of train_distributed.py:303-318)."""
    cin: int = 2
    k1: Tuple[int, int] = (2, 3)
    k2: Tuple[int, int] = (1, 3)
    c: int = 64
    kd1: int = 3
    cd1: int = 64
    d_feat: int = 256
    p: int = 2
    q: int = 3
    dilas: Tuple[int, ...] = (1, 2, 5, 9)
    fft_num: int = 320
    is_u2: bool = True
    is_causal: bool = True
    is_squeezed: bool = False
    acti_type: str = "sigmoid"
    intra_connect: str = "cat"
    norm_type: str = "IN"

    k_beg: Tuple[int, int] = (2, 5)
    c_end: int = 64

    @property
    def freq(self) -> int:
        return self.fft_num // 2 + 1

    def check_supported(self) -> None:
        bad = []
        if self.cin != 2:
            bad.append(f"cin={self.cin} (the post-filter runs on real/imag pairs)")
        if self.acti_type not in ("sigmoid", "tanh", "relu"):
            bad.append(f"acti_type={self.acti_type!r}")
        if self.intra_connect not in ("cat", "add"):
            bad.append(f"intra_connect={self.intra_connect!r}")
        # (the post-filter is offered with BN / IN only: train_distributed.py:318 `--gagnet_norm_type`, choices ["BN", "IN"])
        if self.norm_type not in ("IN", "BN"):
            bad.append(f"norm_type={self.norm_type!r}")
        # (k1 = (k_t, 3): the gated convolutions' time extent is free, as in NetConfig.check_supported)
        if len(tuple(self.k1)) != 2 or self.k1[1] != 3 or not 1 <= self.k1[0] <= 5 or tuple(self.k2) != (1, 3) \
                or self.c != 64 or self.cd1 != 64:
            bad.append("k1/k2/c/cd1 away from (1..5,3)/(1,3)/64/64")
        if self.d_feat != self.c_end * 4 or self.fft_num != 320:
            bad.append("d_feat != 256 or fft_num != 320")
        if not self.is_causal and any(((self.kd1 - 1) * d) % 2 for d in self.dilas):
            bad.append("is_causal=False with an odd (kd1-1)*dilation (the reference's residual add fails)")
        if self.kd1 < 1 or self.p < 1 or self.q < 1 or not self.dilas:
            bad.append("kd1/p/q/dilas")
        if bad:
            raise NotImplementedError("eabnet_amd.GaGNet: unsupported option(s): " + ", ".join(bad))


def gag_param_specs(cfg: GagConfig) -> "OrderedDict[str, ParamSpec]":
    """Ordered ``key -> ParamSpec`` of GaGNet (reference GaGNet.py:68-74 and the block constructors
    :136-327)."""
    cfg.check_supported()
    tab: "OrderedDict[str, ParamSpec]" = OrderedDict()
    c = cfg.c
    bn, add = cfg.norm_type == "BN", cfg.intra_connect == "add"
    if cfg.is_u2:
        en_k = [cfg.k_beg, cfg.k1, cfg.k1, cfg.k1]
        en_cin = [2 * cfg.cin, c, c, c]
        for i in range(4):
            _unet_module(tab, f"en.meta_unet_list.{i}", en_cin[i], c, en_k[i], cfg.k2, 4 - i, False, bn, add)
        _gate_conv(tab, "en.last_conv.0", c, cfg.c_end, cfg.k1, False)
        _norm_prelu(tab, "en.last_conv.1", "en.last_conv.2", cfg.c_end, bn)
    else:
        # GaGNet's plain encoder normalises every layer (GaGNet.py:383-406), unlike EaBNet's
        for i in range(5):
            _gate_conv(tab, f"en.unet_list.{i}.0", 2 * cfg.cin if i == 0 else c, cfg.c_end if i == 4 else c,
                       cfg.k_beg if i == 0 else cfg.k1, False)
            _norm_prelu(tab, f"en.unet_list.{i}.1", f"en.unet_list.{i}.2", cfg.c_end if i == 4 else c, bn)

    D, cd, kd, Fq = cfg.d_feat, cfg.cd1, cfg.kd1, cfg.freq
    ci = 2 * Fq + D

    def gated_in(p):
        tab[f"{p}.in_conv_main.weight"] = ParamSpec((D, ci, 1), "conv_w", ci)
        tab[f"{p}.in_conv_main.bias"] = ParamSpec((D,), "bias", ci)
        tab[f"{p}.in_conv_gate.0.weight"] = ParamSpec((D, ci, 1), "conv_w", ci)
        tab[f"{p}.in_conv_gate.0.bias"] = ParamSpec((D,), "bias", ci)

    def tcn_chain(p):
        for j in range(cfg.p):
            for k in range(len(cfg.dilas)):
                t = f"{p}.{j}.tcns.{k}"
                tab[f"{t}.in_conv.weight"] = ParamSpec((cd, D, 1), "conv_w", D)
                tab[f"{t}.d_conv.0.weight"] = ParamSpec((cd,), "prelu", cd)
                _norm(tab, f"{t}.d_conv.1", cd, bn)
                tab[f"{t}.d_conv.3.weight"] = ParamSpec((cd, cd, kd), "conv_w", cd * kd)
                tab[f"{t}.out_conv.0.weight"] = ParamSpec((cd,), "prelu", cd)
                _norm(tab, f"{t}.out_conv.1", cd, bn)
                tab[f"{t}.out_conv.2.weight"] = ParamSpec((D, cd, 1), "conv_w", cd)

    def linear(p):
        tab[f"{p}.weight"] = ParamSpec((Fq, D, 1), "conv_w", D)
        tab[f"{p}.bias"] = ParamSpec((Fq,), "bias", D)

    for g in range(cfg.q):
        gl, gz = f"gags.{g}.glance_block", f"gags.{g}.gaze_block"
        gated_in(gl)
        tcn_chain(f"{gl}.tcn_g")
        linear(f"{gl}.linear_g.0")
        gated_in(gz)
        for name in (("tcm_ri",) if cfg.is_squeezed else ("tcm_r", "tcm_i")):
            tcn_chain(f"{gz}.{name}")
        linear(f"{gz}.linear_r")
        linear(f"{gz}.linear_i")
    return tab
