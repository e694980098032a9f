"""COMPARATOR, not product code: EaBNet / GaGNet evaluated with PyTorch-ROCm operators (MIOpen convolutions, ATen
LSTM) on a module's own ``nn.Parameter``s, differentiable through ordinary autograd.

Until round 3 this file lived in the package (eabnet_amd/autograd_path.py) and served differentiable calls the HIP training
programs did not cover; the package now has ONE backend (a differentiable call runs eabnet_amd/train.py / train_gag.py or
raises) and this file is what the GPU tests and ``bench.py``'s untimed in-run check compare the HIP training programs
against (``test_config3_full_size_*``, ``test_hip_training_step_matches_operator_path``, the torch.autocast(bf16) side of
the bf16 gradient test, ``bench.py --train --train-operator-path``).  Nothing under eabnet_amd/ imports it.

The network definition follows the reference's forward (EaBNet.py:88-117 and the block forwards :372-388, :455-460,
:485-490, :572-578, :600-614; GaGNet.py:76-133); parameters are looked up by their state-dict keys (eabnet_amd/spec.py).
``OperatorPath(module)`` wraps an eabnet_amd module (EaBNet, GaGNet or EaBNetWithPostNet) so that a differentiable call
runs here, on the SAME parameter objects, and a ``torch.no_grad()`` call runs the module's own HIP inference program.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _norm(m, x, norm: str):
    """NormSwitch (EaBNet.py:662-694).  BatchNorm follows nn.BatchNorm*d: batch statistics and a
    momentum-0.1 update of the running buffers in training mode, running statistics in eval mode."""
    if m.norm_type == "cLN":                  # CumulativeLayerNorm1d / 2d (EaBNet.py:696-769), differentiable
        g, b = m.get_parameter(f"{norm}.norm.gain"), m.get_parameter(f"{norm}.norm.bias")
        dims = (1,) if x.ndim == 3 else (1, 3)
        per_frame = x.shape[1] * (x.shape[3] if x.ndim == 4 else 1)
        cs, cq = torch.cumsum(x.sum(dims), dim=1), torch.cumsum(x.pow(2).sum(dims), dim=1)
        cnt = per_frame * torch.arange(1, x.shape[2] + 1, dtype=x.dtype, device=x.device).view(1, -1)
        mean = cs / cnt
        std = ((cq - 2 * mean * cs) / cnt + mean.pow(2) + 1e-5).sqrt()
        shape = (x.shape[0], 1, x.shape[2]) + ((1,) if x.ndim == 4 else ())
        return (x - mean.view(shape)) / std.view(shape) * g + b
    w, b = m.get_parameter(f"{norm}.norm.weight"), m.get_parameter(f"{norm}.norm.bias")
    if m.norm_type == "BN":
        if m.training:
            m.get_buffer(f"{norm}.norm.num_batches_tracked").add_(1)
        return F.batch_norm(x, m.get_buffer(f"{norm}.norm.running_mean"), m.get_buffer(f"{norm}.norm.running_var"),
                            w, b, training=m.training, momentum=0.1, eps=1e-5)
    return F.instance_norm(x, weight=w, bias=b, use_input_stats=True, eps=1e-5)


def _norm_act(m, x, norm: str, act: str):
    return F.prelu(_norm(m, x, norm), m.get_parameter(f"{act}.weight"))


def _glu(y):
    a, g = y.chunk(2, dim=1)
    return a * torch.sigmoid(g)


def _gkey(m, key: str) -> str:
    """"...conv.1" / "...conv.0", or the bare "...conv" of a one-frame gated kernel (EaBNet.py:452-454,482-484)"""
    try:
        m.get_parameter(f"{key}.weight")
        return key
    except AttributeError:
        return key.rsplit(".", 1)[0]


def _gate_conv(m, x, key: str):
    key = _gkey(m, key)
    w, b = m.get_parameter(f"{key}.weight"), m.get_parameter(f"{key}.bias")
    return _glu(F.conv2d(F.pad(x, (0, 0, w.shape[2] - 1, 0)), w, b, stride=(1, 2)))      # causal top pad


def _gate_deconv(m, x, key: str):
    key = _gkey(m, key)
    w, b = m.get_parameter(f"{key}.weight"), m.get_parameter(f"{key}.bias")
    y = F.conv_transpose2d(x, w, b, stride=(1, 2))
    kt = w.shape[2]
    return _glu(y[:, :, :y.shape[2] - (kt - 1)] if kt > 1 else y)                          # chomp the last rows


def _unet(m, x, pre: str, scale: int, transposed: bool):
    g = _gate_deconv(m, x, f"{pre}.in_conv.0.conv.0") if transposed else _gate_conv(m, x, f"{pre}.in_conv.0.conv.1")
    resi = _norm_act(m, g, f"{pre}.in_conv.1", f"{pre}.in_conv.2")
    y, downs = resi, []
    for j in range(scale):
        q = f"{pre}.enco.{j}.conv"
        y = _norm_act(m, F.conv2d(y, m.get_parameter(f"{q}.0.weight"), m.get_parameter(f"{q}.0.bias"), stride=(1, 2)),
                      f"{q}.1", f"{q}.2")
        downs.append(y)
    for j in range(scale):
        q = f"{pre}.deco.{j}.deconv"
        if j:
            y = y + downs[-(j + 1)] if m.intra_connect == "add" else torch.cat((y, downs[-(j + 1)]), dim=1)
        y = _norm_act(m, F.conv_transpose2d(y, m.get_parameter(f"{q}.0.weight"), m.get_parameter(f"{q}.0.bias"),
                                            stride=(1, 2)), f"{q}.1", f"{q}.2")
    return resi + y


def _tcm(m, x, pre: str, dilation: int):
    span = (m.kd1 - 1) * dilation
    pad = (span, 0) if m.is_causal else (span // 2, span // 2)          # EaBNet.py:550-553
    y = F.conv1d(x, m.get_parameter(f"{pre}.in_conv.weight"))

    def branch(side):
        z = _norm(m, F.prelu(y, m.get_parameter(f"{pre}.{side}.0.weight")), f"{pre}.{side}.1")
        return F.conv1d(F.pad(z, pad), m.get_parameter(f"{pre}.{side}.3.weight"), dilation=dilation)

    z = branch("left_conv") * torch.sigmoid(branch("right_conv"))
    z = _norm(m, F.prelu(z, m.get_parameter(f"{pre}.out_conv.0.weight")), f"{pre}.out_conv.1")
    return F.conv1d(z, m.get_parameter(f"{pre}.out_conv.2.weight")) + x


def _lstm(m, x, name: str):
    p = f"bf_map.{name}"
    flat = [m.get_parameter(f"{p}.weight_ih_l0"), m.get_parameter(f"{p}.weight_hh_l0"),
            m.get_parameter(f"{p}.bias_ih_l0"), m.get_parameter(f"{p}.bias_hh_l0")]
    z = x.new_zeros(1, x.shape[0], flat[1].shape[1])
    return torch._VF.lstm(x, (z, z), flat, True, 1, 0.0, m.training, False, True)[0]


def forward_autograd(m, inpt: torch.Tensor) -> torch.Tensor:
    """(B,T,F,M,2) -> (B,2,T,F) [(B,2,T) for topo_type="miso"], differentiable w.r.t. the module's
    parameters and the input."""
    if inpt.ndim == 4:
        inpt = inpt.unsqueeze(-2)
    B, T, Fq, M, _ = inpt.shape
    x = inpt.transpose(-2, -1).contiguous().view(B, T, Fq, 2 * M).permute(0, 3, 1, 2)     # channel = ri*M + m
    skips = []
    if m.is_u2:
        for i in range(4):
            x = _unet(m, x, f"en.meta_unet_list.{i}", 4 - i, False)
            skips.append(x)
        x = _norm_act(m, _gate_conv(m, x, "en.last_conv.0.conv.1"), "en.last_conv.1", "en.last_conv.2")
        skips.append(x)
    else:                                                 # UNet_Encoder, EaBNet.py:234-239
        for i in range(5):
            q = f"en.unet_list.{i}"
            x = _gate_conv(m, x, f"{q}.0.conv.1")
            x = F.prelu(x, m.get_parameter(f"{q}.1.weight")) if i in (1, 2) else _norm_act(m, x, f"{q}.1", f"{q}.2")
            skips.append(x)
    C = x.shape[1]
    x = x.transpose(-2, -1).contiguous().view(B, -1, T)
    acc = torch.zeros_like(x)
    for g in range(m.q):
        for i in range(m.p):
            x = _tcm(m, x, f"stcns.{g}.tcm_list.{i}", 2 ** i)
        acc = acc + x
    x = acc.view(B, C, -1, T).transpose(-2, -1).contiguous()
    if m.is_u2:
        for i in range(4):
            x = _unet(m, torch.cat((x, skips[-(i + 1)]), dim=1), f"de.meta_unet_list.{i}", i + 1, True)
        x = _norm_act(m, _gate_deconv(m, torch.cat((x, skips[0]), dim=1), "de.last_conv.0.conv.0"),
                      "de.last_conv.1", "de.last_conv.2")
    else:                                                 # UNet_Decoder, EaBNet.py:324-328
        for i in range(5):
            q = f"de.unet_list.{i}"
            x = _norm_act(m, _gate_deconv(m, torch.cat((x, skips[-(i + 1)]), dim=1), f"{q}.0.conv.0"), f"{q}.1", f"{q}.2")
    if m.topo_type == "miso":                             # EaBNet.py:118-125 (sum over frequency, as written there)
        k = F.conv2d(x, m.get_parameter("bf_map.weight"), m.get_parameter("bf_map.bias")).permute(0, 2, 3, 1)
        kr, ki, xr, xi = k[..., 0], k[..., -1], inpt[..., 0, 0], inpt[..., 0, -1]
        return torch.stack(((kr * xr - ki * xi).sum(-1), (kr * xi + ki * xr).sum(-1)), dim=1)
    if m.bf_type == "cnn":                                # EaBNet.py:111-113
        w = F.conv2d(x, m.get_parameter("bf_map.weight"), m.get_parameter("bf_map.bias"))
        w = w.view(B, M, -1, T, Fq).permute(0, 3, 4, 1, 2)
        wr, wi, xr, xi = w[..., 0], w[..., 1], inpt[..., 0], inpt[..., 1]
        return torch.stack(((wr * xr - wi * xi).sum(-1), (wr * xi + wi * xr).sum(-1)), dim=1)
    # LSTM_BF
    e = F.layer_norm(x.permute(0, 3, 2, 1).contiguous(), (C,), m.get_parameter("bf_map.norm.weight"),
                     m.get_parameter("bf_map.norm.bias"), 1e-5).view(B * Fq, T, C)
    h = _lstm(m, _lstm(m, e, "rnn1"), "rnn2").view(B, Fq, T, -1).transpose(1, 2).contiguous()
    h = F.relu(F.linear(h, m.get_parameter("bf_map.w_dnn.0.weight"), m.get_parameter("bf_map.w_dnn.0.bias")))
    w = F.linear(h, m.get_parameter("bf_map.w_dnn.2.weight"), m.get_parameter("bf_map.w_dnn.2.bias")).view(B, T, Fq, M, 2)
    wr, wi, xr, xi = w[..., 0], w[..., 1], inpt[..., 0], inpt[..., 1]
    return torch.stack(((wr * xr - wi * xi).sum(-1), (wr * xi + wi * xr).sum(-1)), dim=1)


# ----------------------------------------------------------------------------
# GaGNet post-filter (reference GaGNet.py:76-133), differentiable
# ----------------------------------------------------------------------------
def _gag_tcm(m, x, pre: str, dilation: int):
    span = (m.kd1 - 1) * dilation
    pad = (span, 0) if m.is_causal else (span // 2, span // 2)
    y = F.conv1d(x, m.get_parameter(f"{pre}.in_conv.weight"))
    y = _norm(m, F.prelu(y, m.get_parameter(f"{pre}.d_conv.0.weight")), f"{pre}.d_conv.1")
    y = F.conv1d(F.pad(y, pad), m.get_parameter(f"{pre}.d_conv.3.weight"), dilation=dilation)
    y = _norm(m, F.prelu(y, m.get_parameter(f"{pre}.out_conv.0.weight")), f"{pre}.out_conv.1")
    return F.conv1d(y, m.get_parameter(f"{pre}.out_conv.2.weight")) + x


def _gag_chain(m, x, pre: str):
    for j in range(m.p):
        for k, d in enumerate(m.dilas):
            x = _gag_tcm(m, x, f"{pre}.{j}.tcns.{k}", d)
    return x


def forward_gagnet(m, inpt: torch.Tensor, pre_x: torch.Tensor) -> list:
    """inpt, pre_x (B,2,T,F) -> list of q (B,2,F,T), differentiable."""
    B, _, T, Fq = inpt.shape
    x = torch.cat([inpt, pre_x], dim=1)
    if m.is_u2:
        for i in range(4):
            x = _unet(m, x, f"en.meta_unet_list.{i}", 4 - i, False)
        x = _norm_act(m, _gate_conv(m, x, "en.last_conv.0.conv.1"), "en.last_conv.1", "en.last_conv.2")
    else:
        for i in range(5):
            q = f"en.unet_list.{i}"
            x = _norm_act(m, _gate_conv(m, x, f"{q}.0.conv.1"), f"{q}.1", f"{q}.2")
    feat = x.transpose(-2, -1).contiguous().view(B, -1, T)
    pre = pre_x.transpose(-2, -1).contiguous()
    act = {"sigmoid": torch.sigmoid, "tanh": torch.tanh, "relu": torch.relu}[m.acti_type]
    outs = []
    for g in range(m.q):
        gl, gz = f"gags.{g}.glance_block", f"gags.{g}.gaze_block"
        cat = torch.cat((feat, pre.view(B, -1, T)), dim=1)

        def gated(pfx):
            return F.conv1d(cat, m.get_parameter(f"{pfx}.in_conv_main.weight"), m.get_parameter(f"{pfx}.in_conv_main.bias")) \
                * torch.sigmoid(F.conv1d(cat, m.get_parameter(f"{pfx}.in_conv_gate.0.weight"),
                                         m.get_parameter(f"{pfx}.in_conv_gate.0.bias")))

        xg = _gag_chain(m, gated(gl), f"{gl}.tcn_g")
        gain = act(F.conv1d(xg, m.get_parameter(f"{gl}.linear_g.0.weight"), m.get_parameter(f"{gl}.linear_g.0.bias")))
        xz = gated(gz)
        if m.is_squeezed:
            xr = xi = _gag_chain(m, xz, f"{gz}.tcm_ri")
        else:
            xr, xi = _gag_chain(m, xz, f"{gz}.tcm_r"), _gag_chain(m, xz, f"{gz}.tcm_i")
        resi = torch.stack((F.conv1d(xr, m.get_parameter(f"{gz}.linear_r.weight"), m.get_parameter(f"{gz}.linear_r.bias")),
                            F.conv1d(xi, m.get_parameter(f"{gz}.linear_i.weight"), m.get_parameter(f"{gz}.linear_i.bias"))),
                           dim=1)
        mag, ph = torch.norm(pre, dim=1), torch.atan2(pre[:, -1], pre[:, 0])              # GaGNet.py:129-132
        filt = mag * gain
        pre = torch.stack((filt * torch.cos(ph), filt * torch.sin(ph)), dim=1) + resi
        outs.append(pre)
    return outs


# ----------------------------------------------------------------------------
# wrapper: the same call surface as the wrapped module, differentiable calls on operators
# ----------------------------------------------------------------------------
class OperatorPath(torch.nn.Module):
    """``OperatorPath(net)(x)`` == ``net(x)`` with the differentiable work done by PyTorch-ROCm operators.  Parameters and
    buffers are the wrapped module's own objects (``parameters()`` / optimisers / DDP see them under the prefix ``net.``)."""

    def __init__(self, net: torch.nn.Module):
        super().__init__()
        self.net = net

    @staticmethod
    def _needs_graph(net, inputs) -> bool:
        return torch.is_grad_enabled() and (any(t.requires_grad for t in inputs) or any(p.requires_grad for p in net.parameters())) \
            or (getattr(net, "norm_type", None) == "BN" and net.training)

    def _one(self, net, *inputs):
        import eabnet_amd
        if not self._needs_graph(net, inputs):
            return net(*inputs)                                   # the module's own HIP inference program
        if isinstance(net, eabnet_amd.GaGNet):
            return forward_gagnet(net, *inputs)
        return forward_autograd(net, *inputs)

    def forward(self, *inputs):
        import eabnet_amd
        net = self.net
        if isinstance(net, eabnet_amd.EaBNetWithPostNet):         # EaBNet.py:138-148
            noisy = inputs[0]
            esti0 = self._one(net.eabnet, noisy)
            inpt = noisy[..., net.ref_mic, :].permute(0, 3, 1, 2)
            lst = self._one(net.postnet, inpt, esti0.detach())
            return {"esti0_stft": esti0, "esti1_stft_list": lst, "esti_stft": lst[-1].permute(0, 1, 3, 2)}
        return self._one(net, *inputs)
