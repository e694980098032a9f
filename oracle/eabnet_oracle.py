"""CPU ORACLE for the EaBNet hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module, and only as the checker.  ``eabnet_amd`` never
imports it; the product path fails loudly when the HIP library is missing.

What it is: a functional (stateless, parameters-as-dict) restatement on
PyTorch-CPU of ``prepare_data`` (reference train_distributed.py:68-95) and
``EaBNet.forward`` (reference EaBNet.py:88-125): the default topology (is_u2,
lstm beam-former, mimo, cat skips, InstanceNorm, causal) and, by keyword, the other
constructor branches (plain U-Net, cnn/miso heads, add skips, BatchNorm in eval
mode, non-causal S-TCMs).  It uses the
same ATen primitives the reference reaches (conv2d / conv_transpose2d /
conv1d / instance_norm / prelu / layer_norm / linear) so its rounding behaviour
is the reference's; the LSTM is spelled out step by step (gate order i,f,g,o)
with ``fast_lstm=True`` switching to ATen's fused LSTM for the timed baseline.

Parity pin: tests/test_oracle_golden.py checks this file against fixtures in
tests/golden/*.npz that were produced by importing the reference's own
EaBNet.py in the authoring container (tests/golden/make_golden.py).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]
EPS_IN = 1e-5          # nn.InstanceNorm default eps (reference EaBNet.py:684-686)
EPS_LN = 1e-5          # nn.LayerNorm default eps (reference EaBNet.py:598)


# ----------------------------------------------------------------------------
# STFT front end
# ----------------------------------------------------------------------------
def frame_index(L: int, n_fft: int, hop: int):
    """Sample index read by (frame t, tap n) of a centred, reflect-padded STFT
    (torch.stft(center=True, pad_mode='reflect'), reference
    train_distributed.py:83).  Returns an int64 (T, n_fft) tensor of indices into
    the un-padded wave; T = 1 + L // hop."""
    pad = n_fft // 2
    T = 1 + L // hop
    j = torch.arange(T).unsqueeze(1) * hop + torch.arange(n_fft).unsqueeze(0) - pad
    j = torch.where(j < 0, -j, j)                       # left reflection (no edge repeat)
    j = torch.where(j >= L, 2 * (L - 1) - j, j)         # right reflection
    return j


def stft_frames(wav: torch.Tensor, n_fft: int, hop: int) -> torch.Tensor:
    """(N, L) -> (N, T, n_fft) raw (un-windowed) frames; pure indexing."""
    return wav[:, frame_index(wav.shape[-1], n_fft, hop)]


def hann_periodic(n: int, dtype=torch.float32) -> torch.Tensor:
    """The window prepare_data builds (train_distributed.py:83):
    torch.hann_window(n), periodic, = 0.5 - 0.5 cos(2 pi k / n) evaluated in
    fp32 by ATen (1-3 ulp away from the double-rounded formula, so the
    product takes the window as an input instead of recomputing it)."""
    return torch.hann_window(n, dtype=dtype)


def padded_window(n_fft: int, win: int, dtype=torch.float32) -> torch.Tensor:
    """torch.stft(win_length=win < n_fft): the window is zero-padded on both sides to n_fft, centred"""
    w = hann_periodic(win, dtype)
    if win == n_fft:
        return w
    left = (n_fft - win) // 2
    return torch.nn.functional.pad(w, (left, n_fft - win - left))


def stft_oracle(wav: torch.Tensor, n_fft: int, hop: int, win: Optional[int] = None) -> torch.Tensor:
    """(N, L) -> (N, F, T, 2); frames * periodic Hann (zero-padded to n_fft when shorter) -> one-sided rfft."""
    fr = stft_frames(wav, n_fft, hop) * padded_window(n_fft, win or n_fft, wav.dtype)
    X = torch.fft.rfft(fr, n=n_fft, dim=-1)             # (N, T, F)
    return torch.view_as_real(X).permute(0, 2, 1, 3).contiguous()


def compress_oracle(X: torch.Tensor, ri_dim: int) -> torch.Tensor:
    """sqrt-magnitude compression keeping the phase (train_distributed.py:89-92):
    mag = |X|**0.5, phase = atan2(im, re), out = mag*(cos, sin)."""
    re, im = X.select(ri_dim, 0), X.select(ri_dim, -1)
    mag = torch.norm(X, dim=ri_dim) ** 0.5
    ph = torch.atan2(im, re)
    return torch.stack((mag * torch.cos(ph), mag * torch.sin(ph)), dim=ri_dim)


def prepare_data_oracle(x: torch.Tensor, target: Optional[torch.Tensor], n_fft: int = 320,
                        hop: int = 160, win: int = 320
                        ) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """x (B,M,L), target (B,1,L) -> noisy (B,T,F,M,2), target (B,2,T,F)."""
    assert win <= n_fft, "torch.stft needs win_length <= n_fft"
    B, M, L = x.shape
    X = stft_oracle(x.reshape(B * M, L), n_fft, hop, win)          # (B*M, F, T, 2)
    X = X.view(B, M, X.shape[1], X.shape[2], 2).permute(0, 3, 2, 1, 4)   # (B,T,F,M,2)
    noisy = compress_oracle(X, -1).contiguous()
    tgt = None
    if target is not None:
        Y = stft_oracle(target.reshape(B, L), n_fft, hop, win).permute(0, 3, 2, 1)   # (B,2,T,F)
        tgt = compress_oracle(Y, 1).contiguous()
    return noisy, tgt


def istft_oracle(esti: torch.Tensor, n_fft: int = 320, hop: int = 160) -> torch.Tensor:
    """Back end of enhance.py:59-62 spelled out: (B,2,T,F) -> (B, hop*(T-1)).
    irfft per frame (imaginary parts of DC/Nyquist ignored), synthesis window, overlap-add,
    division by the overlap-added squared window, n_fft/2 trimmed from both ends
    (torch.istft with center=True, length=None)."""
    B, _, T, Fq = esti.shape
    X = torch.complex(esti[:, 0], esti[:, 1])                        # (B,T,F)
    w = hann_periodic(n_fft, esti.dtype)
    fr = torch.fft.irfft(X, n=n_fft, dim=-1) * w                     # (B,T,n_fft)
    full = n_fft + hop * (T - 1)
    y = esti.new_zeros(B, full)
    env = esti.new_zeros(full)
    for t in range(T):
        y[:, t * hop:t * hop + n_fft] += fr[:, t]
        env[t * hop:t * hop + n_fft] += w * w
    return y[:, n_fft // 2:full - n_fft // 2] / env[n_fft // 2:full - n_fft // 2]


# ----------------------------------------------------------------------------
# network blocks
# ----------------------------------------------------------------------------
def cumulative_layer_norm(x, gain, bias, eps: float = EPS_IN):
    """CumulativeLayerNorm1d / 2d.forward (EaBNet.py:713-733, 752-769): statistics over all channels, all frequency bins
    and all frames up to the current one.  x (B,C,T) or (B,C,T,F); gain / bias (1,C,1[,1])."""
    dims = (1,) if x.ndim == 3 else (1, 3)
    per_frame = x.shape[1] * (x.shape[3] if x.ndim == 4 else 1)
    cum_sum = torch.cumsum(x.sum(dims), dim=1)                                 # (B,T)
    cum_pow = torch.cumsum(x.pow(2).sum(dims), dim=1)
    cnt = per_frame * torch.arange(1, x.shape[2] + 1, dtype=x.dtype, device=x.device).view(1, -1)
    mean = cum_sum / cnt
    var = (cum_pow - 2 * mean * cum_sum) / cnt + mean.pow(2)
    std = (var + eps).sqrt()
    shape = (x.shape[0], 1, x.shape[2]) + ((1,) if x.ndim == 4 else ())
    return (x - mean.view(shape)) / std.view(shape) * gain + bias


def _norm(x, P: Params, norm_prefix: str, bn=False):
    """NormSwitch (EaBNet.py:662-694): affine InstanceNorm, BatchNorm in eval mode (running statistics; nn.BatchNorm
    default eps 1e-5), or -- bn == "cLN" -- the cumulative LayerNorm the switch means to build (with ``c`` passed where the
    reference passes the string dim_size, EaBNet.py:689,691)."""
    if isinstance(bn, str) and bn == "cLN":
        return cumulative_layer_norm(x, P[f"{norm_prefix}.norm.gain"], P[f"{norm_prefix}.norm.bias"])
    w, b = P[f"{norm_prefix}.norm.weight"], P[f"{norm_prefix}.norm.bias"]
    if isinstance(bn, str) and bn == "train":
        # nn.BatchNorm{1,2}d in train mode (the reference trains with module.train(), train_distributed.py:212): batch
        # statistics, and the momentum-0.1 update of the running buffers (unbiased variance).  The updated buffers are
        # left in P["__bn_updates__"] (when the caller provides that dict) -- P's own entries are not modified.
        rm = P[f"{norm_prefix}.norm.running_mean"].detach().clone()
        rv = P[f"{norm_prefix}.norm.running_var"].detach().clone()
        y = F.batch_norm(x, rm, rv, w, b, training=True, momentum=0.1, eps=EPS_IN)
        if "__bn_updates__" in P:
            P["__bn_updates__"][norm_prefix] = (rm, rv)
        return y
    if bn:
        return F.batch_norm(x, P[f"{norm_prefix}.norm.running_mean"], P[f"{norm_prefix}.norm.running_var"], w, b,
                            training=False, eps=EPS_IN)
    return F.instance_norm(x, weight=w, bias=b, use_input_stats=True, eps=EPS_IN)


def _in_prelu(x, P: Params, norm_prefix: str, act_prefix: str, bn: bool = False):
    return F.prelu(_norm(x, P, norm_prefix, bn), P[f"{act_prefix}.weight"])


def gate_conv2d(x, w, b):
    """GateConv2d (EaBNet.py:434-460): zero-pad k_t-1 rows on top (causal),
    strided conv to 2*C channels, first half * sigmoid(second half)."""
    kt = w.shape[2]
    y = F.conv2d(F.pad(x, (0, 0, kt - 1, 0)), w, b, stride=(1, 2))
    a, g = y.chunk(2, dim=1)
    return a * torch.sigmoid(g)


def gate_deconv2d(x, w, b):
    """GateConvTranspose2d (EaBNet.py:463-490, Chomp_T 617-624): transposed conv,
    drop the LAST k_t-1 time rows, GLU."""
    kt = w.shape[2]
    y = F.conv_transpose2d(x, w, b, stride=(1, 2))
    if kt > 1:
        y = y[:, :, :-(kt - 1), :]
    a, g = y.chunk(2, dim=1)
    return a * torch.sigmoid(g)


def _gk(P: Params, wkey: str) -> str:
    """prefix of a gated convolution's parameters: "...conv.1" (behind the causal pad) / "...conv.0" (in front of the chomp), or
    the bare "...conv" the reference registers for a one-frame kernel (no pad / chomp module, EaBNet.py:452-454,482-484)"""
    return wkey if f"{wkey}.weight" in P else wkey.rsplit(".", 1)[0]


def unet_module(x, P: Params, pre: str, scale: int, is_deconv: bool, taps=None, bn: bool = False, add: bool = False):
    """En_unet_module.forward (EaBNet.py:372-388); Skip_connect (:493-503) cat or add."""
    wk = _gk(P, f"{pre}.in_conv.0.conv.{0 if is_deconv else 1}")
    gated = (gate_deconv2d if is_deconv else gate_conv2d)(x, P[f"{wk}.weight"], P[f"{wk}.bias"])
    resi = _in_prelu(gated, P, f"{pre}.in_conv.1", f"{pre}.in_conv.2", bn)
    if taps is not None:
        taps[f"{pre}.in_conv"] = resi
    y = resi
    downs: List[torch.Tensor] = []
    for j in range(scale):
        q = f"{pre}.enco.{j}.conv"
        y = _in_prelu(F.conv2d(y, P[f"{q}.0.weight"], P[f"{q}.0.bias"], stride=(1, 2)), P, f"{q}.1", f"{q}.2", bn)
        downs.append(y)
    for j in range(scale):
        q = f"{pre}.deco.{j}.deconv"
        if j > 0:
            y = y + downs[-(j + 1)] if add else torch.cat((y, downs[-(j + 1)]), dim=1)
        y = _in_prelu(F.conv_transpose2d(y, P[f"{q}.0.weight"], P[f"{q}.0.bias"], stride=(1, 2)),
                      P, f"{q}.1", f"{q}.2", bn)
    return resi + y


def squeezed_tcm(x, P: Params, pre: str, dilation: int, kd: int, bn: bool = False, causal: bool = True):
    """SqueezedTCM.forward (EaBNet.py:572-578); padding of the dilated branch convs :550-553
    (all on the left when causal, split evenly otherwise)."""
    y = F.conv1d(x, P[f"{pre}.in_conv.weight"])
    span = (kd - 1) * dilation
    pad = (span, 0) if causal else (span // 2, span // 2)

    def branch(side):
        z = F.prelu(y, P[f"{pre}.{side}.0.weight"])
        z = _norm(z, P, f"{pre}.{side}.1", bn)
        return F.conv1d(F.pad(z, pad), P[f"{pre}.{side}.3.weight"], dilation=dilation)

    z = branch("left_conv") * torch.sigmoid(branch("right_conv"))
    z = F.prelu(z, P[f"{pre}.out_conv.0.weight"])
    z = _norm(z, P, f"{pre}.out_conv.1", bn)
    return F.conv1d(z, P[f"{pre}.out_conv.2.weight"]) + x


def lstm_layer(x, w_ih, w_hh, b_ih, b_hh, fast: bool = False):
    """One batch_first nn.LSTM layer with zero initial state (EaBNet.py:591-592,
    610-611).  x: (N, T, I) -> (N, T, H).  Gate order i, f, g, o."""
    N, T, _ = x.shape
    H = w_hh.shape[1]
    if fast:
        zeros = x.new_zeros(1, N, H)
        out, _, _ = torch._VF.lstm(x, (zeros, zeros), [w_ih, w_hh, b_ih, b_hh], True, 1, 0.0, False, False, True)
        return out
    pre = x @ w_ih.t() + (b_ih + b_hh)                 # (N, T, 4H)
    h = x.new_zeros(N, H)
    c = x.new_zeros(N, H)
    outs = []
    for t in range(T):
        g = pre[:, t] + h @ w_hh.t()
        i, f, gg, o = g.chunk(4, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        outs.append(h)
    return torch.stack(outs, dim=1)


def lstm_bf(e, P: Params, M: int, fast: bool = False, taps=None):
    """LSTM_BF.forward (EaBNet.py:600-614): e (B,C,T,F) -> weights (B,T,F,M,2)."""
    B, C, T, Fq = e.shape
    x = F.layer_norm(e.permute(0, 3, 2, 1).contiguous(), (C,), P["bf_map.norm.weight"], P["bf_map.norm.bias"], EPS_LN)
    x = x.view(B * Fq, T, C)
    for name in ("rnn1", "rnn2"):
        x = lstm_layer(x, P[f"bf_map.{name}.weight_ih_l0"], P[f"bf_map.{name}.weight_hh_l0"],
                       P[f"bf_map.{name}.bias_ih_l0"], P[f"bf_map.{name}.bias_hh_l0"], fast)
        if taps is not None:
            taps[f"bf_map.{name}"] = x.view(B, Fq, T, -1)
    x = x.view(B, Fq, T, -1).transpose(1, 2).contiguous()
    x = F.relu(F.linear(x, P["bf_map.w_dnn.0.weight"], P["bf_map.w_dnn.0.bias"]))
    w = F.linear(x, P["bf_map.w_dnn.2.weight"], P["bf_map.w_dnn.2.bias"])
    return w.view(B, T, Fq, M, 2)


def filter_and_sum(w, x):
    """Y = sum_m W_m * X_m, complex, no conjugate (EaBNet.py:114-117).
    w, x: (B,T,F,M,2) -> (B,2,T,F)."""
    wr, wi = w[..., 0], w[..., 1]
    xr, xi = x[..., 0], x[..., 1]
    return torch.stack(((wr * xr - wi * xi).sum(-1), (wr * xi + wi * xr).sum(-1)), dim=1)


def unet_encoder(x, P: Params, bn: bool):
    """UNet_Encoder.forward (EaBNet.py:234-239): five gated convs; layers 1 and 2 carry a PReLU
    but no norm (:217-224)."""
    skips = []
    for i in range(5):
        x = gate_conv2d(x, P[_gk(P, f"en.unet_list.{i}.0.conv.1") + ".weight"], P[_gk(P, f"en.unet_list.{i}.0.conv.1") + ".bias"])
        if i in (1, 2):
            x = F.prelu(x, P[f"en.unet_list.{i}.1.weight"])
        else:
            x = _in_prelu(x, P, f"en.unet_list.{i}.1", f"en.unet_list.{i}.2", bn)
        skips.append(x)
    return x, skips


def unet_decoder(x, skips, P: Params, bn: bool):
    """UNet_Decoder.forward (EaBNet.py:324-328)."""
    for i in range(5):
        x = gate_deconv2d(torch.cat((x, skips[-(i + 1)]), dim=1), P[_gk(P, f"de.unet_list.{i}.0.conv.0") + ".weight"],
                          P[_gk(P, f"de.unet_list.{i}.0.conv.0") + ".bias"])
        x = _in_prelu(x, P, f"de.unet_list.{i}.1", f"de.unet_list.{i}.2", bn)
    return x


def eabnet_forward(P: Params, inpt: torch.Tensor, p: int = 6, q: int = 3, kd: int = 5,
                   fast_lstm: bool = False, taps: Optional[dict] = None, *, is_causal: bool = True,
                   is_u2: bool = True, bf_type: str = "lstm", topo_type: str = "mimo",
                   intra_connect: str = "cat", norm_type: str = "IN", bn_train: bool = False,
                   k1: Optional[tuple] = None, k2: Optional[tuple] = None) -> torch.Tensor:
    """EaBNet.forward (EaBNet.py:88-125).
    inpt (B,T,F,M,2) [or (B,T,F,2)] -> (B,2,T,F)  [(B,2,T) for topo_type="miso", as the reference].
    bn_train: norm_type="BN" with the module in train mode (batch statistics; see _norm).
    k1 / k2: the constructor's kernel sizes -- the convolutions take their extents (and the causal pad / chomp of k_t - 1 rows,
    EaBNet.py:447-452,477-482) from the weight shapes, so these are only checked against the parameters."""
    if inpt.ndim == 4:
        inpt = inpt.unsqueeze(-2)
    B, T, Fq, M, _ = inpt.shape
    for pre, want in (("de.", k1), ("en.", k2)):
        if want is not None:
            key = next((k for k in P if k.startswith(pre) and (".enco.0.conv.0.weight" if pre == "en." else ".in_conv.0.conv.0.weight") in k), None)
            assert key is None or tuple(P[key].shape[2:]) == tuple(want), (key, tuple(P[key].shape), want)
    assert norm_type in ("IN", "BN", "cLN") and intra_connect in ("cat", "add")
    bn, add = {"BN": "train" if bn_train else True, "IN": False, "cLN": "cLN"}[norm_type], intra_connect == "add"
    # (B,T,F,M,2) -> (B,2M,T,F), channel = ri*M + m   (EaBNet.py:96-97)
    x = inpt.transpose(-2, -1).contiguous().view(B, T, Fq, 2 * M).permute(0, 3, 1, 2)

    skips = []
    if not is_u2:
        x, skips = unet_encoder(x, P, bn)
    else:
        for i in range(4):
            x = unet_module(x, P, f"en.meta_unet_list.{i}", 4 - i, False, taps, bn, add)
            skips.append(x)
            if taps is not None:
                taps[f"en.{i}"] = x
        x = gate_conv2d(x, P[_gk(P, "en.last_conv.0.conv.1") + ".weight"], P[_gk(P, "en.last_conv.0.conv.1") + ".bias"])
        x = _in_prelu(x, P, "en.last_conv.1", "en.last_conv.2", bn)
        skips.append(x)
    if taps is not None:
        taps["en.4"] = x

    C = x.shape[1]
    x = x.transpose(-2, -1).contiguous().view(B, -1, T)           # (B, C*4, T), ch = c*4 + f
    acc = torch.zeros_like(x)
    for g in range(q):
        for i in range(p):
            x = squeezed_tcm(x, P, f"stcns.{g}.tcm_list.{i}", 2 ** i, kd, bn, is_causal)
            if taps is not None and g == 0 and i == 0:
                taps["stcns.0.0"] = x
        acc = acc + x
    x = acc.view(B, C, -1, T).transpose(-2, -1).contiguous()      # (B,C,T,4)
    if taps is not None:
        taps["stcns"] = x

    if not is_u2:
        x = unet_decoder(x, skips, P, bn)
    else:
        for i in range(4):
            x = unet_module(torch.cat((x, skips[-(i + 1)]), dim=1), P, f"de.meta_unet_list.{i}", i + 1, True, taps,
                            bn, add)
            if taps is not None:
                taps[f"de.{i}"] = x
        x = gate_deconv2d(torch.cat((x, skips[0]), dim=1), P[_gk(P, "de.last_conv.0.conv.0") + ".weight"],
                          P[_gk(P, "de.last_conv.0.conv.0") + ".bias"])
        x = _in_prelu(x, P, "de.last_conv.1", "de.last_conv.2", bn)
    if taps is not None:
        taps["de.4"] = x

    if topo_type == "miso":
        # EaBNet.py:118-125: one complex mask on microphone 0, then -- as written there -- a sum over
        # the LAST axis of a (B,T,F) tensor, i.e. over frequency: the result is (B,2,T)
        m = F.conv2d(x, P["bf_map.weight"], P["bf_map.bias"]).permute(0, 2, 3, 1)           # (B,T,F,2)
        mr, mi, xr, xi = m[..., 0], m[..., -1], inpt[..., 0, 0], inpt[..., 0, -1]
        return torch.stack(((mr * xr - mi * xi).sum(-1), (mr * xi + mi * xr).sum(-1)), dim=1)
    if bf_type == "cnn":
        # EaBNet.py:111-113: pointwise conv to 2M planes, plane m*2+ri
        w = F.conv2d(x, P["bf_map.weight"], P["bf_map.bias"]).view(B, M, -1, T, Fq).permute(0, 3, 4, 1, 2)
    else:
        w = lstm_bf(x, P, M, fast_lstm, taps)
    if taps is not None:
        taps["bf_w"] = w
    return filter_and_sum(w, inpt)


def com_mag_mse_loss(esti: torch.Tensor, label: torch.Tensor, frame_list) -> torch.Tensor:
    """EaBNet.py:627-640 with per-utterance frame masks."""
    B, _, T, Fq = esti.shape
    mask = torch.zeros(B, T, Fq, dtype=esti.dtype)
    for i, n in enumerate(frame_list):
        mask[i, :n] = 1.0
    mag_e, mag_l = torch.norm(esti, dim=1), torch.norm(label, dim=1)
    l1 = (((mag_e - mag_l) ** 2) * mask).sum() / mask.sum()
    l2 = (((esti - label) ** 2) * mask.unsqueeze(1)).sum() / (2.0 * mask.sum())
    return 0.5 * (l1 + l2)


# ----------------------------------------------------------------------------
# GaGNet post-filter (reference GaGNet.py) and the two-stage wrapper (EaBNet.py:127-148)
# ----------------------------------------------------------------------------
def gag_tcm(x, P: Params, pre: str, dilation: int, kd: int, bn: bool, causal: bool):
    """GaGNet's single-branch SqueezedTCM.forward (GaGNet.py:321-327)."""
    span = (kd - 1) * dilation
    pad = (span, 0) if causal else (span // 2, span // 2)
    y = F.conv1d(x, P[f"{pre}.in_conv.weight"])
    y = _norm(F.prelu(y, P[f"{pre}.d_conv.0.weight"]), P, f"{pre}.d_conv.1", bn)
    y = F.conv1d(F.pad(y, pad), P[f"{pre}.d_conv.3.weight"], dilation=dilation)
    y = _norm(F.prelu(y, P[f"{pre}.out_conv.0.weight"]), P, f"{pre}.out_conv.1", bn)
    return F.conv1d(y, P[f"{pre}.out_conv.2.weight"]) + x


def gag_chain(x, P: Params, pre: str, p: int, dilas, kd, bn, causal):
    for j in range(p):
        for k, d in enumerate(dilas):
            x = gag_tcm(x, P, f"{pre}.{j}.tcns.{k}", d, kd, bn, causal)
    return x


def gagnet_forward(P: Params, inpt: torch.Tensor, pre_x: torch.Tensor, *, kd1: int = 3, p: int = 2, q: int = 3,
                   dilas=(1, 2, 5, 9), is_u2: bool = True, is_causal: bool = True, is_squeezed: bool = False,
                   acti_type: str = "sigmoid", intra_connect: str = "cat", norm_type: str = "IN",
                   bn_train: bool = False, k1: Optional[tuple] = None) -> List[torch.Tensor]:
    """GaGNet.forward (GaGNet.py:76-90): inpt, pre_x (B,2,T,F) -> q stage outputs (B,2,F,T).
    bn_train: norm_type="BN" with the module in train mode (batch statistics; see _norm).
    k1: the constructor's gated-kernel size; the convolutions take it from the weight shapes, so it is only checked."""
    B, _, T, Fq = inpt.shape
    if k1 is not None and "en.last_conv.0.conv.1.weight" in P:
        assert tuple(P[_gk(P, "en.last_conv.0.conv.1") + ".weight"].shape[2:]) == tuple(k1)
    bn, add = ("train" if bn_train else True) if norm_type == "BN" else False, intra_connect == "add"
    x = torch.cat([inpt, pre_x], dim=1)
    if is_u2:
        for i in range(4):
            x = unet_module(x, P, f"en.meta_unet_list.{i}", 4 - i, False, None, bn, add)
        x = _in_prelu(gate_conv2d(x, P[_gk(P, "en.last_conv.0.conv.1") + ".weight"], P[_gk(P, "en.last_conv.0.conv.1") + ".bias"]), P,
                      "en.last_conv.1", "en.last_conv.2", bn)
    else:
        for i in range(5):
            x = _in_prelu(gate_conv2d(x, P[_gk(P, f"en.unet_list.{i}.0.conv.1") + ".weight"], P[_gk(P, f"en.unet_list.{i}.0.conv.1") + ".bias"]), P,
                          f"en.unet_list.{i}.1", f"en.unet_list.{i}.2", bn)
    feat = x.transpose(-2, -1).contiguous().view(B, -1, T)
    pre = pre_x.transpose(-2, -1).contiguous()                       # (B,2,F,T)
    act = {"sigmoid": torch.sigmoid, "tanh": torch.tanh, "relu": torch.relu}[acti_type]
    outs = []
    for g in range(q):
        gl, gz = f"gags.{g}.glance_block", f"gags.{g}.gaze_block"
        cat = torch.cat((feat, pre.view(B, -1, T)), dim=1)

        def gated(pfx):
            return F.conv1d(cat, P[f"{pfx}.in_conv_main.weight"], P[f"{pfx}.in_conv_main.bias"]) * torch.sigmoid(
                F.conv1d(cat, P[f"{pfx}.in_conv_gate.0.weight"], P[f"{pfx}.in_conv_gate.0.bias"]))

        xg = gag_chain(gated(gl), P, f"{gl}.tcn_g", p, dilas, kd1, bn, is_causal)
        gain = act(F.conv1d(xg, P[f"{gl}.linear_g.0.weight"], P[f"{gl}.linear_g.0.bias"]))          # (B,F,T)
        xz = gated(gz)
        if is_squeezed:
            xr = xi = gag_chain(xz, P, f"{gz}.tcm_ri", p, dilas, kd1, bn, is_causal)
        else:
            xr = gag_chain(xz, P, f"{gz}.tcm_r", p, dilas, kd1, bn, is_causal)
            xi = gag_chain(xz, P, f"{gz}.tcm_i", p, dilas, kd1, bn, is_causal)
        resi = torch.stack((F.conv1d(xr, P[f"{gz}.linear_r.weight"], P[f"{gz}.linear_r.bias"]),
                            F.conv1d(xi, P[f"{gz}.linear_i.weight"], P[f"{gz}.linear_i.bias"])), dim=1)
        # GaGNet.py:128-133: polar form of "previous estimate times gain" plus the complex residual
        mag, ph = torch.norm(pre, dim=1), torch.atan2(pre[:, -1], pre[:, 0])
        filt = mag * gain
        pre = torch.stack((filt * torch.cos(ph), filt * torch.sin(ph)), dim=1) + resi
        outs.append(pre)
    return outs


def eabnet_postnet_forward(P: Params, noisy: torch.Tensor, ref_mic: int = 0, eab_kw: Optional[dict] = None,
                           gag_kw: Optional[dict] = None, fast_lstm: bool = False) -> dict:
    """EaBNetWithPostNet.forward (EaBNet.py:138-148); P holds ``eabnet.*`` and ``postnet.*`` keys."""
    Pe = {k[len("eabnet."):]: v for k, v in P.items() if k.startswith("eabnet.")}
    Pg = {k[len("postnet."):]: v for k, v in P.items() if k.startswith("postnet.")}
    esti0 = eabnet_forward(Pe, noisy, fast_lstm=fast_lstm, **(eab_kw or {}))
    inpt = noisy[..., ref_mic, :].permute(0, 3, 1, 2)
    lst = gagnet_forward(Pg, inpt, esti0, **(gag_kw or {}))
    return {"esti0_stft": esti0, "esti1_stft_list": lst, "esti_stft": lst[-1].permute(0, 1, 3, 2)}


def stagewise_com_mag_mse_loss(esti_list, label, frame_list) -> torch.Tensor:
    """GaGNet.py:601-619: stage weights 0.1 (last stage 1); esti (B,2,F,T), label (B,2,F,T)."""
    B, _, Fq, T = label.shape
    mask = torch.zeros(B, Fq, T, dtype=label.dtype)
    for i, n in enumerate(frame_list):
        mask[i, :, :n] = 1.0
    l1 = l2 = 0.0
    mag_l = torch.norm(label, dim=1)
    for i, e in enumerate(esti_list):
        a = 1.0 if i == len(esti_list) - 1 else 0.1
        l1 = l1 + a * (((e - label) ** 2) * mask.unsqueeze(1)).sum() / (2.0 * mask.sum())
        l2 = l2 + a * (((torch.norm(e, dim=1) - mag_l) ** 2) * mask).sum() / mask.sum()
    return 0.5 * (l1 + l2)
