"""The oracle (oracle/eabnet_oracle.py) against fixtures produced by the
reference itself (tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

import paramgen
from eabnet_amd.spec import NetConfig, param_specs
from oracle import eabnet_oracle as orc
from util import GOLDEN, TOL_ORACLE, assert_close, assert_compressed_close, load, torch_params


@pytest.mark.parametrize("M", [8, 9])
def test_spec_matches_reference_state_dict(M):
    ref = json.load(open(os.path.join(GOLDEN, f"keys_M{M}.json")))
    specs = param_specs(NetConfig(M=M))
    assert [k for k, _ in ref] == list(specs.keys())
    for k, shape in ref:
        assert tuple(shape) == tuple(specs[k].shape), k
    assert len(ref) == 498
    if M == 8:
        assert sum(int(np.prod(s)) for _, s in ref) == 2_835_920


@pytest.mark.parametrize("name", ["stft_B1_M2_L1600.npz", "stft_B2_M8_L4000.npz", "stft_B1_M3_L2085.npz"])
def test_prepare_data_oracle(name):
    g = load(name)
    B, T, F, M, _ = g["noisy"].shape
    L = int(name.split("_L")[1].split(".")[0])
    x = torch.from_numpy(paramgen.make_wave(B, M, L, int(g["seed"])))
    noisy, tgt = orc.prepare_data_oracle(x, x[:, :1])
    assert T == 1 + L // 160 and F == 161
    assert_compressed_close(noisy.numpy(), g["noisy"], TOL_ORACLE, "noisy")
    assert_compressed_close(np.moveaxis(tgt.numpy(), 1, -1), np.moveaxis(g["target"], 1, -1), TOL_ORACLE, "target")


def test_stft_frames_bit_exact_vs_torch_stft():
    """Frame indexing must be bit-exact: un-windowed frames through a
    rectangular-window torch.stft equal rfft of the oracle's gathered frames, and
    the gather equals an explicit reflect pad."""
    x = torch.from_numpy(paramgen.make_wave(1, 2, 2085, 5))[0]
    fr = orc.stft_frames(x, 320, 160)
    padded = torch.nn.functional.pad(x.unsqueeze(0), (160, 160), mode="reflect")[0]
    T = 1 + 2085 // 160
    for t in range(T):
        assert torch.equal(fr[:, t], padded[:, t * 160:t * 160 + 320])
    k = torch.arange(320, dtype=torch.float64)
    assert (orc.hann_periodic(320).double() - (0.5 - 0.5 * torch.cos(2 * torch.pi * k / 320))).abs().max() < 3e-7


def test_zero_mic_maps_to_zero():
    g = load("stft_zero_mic.npz")
    x = torch.from_numpy(paramgen.make_wave(1, 2, 1600, 3)); x[:, 1] = 0.0
    noisy, _ = orc.prepare_data_oracle(x, None)
    assert torch.count_nonzero(noisy[..., 1, :]) == 0
    assert np.count_nonzero(g["noisy"][..., 1, :]) == 0
    assert_compressed_close(noisy.numpy(), g["noisy"], TOL_ORACLE)


def test_e2e_taps():
    g = load("e2e_M8_B1_T12_taps.npz")
    P = torch_params(8, int(g["param_seed"]))
    x = torch.from_numpy(paramgen.make_spec_input(1, 12, 161, 8, int(g["input_seed"])))
    taps = {}
    with torch.no_grad():
        y = orc.eabnet_forward(P, x, taps=taps)
    for k in g.files:
        if not k.startswith("tap/"):
            continue
        name = k[4:]
        if name in ("rnn1", "rnn2"):
            mine = taps[f"bf_map.{name}"].reshape(g[k].shape)
        elif name == "stcns.0.0":
            mine = taps[name]
        else:
            mine = taps[name]
        assert_close(mine.numpy(), g[k], TOL_ORACLE, name)
    assert_close(y.numpy(), g["out"], TOL_ORACLE, "out")


def test_e2e_batch2_and_loss():
    g = load("e2e_M8_B2_T20.npz")
    P = torch_params(8, int(g["param_seed"]))
    x = torch.from_numpy(paramgen.make_spec_input(2, 20, 161, 8, int(g["input_seed"])))
    with torch.no_grad():
        y = orc.eabnet_forward(P, x)
        y_fast = orc.eabnet_forward(P, x, fast_lstm=True)
    assert_close(y.numpy(), g["out"], TOL_ORACLE)
    assert_close(y_fast.numpy(), g["out"], TOL_ORACLE)
    label = torch.from_numpy(paramgen.make_spec_input(2, 20, 161, 1, int(g["label_seed"]))[..., 0, :]).permute(0, 3, 1, 2)
    assert abs(float(orc.com_mag_mse_loss(y, label, [20, 20])) - float(g["loss_full"])) < 1e-5 * float(g["loss_full"])
    assert abs(float(orc.com_mag_mse_loss(y, label, [20, 13])) - float(g["loss_ragged"])) < 1e-5 * float(g["loss_ragged"])


@pytest.mark.parametrize("M,name", [(1, "e2e_M1_B1_T10.npz"), (9, "e2e_M9_B1_T10.npz")])
def test_e2e_other_mic_counts(M, name):
    g = load(name)
    P = torch_params(M, int(g["param_seed"]))
    x = torch.from_numpy(paramgen.make_spec_input(1, 10, 161, M, int(g["input_seed"])))
    with torch.no_grad():
        y = orc.eabnet_forward(P, x)
        if M == 1:
            assert torch.equal(orc.eabnet_forward(P, x[..., 0, :]), y)     # 4-D input path
    assert_close(y.numpy(), g["out"], TOL_ORACLE)


def test_c1_full_size():
    g = load("c1_M8_T401.npz")
    P = torch_params(8, int(g["param_seed"]))
    wav = torch.from_numpy(paramgen.make_wave(1, 8, 64000, int(g["wave_seed"])))
    with torch.no_grad():
        ns, ts = orc.prepare_data_oracle(wav, wav[:, :1])
        y = orc.eabnet_forward(P, ns, fast_lstm=True)
    assert ns.shape == (1, 401, 161, 8, 2)
    assert_compressed_close(ns[:, g["stft_probe_t"].tolist()].numpy(), g["stft_probe"], TOL_ORACLE)
    assert abs(float(torch.linalg.vector_norm(ns.double())) - float(g["stft_l2"])) < 1e-6 * float(g["stft_l2"])
    assert abs(float(torch.linalg.vector_norm(ts.double())) - float(g["target_l2"])) < 1e-6 * float(g["target_l2"])
    assert_close(y.numpy(), g["out"], TOL_ORACLE)


def _c_oracle():
    import ctypes
    import subprocess
    root = os.path.dirname(GOLDEN.rstrip("/")).rsplit("/tests", 1)[0]
    subprocess.run(["make", "-C", os.path.join(root, "oracle")], check=True, capture_output=True)
    lib = ctypes.CDLL(os.path.join(root, "oracle", "build", "libfrontend_ref.so"))
    lib.oracle_stft_compress.restype = ctypes.c_int
    lib.oracle_stft_compress.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_int, ctypes.c_long,
                                                                 ctypes.c_int, ctypes.c_int]
    lib.oracle_filter_sum.restype = None
    lib.oracle_filter_sum.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_long, ctypes.c_int]
    return lib


@pytest.mark.parametrize("name", ["stft_B1_M2_L1600.npz", "stft_B1_M3_L2085.npz", "stft_zero_mic.npz"])
def test_plain_c_front_end_vs_reference_fixtures(name):
    """oracle/frontend_ref.c (O(N^2) double-precision DFT, no PyTorch) against the reference fixtures."""
    lib = _c_oracle()
    g = load(name)
    B, T, F, M, _ = g["noisy"].shape
    L = 1600 if "zero" in name else int(name.split("_L")[1].split(".")[0])
    x = paramgen.make_wave(B, M, L, int(g["seed"]))
    if "zero" in name:
        x[:, 1] = 0.0
    x = np.ascontiguousarray(x)
    win = torch.hann_window(320).numpy().copy()
    out = np.empty((B, T, F, M, 2), dtype=np.float32)
    assert lib.oracle_stft_compress(x.ctypes.data, win.ctypes.data, out.ctypes.data, B, M, L, 320, 160) == 0
    assert_compressed_close(out, g["noisy"], TOL_ORACLE)
    if "zero" in name:
        assert np.count_nonzero(out[..., 1, :]) == 0


def test_plain_c_filter_sum_vs_oracle():
    lib = _c_oracle()
    rng = np.random.default_rng(3)
    w = rng.standard_normal((2, 5, 161, 8, 2)).astype(np.float32)
    x = rng.standard_normal((2, 5, 161, 8, 2)).astype(np.float32)
    yr = np.empty(2 * 5 * 161, np.float32)
    yi = np.empty_like(yr)
    lib.oracle_filter_sum(w.ctypes.data, x.ctypes.data, yr.ctypes.data, yi.ctypes.data, yr.size, 8)
    ref = orc.filter_and_sum(torch.from_numpy(w), torch.from_numpy(x)).numpy()
    assert_close(yr.reshape(2, 5, 161), ref[:, 0], 1e-6)
    assert_close(yi.reshape(2, 5, 161), ref[:, 1], 1e-6)


def _variants():
    import json
    import os
    with open(os.path.join(GOLDEN, "keys_variants.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("name", sorted(_variants()))
def test_oracle_constructor_variants_match_reference(name):
    """The oracle's keyword branches (BN eval, plain U-Net, cnn / miso heads, add skips,
    non-causal S-TCMs) against outputs of the reference constructed with the same keywords."""
    e = _variants()[name]
    g = load(f"var_{name}.npz")
    P = torch_params(e["M"], int(g["param_seed"]), **e["kwargs"])
    x = torch.from_numpy(paramgen.make_spec_input(2, 20, 161, e["M"], int(g["input_seed"])))
    with torch.no_grad():
        y = orc.eabnet_forward(P, x, **e["kwargs"])
    assert tuple(y.shape) == tuple(g["out"].shape)
    assert_close(y.numpy(), g["out"], TOL_ORACLE, name)


def test_oracle_batchnorm_train_mode_matches_reference():
    """norm_type="BN" in TRAIN mode (what the reference's trainer runs): the oracle's batch-statistics branch and its
    running-buffer update against the reference module's forward and state after one step (tests/golden/bn_train.npz)."""
    g = load("bn_train.npz")
    kw = dict(norm_type="BN", p=int(g["p"]), q=int(g["q"]))
    M = int(g["M"])
    P = torch_params(M, int(g["param_seed"]), **kw)
    P["__bn_updates__"] = {}
    x = torch.from_numpy(paramgen.make_spec_input(int(g["B"]), int(g["T"]), 161, M, int(g["input_seed"])))
    with torch.no_grad():
        y = orc.eabnet_forward(P, x, bn_train=True, **kw)
    assert_close(y.numpy(), g["out"], TOL_ORACLE, "train-mode BatchNorm forward")
    upd = P["__bn_updates__"]
    assert len(upd) == sum(1 for k in g.files if k.endswith("running_mean")) > 50
    for k, (rm, rv) in upd.items():
        np.testing.assert_allclose(rm.numpy(), g[f"{k}.norm.running_mean"], rtol=1e-5, atol=1e-6, err_msg=k)
        np.testing.assert_allclose(rv.numpy(), g[f"{k}.norm.running_var"], rtol=1e-5, atol=1e-6, err_msg=k)
        assert int(g[f"{k}.norm.num_batches_tracked"]) == int(P[f"{k}.norm.num_batches_tracked"]) + 1


@pytest.mark.parametrize("B,T", [(1, 2), (2, 9), (3, 40)])
def test_istft_oracle_matches_reference_call(B, T):
    """Back end: the spelled-out irfft / window / overlap-add / envelope / trim
    against torch.istft called exactly as enhance.py:59-62 does."""
    g = load(f"istft_B{B}_T{T}.npz")
    esti = torch.from_numpy(paramgen.make_spec_input(B, T, 161, 1, int(g["seed"]))[..., 0, :]).permute(0, 3, 1, 2)
    wav = orc.istft_oracle(esti.contiguous())
    assert wav.shape == (B, 160 * (T - 1))
    assert_close(wav.numpy(), g["wav"], TOL_ORACLE, "istft")


def _gag_variants():
    import json
    import os
    with open(os.path.join(GOLDEN, "keys_gagnet.json")) as f:
        return json.load(f)


def _gag_case(name):
    from eabnet_amd.spec import GagConfig, gag_param_specs
    e = _gag_variants()[name]
    g = load(f"gag_{name}.npz")
    kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in e["kwargs"].items()}
    cfg = GagConfig(**kw)
    P = {k: torch.from_numpy(v) for k, v in paramgen.make_params(gag_param_specs(cfg), int(g["param_seed"])).items()}
    mk = lambda seed: torch.from_numpy(paramgen.make_spec_input(2, 14, 161, 1, seed)[..., 0, :]).permute(0, 3, 1, 2).contiguous()  # noqa: E731
    return cfg, kw, P, mk(int(g["inpt_seed"])), mk(int(g["pre_seed"])), g, e


@pytest.mark.parametrize("name", sorted(_gag_variants()))
def test_gagnet_oracle_matches_reference(name):
    """Post-filter: functional GaGNet against the reference class run on CPU
    (default topology; BN + squeezed + tanh; plain encoder + add + non-causal + relu)."""
    from eabnet_amd.spec import gag_param_specs
    cfg, kw, P, inpt, pre, g, e = _gag_case(name)
    assert [[k, list(s.shape)] for k, s in gag_param_specs(cfg).items()] == e["keys"]
    with torch.no_grad():
        outs = orc.gagnet_forward(P, inpt, pre, **kw)
    assert len(outs) == cfg.q
    for j, o in enumerate(outs):
        assert_close(o.numpy(), g[f"out{j}"], TOL_ORACLE, f"{name} stage {j}")
    if name == "default":
        label = torch.from_numpy(paramgen.make_spec_input(2, 14, 161, 1, 800)[..., 0, :]).permute(0, 3, 2, 1).contiguous()
        assert abs(float(orc.stagewise_com_mag_mse_loss(outs, label, [14, 9])) - float(g["stage_loss"])) < 1e-5 * float(g["stage_loss"])


def test_two_stage_oracle_matches_reference_composition():
    """EaBNetWithPostNet.forward (EaBNet.py:138-148): beam-former -> post-filter on (reference mic,
    estimate).  The post-filter amplifies a 3e-7 difference of its input ~100x (random hot weights,
    12 frames), hence the looser bar on the final stage; the beam-former stage is at the oracle bar."""
    from eabnet_amd.spec import GagConfig, gag_param_specs
    g = load("postnet_M4_T12.npz")
    P = {"eabnet." + k: v for k, v in torch_params(4, int(g["eab_seed"])).items()}
    P.update({"postnet." + k: torch.from_numpy(v) for k, v in
              paramgen.make_params(gag_param_specs(GagConfig()), int(g["gag_seed"])).items()})
    noisy = torch.from_numpy(paramgen.make_spec_input(1, 12, 161, 4, int(g["input_seed"])))
    with torch.no_grad():
        o = orc.eabnet_postnet_forward(P, noisy, ref_mic=int(g["ref_mic"]))
    assert_close(o["esti0_stft"].numpy(), g["esti0"], TOL_ORACLE, "esti0")
    assert o["esti_stft"].shape == (1, 2, 12, 161)
    assert_close(o["esti_stft"].numpy(), g["esti"], 2e-4, "esti")


def _block(name):
    """inputs, reference output and regenerated parameters of one stand-alone reference block
    (tests/golden/blocks.npz + blocks_params.json, make_golden.py main_blocks)"""
    import json
    import os
    g = load("blocks.npz")
    with open(os.path.join(GOLDEN, "blocks_params.json")) as f:
        recipe = json.load(f)[name]
    P = {k: torch.from_numpy(paramgen.make_param("block/" + k, tuple(shape), kind, fan, seed)) for k, shape, kind, fan, seed in recipe}
    return torch.from_numpy(g[f"{name}/x"]), g[f"{name}/y"], P


def test_oracle_blocks_match_the_reference_classes():
    """Every block function of the oracle against the reference's own class run alone:
    GateConv2d, GateConvTranspose2d (+Chomp_T), Conv2dunit, Deconv2dunit (add / cat), En_unet_module in both
    directions, SqueezedTCM at dilation 1 and 32 (longer than the 40-frame input), LSTM_BF."""
    with torch.no_grad():
        for name in ("gateconv_2x5", "gateconv_2x3"):
            x, y, P = _block(name)
            assert_close(orc.gate_conv2d(x, P["conv.1.weight"], P["conv.1.bias"]).numpy(), y, TOL_ORACLE, name)
        for name in ("gatedeconv_2x3", "gatedeconv_2x5"):
            x, y, P = _block(name)
            assert_close(orc.gate_deconv2d(x, P["conv.0.weight"], P["conv.0.bias"]).numpy(), y, TOL_ORACLE, name)
        x, y, P = _block("conv2dunit")
        Pq = {f"u.{k}": v for k, v in P.items()}
        out = orc._in_prelu(torch.nn.functional.conv2d(x, P["conv.0.weight"], P["conv.0.bias"], stride=(1, 2)), Pq, "u.conv.1", "u.conv.2")
        assert_close(out.numpy(), y, TOL_ORACLE, "conv2dunit")
        for name in ("deconv2dunit_add", "deconv2dunit_cat"):
            x, y, P = _block(name)
            Pq = {f"u.{k}": v for k, v in P.items()}
            out = orc._in_prelu(torch.nn.functional.conv_transpose2d(x, P["deconv.0.weight"], P["deconv.0.bias"], stride=(1, 2)),
                                Pq, "u.deconv.1", "u.deconv.2")
            assert_close(out.numpy(), y, TOL_ORACLE, name)
        x, y, P = _block("unet_module_enc_s3")
        assert_close(orc.unet_module(x, {f"m.{k}": v for k, v in P.items()}, "m", 3, False).numpy(), y, TOL_ORACLE, "enc module")
        x, y, P = _block("unet_module_dec_s2")
        assert_close(orc.unet_module(x, {f"m.{k}": v for k, v in P.items()}, "m", 2, True).numpy(), y, TOL_ORACLE, "dec module")
        for name, d in (("stcm_d1", 1), ("stcm_d32", 32)):
            x, y, P = _block(name)
            assert_close(orc.squeezed_tcm(x, {f"t.{k}": v for k, v in P.items()}, "t", d, 5).numpy(), y, TOL_ORACLE, name)
        x, y, P = _block("lstm_bf")
        assert_close(orc.lstm_bf(x, {f"bf_map.{k}": v for k, v in P.items()}, 8).numpy(), y, TOL_ORACLE, "lstm_bf")


def test_cumulative_layer_norm_matches_the_reference_classes():
    """oracle.cumulative_layer_norm against the outputs of the reference's own CumulativeLayerNorm1d / 2d
    (EaBNet.py:696-769) -- the norm NormSwitch(norm_type="cLN") means to build (fixture: cln_classes.npz)."""
    g = load("cln_classes.npz")
    for name in ("1d", "2d"):
        y = orc.cumulative_layer_norm(torch.from_numpy(g[f"{name}/x"]), torch.from_numpy(g[f"{name}/gain"]), torch.from_numpy(g[f"{name}/bias"]))
        assert_close(y.numpy(), g[f"{name}/y"], TOL_ORACLE, name)
    # causal: changing later frames never changes earlier outputs
    x = torch.from_numpy(g["2d/x"]).clone()
    y0 = orc.cumulative_layer_norm(x, torch.from_numpy(g["2d/gain"]), torch.from_numpy(g["2d/bias"]))
    x[:, :, 7:] += 1.0
    y1 = orc.cumulative_layer_norm(x, torch.from_numpy(g["2d/gain"]), torch.from_numpy(g["2d/bias"]))
    assert torch.equal(y0[:, :, :7], y1[:, :, :7]) and not torch.equal(y0[:, :, 7:], y1[:, :, 7:])


def test_prepare_data_oracle_with_a_short_window_equals_torch_stft():
    """win_size < fft_num (prepare_data passes both to torch.stft, train_distributed.py:83): the oracle's zero-padded
    centred window reproduces torch.stft(n_fft=320, win_length=200) bit for bit."""
    x = torch.from_numpy(paramgen.make_wave(1, 2, 2400, 17))
    win = torch.hann_window(200)
    ref = torch.stft(x.view(2, -1), 320, 160, 200, win, return_complex=False)          # (N, F, T, 2)
    got = orc.stft_oracle(x.view(2, -1), 320, 160, 200)
    assert torch.equal(got, ref) or float((got - ref).abs().max()) < 1e-6 * float(ref.abs().max())
