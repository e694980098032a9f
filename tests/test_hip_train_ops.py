"""Kernel-level parity of the training kernels (csrc/train.hip, wgrad.hip, lstm_bwd.hip) on a real MI355X, through
the C ABI, each against fp64 PyTorch autograd of the reference layer it differentiates (the layers the reference's
loss.backward() walks through: train_distributed.py:228 over EaBNet.py).  Tolerance 1e-5 relative (max-abs/max and
L2): these are single kernels on well-conditioned random data, so they sit at the fp32 rounding floor."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from util import assert_close

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    from eabnet_amd import _lib
    return _lib.load()


def _dev(t):
    return t.to("cuda:0", torch.float32).contiguous()


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ck(code, what=""):
    from eabnet_amd import _lib
    _lib.check(code, what)


def _wgrad(lib, dz, src0, src1, N, Kpad, B, T, Fin, Fz, No, ostride, ophase, istride, dt, ioff, dbias=None, precision=0):
    from eabnet_amd import _lib
    d = _lib.WgradDesc()
    dw = torch.zeros(N, Kpad, device="cuda:0")
    d.dz, d.src0, d.src1, d.dw = dz.data_ptr(), src0.data_ptr(), (src1.data_ptr() if src1 is not None else None), dw.data_ptr()
    d.dbias = dbias.data_ptr() if dbias is not None else None
    d.N, d.C0, d.C1, d.Kpad = N, src0.shape[-1], (src1.shape[-1] if src1 is not None else 0), Kpad
    d.B, d.T, d.Fin, d.Fz, d.No, d.ostride, d.ophase, d.istride = B, T, Fin, Fz, No, ostride, ophase, istride
    d.ntaps = len(dt)
    d.precision = precision
    for j in range(len(dt)):
        d.dt[j], d.ioff[j] = dt[j], ioff[j]
    _ck(lib.eab_wgrad_f32(C.byref(d), _st()), "eab_wgrad_f32")
    torch.cuda.synchronize()
    return dw.cpu()


@pytest.mark.parametrize("N,C0,C1,kt,kf,B,T,Fin", [(128, 64, 0, 2, 3, 2, 9, 19), (64, 64, 0, 1, 3, 3, 7, 39), (128, 16, 0, 2, 5, 1, 5, 161),
                                                   (256, 64, 0, 1, 1, 2, 33, 1)])
def test_wgrad_of_a_strided_convolution(lib, N, C0, C1, kt, kf, B, T, Fin):
    """dW of Conv2d((kt,kf), stride (1,2)) with causal top padding == autograd (EaBNet.py:402,450); also the 1x1 case."""
    g = torch.Generator().manual_seed(1)
    stride = 2 if kf > 1 else 1
    Fout = (Fin - kf) // stride + 1
    x = torch.randn(B, C0, T, Fin, generator=g, dtype=torch.float64)
    w = torch.randn(N, C0, kt, kf, generator=g, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(F.pad(x, (0, 0, kt - 1, 0)), w, stride=(1, stride))
    dz = torch.randn(y.shape, generator=g, dtype=torch.float64)
    (y * dz).sum().backward()
    taps = [(a, c) for a in range(kt) for c in range(kf)]
    upt = (C0 + 15) // 16
    db = torch.zeros(N, device="cuda:0")
    got = _wgrad(lib, _dev(dz.permute(0, 2, 3, 1)), _dev(x.permute(0, 2, 3, 1)), None, N, len(taps) * upt * 16, B, T, Fin, Fout, Fout, 1, 0,
                 stride, [a - (kt - 1) for a, _ in taps], [c for _, c in taps], dbias=db)
    got = got.view(N, len(taps), upt * 16)[:, :, :C0]                       # [n][tap][c]
    want = w.grad.reshape(N, C0, kt * kf).permute(0, 2, 1)
    assert_close(got.numpy(), want.numpy(), TOL, "dW")
    assert_close(db.cpu().numpy(), dz.sum((0, 2, 3)).numpy(), TOL, "dbias riding on the weight gradient")
    # bf16 products (EAB_PREC_BF16): both operands rounded to bf16, fp32 accumulation -> compare with the same rounding applied
    # to the fp64 operands (then only the accumulation order differs), and loosely with the unrounded gradient
    db2 = torch.zeros(N, device="cuda:0")
    got16 = _wgrad(lib, _dev(dz.permute(0, 2, 3, 1)), _dev(x.permute(0, 2, 3, 1)), None, N, len(taps) * upt * 16, B, T, Fin, Fout, Fout, 1,
                   0, stride, [a - (kt - 1) for a, _ in taps], [c for _, c in taps], dbias=db2, precision=2)
    got16 = got16.view(N, len(taps), upt * 16)[:, :, :C0]
    xr = x.float().bfloat16().double()
    dzr = dz.float().bfloat16().double()
    wr = torch.zeros_like(w, requires_grad=True)
    (F.conv2d(F.pad(xr, (0, 0, kt - 1, 0)), wr, stride=(1, stride)) * dzr).sum().backward()
    assert_close(got16.numpy(), wr.grad.reshape(N, C0, kt * kf).permute(0, 2, 1).numpy(), TOL, "dW, bf16 products")
    assert_close(got16.numpy(), want.numpy(), 2e-2, "dW, bf16 products vs exact")
    assert_close(db2.cpu().numpy(), dz.sum((0, 2, 3)).numpy(), TOL, "dbias (from the unrounded dz)")


@pytest.mark.parametrize("N,C0,C1,kt,kf,B,T,Fin", [(128, 64, 64, 2, 3, 2, 8, 9), (64, 64, 64, 1, 3, 2, 6, 19), (128, 64, 64, 2, 5, 1, 4, 79)])
def test_wgrad_of_a_transposed_convolution_two_sources(lib, N, C0, C1, kt, kf, B, T, Fin):
    """dW of ConvTranspose2d((kt,kf), stride (1,2)) + chomp on a concatenation of two tensors, one launch per output
    column parity (EaBNet.py:423-425, 478-480, 275)."""
    g = torch.Generator().manual_seed(2)
    Cin = C0 + C1
    Fout = (Fin - 1) * 2 + kf
    xa = torch.randn(B, C0, T, Fin, generator=g, dtype=torch.float64)
    xb = torch.randn(B, C1, T, Fin, generator=g, dtype=torch.float64)
    w = torch.randn(Cin, N, kt, kf, generator=g, dtype=torch.float64, requires_grad=True)
    y = F.conv_transpose2d(torch.cat((xa, xb), 1), w, stride=(1, 2))[:, :, :T]
    dz = torch.randn(y.shape, generator=g, dtype=torch.float64)
    (y * dz).sum().backward()
    upt = (Cin + 15) // 16
    dzd, xad, xbd = _dev(dz.permute(0, 2, 3, 1)), _dev(xa.permute(0, 2, 3, 1)), _dev(xb.permute(0, 2, 3, 1))
    for ph in (0, 1):
        tp = [(a, c) for a in range(kt) for c in range(ph, kf, 2)]
        No = (Fout + 1) // 2 if ph == 0 else Fout // 2
        got = _wgrad(lib, dzd, xad, xbd, N, len(tp) * upt * 16, B, T, Fin, Fout, No, 2, ph, 1, [-a for a, _ in tp],
                     [-(c - ph) // 2 for _, c in tp]).view(N, len(tp), upt * 16)[:, :, :Cin]
        want = torch.stack([w.grad[:, :, a, c].T for a, c in tp], dim=1)    # [n][tap][ci]
        assert_close(got.numpy(), want.numpy(), TOL, f"dW phase {ph}")


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("B,P,Cc", [(2, 57, 64), (3, 1300, 64), (1, 31, 256), (2, 5409, 64), (2, 2561, 128)])   # <= 2560: one launch
def test_instance_norm_prelu_forward_and_backward(lib, mode, B, P, Cc):
    """y = prelu(IN(x)) (2-D units) and y = IN(prelu(x)) (S-TCM): statistics kernel, apply kernel and the backward
    (dx, dgamma, dbeta, dslope, with and without an accumulate operand) against autograd."""
    g = torch.Generator().manual_seed(3 + mode)
    x = (torch.randn(B, P, Cc, generator=g, dtype=torch.float64) * 1.3 + 0.4).requires_grad_(True)
    gam = (torch.rand(Cc, generator=g, dtype=torch.float64) + 0.5).requires_grad_(True)
    bet = (torch.randn(Cc, generator=g, dtype=torch.float64) * 0.3).requires_grad_(True)
    slp = (torch.rand(Cc, generator=g, dtype=torch.float64) * 0.4 + 0.05).requires_grad_(True)
    xc = x.permute(0, 2, 1)                                                 # (B, C, P)
    if mode == 1:
        y = F.prelu(F.instance_norm(xc, weight=gam, bias=bet, eps=1e-5), slp)
    else:
        y = F.instance_norm(F.prelu(xc, slp), weight=gam, bias=bet, eps=1e-5)
    dy = torch.randn(B, P, Cc, generator=g, dtype=torch.float64)
    (y.permute(0, 2, 1) * dy).sum().backward()
    xd, gd, bd, sd, dyd = _dev(x.detach()), _dev(gam.detach()), _dev(bet.detach()), _dev(slp.detach()), _dev(dy)
    xf, mr = torch.empty(B, Cc, 2, device="cuda:0"), torch.empty(B, Cc, 2, device="cuda:0")
    _ck(lib.eab_train_in_stats_f32(xd.data_ptr(), sd.data_ptr() if mode == 2 else None, B, P, Cc, 1e-5, gd.data_ptr(), bd.data_ptr(),
                                   xf.data_ptr(), mr.data_ptr(), _st()))
    yd = torch.empty_like(xd)
    _ck(lib.eab_train_norm_act_f32(xd.data_ptr(), xf.data_ptr(), sd.data_ptr(), None, yd.data_ptr(), B, P, Cc, mode, _st()))
    assert_close(yd.cpu().numpy(), y.detach().permute(0, 2, 1).numpy(), TOL, "forward")
    if mode == 2:       # the S-TCM unit as one launch: same statistics, same output
        xf1, mr1, y1 = torch.empty_like(xf), torch.empty_like(mr), torch.empty_like(xd)
        _ck(lib.eab_train_in1d_f32(xd.data_ptr(), sd.data_ptr(), B, P, Cc, 1e-5, gd.data_ptr(), bd.data_ptr(), xf1.data_ptr(),
                                   mr1.data_ptr(), y1.data_ptr(), _st()))
        # (the one-launch form sums its fp64 partials over 64 position lanes when few workgroups walk long slabs, the
        # statistics-only kernel over 16: the fp32 results agree to the last bit or two)
        for got_t, want_t, what in ((xf1, xf, "xf"), (mr1, mr, "mr"), (y1, yd, "y")):
            assert_close(got_t.cpu().numpy(), want_t.cpu().numpy(), 1e-6, f"in1d one launch: {what}")
    sums = torch.full((8, B, Cc, 4), 7.0, device="cuda:0")           # EAB_NB_SUM_COPIES copies of [B][C][4]
    acc = torch.randn(B, P, Cc, device="cuda:0")
    for acc_in in (None, acc):
        dx = torch.empty_like(xd)
        dg, db, ds = (torch.zeros(Cc, device="cuda:0") for _ in range(3))
        flag = 0
        if acc_in is not None:              # second round: the caller zero-fills the scratch itself (EAB_NB_SUMS_ZEROED)
            sums.zero_()
            flag = 0x100
        _ck(lib.eab_train_norm_bwd_f32(dyd.data_ptr(), xd.data_ptr(), mr.data_ptr(), gd.data_ptr(), bd.data_ptr(), sd.data_ptr(),
                                       sums.data_ptr(), acc_in.data_ptr() if acc_in is not None else None, dx.data_ptr(), dg.data_ptr(),
                                       db.data_ptr(), ds.data_ptr(), B, P, Cc, mode | flag, _st()))
        want_dx = x.grad + (acc_in.cpu().double() if acc_in is not None else 0.0)
        assert_close(dx.cpu().numpy(), want_dx.numpy(), TOL, "dx")
        assert_close(dg.cpu().numpy(), gam.grad.numpy(), TOL, "dgamma")
        assert_close(db.cpu().numpy(), bet.grad.numpy(), TOL, "dbeta")
        assert_close(ds.cpu().numpy(), slp.grad.numpy(), TOL, "dslope")


def test_elementwise_backward_kernels(lib):
    """GLU (with the forward's dump layout), S-TCM gate, ReLU mask, add, column sums, filter-and-sum, LayerNorm."""
    g = torch.Generator().manual_seed(5)
    rows, N = 777, 128
    # GLU: packed column r: half = (r%64)/32, channel = (r/64)*32 + r%32
    a = torch.randn(rows, 64, generator=g, dtype=torch.float64, requires_grad=True)
    gt = torch.randn(rows, 64, generator=g, dtype=torch.float64, requires_grad=True)
    y = a * torch.sigmoid(gt)
    dy = torch.randn(rows, 64, generator=g, dtype=torch.float64)
    (y * dy).sum().backward()
    r = np.arange(N)
    half, ch = (r % 64) // 32, (r // 64) * 32 + r % 32
    dump = torch.where(torch.from_numpy(half == 0), a.detach()[:, ch], torch.sigmoid(gt.detach())[:, ch])
    dz = torch.empty(rows, N, device="cuda:0")
    dyd, dumpd = _dev(dy), _dev(dump)            # (device operands are kept in variables: a temporary would be freed, and its
    _ck(lib.eab_glu_bwd_f32(dyd.data_ptr(), dumpd.data_ptr(), dz.data_ptr(), rows, N, _st()))     # block reused, before the launch)
    want = torch.where(torch.from_numpy(half == 0), a.grad[:, ch], gt.grad[:, ch])
    assert_close(dz.cpu().numpy(), want.numpy(), TOL, "glu_bwd")
    # gate
    n = rows * 64
    av, rv = a.detach(), gt.detach()
    z = torch.empty(rows, 64, device="cuda:0")
    avd, rvd = _dev(av), _dev(rv)
    _ck(lib.eab_gate_fwd_f32(avd.data_ptr(), rvd.data_ptr(), z.data_ptr(), n, _st()))
    assert_close(z.cpu().numpy(), y.detach().numpy(), TOL, "gate_fwd")
    da, dr = torch.empty(rows, 64, device="cuda:0"), torch.empty(rows, 64, device="cuda:0")
    _ck(lib.eab_gate_bwd_f32(dyd.data_ptr(), avd.data_ptr(), rvd.data_ptr(), da.data_ptr(), dr.data_ptr(), n, _st()))
    assert_close(da.cpu().numpy(), a.grad.numpy(), TOL, "gate da")
    assert_close(dr.cpu().numpy(), gt.grad.numpy(), TOL, "gate dr")
    # relu mask / add / colsum
    yv = torch.relu(av)
    dx = torch.empty(rows, 64, device="cuda:0")
    yvd = _dev(yv)
    _ck(lib.eab_relu_bwd_f32(dyd.data_ptr(), yvd.data_ptr(), dx.data_ptr(), n, _st()))
    assert torch.equal(dx.cpu(), (dy * (yv > 0)).float())
    out = torch.empty(rows, 64, device="cuda:0")
    _ck(lib.eab_add_f32(avd.data_ptr(), rvd.data_ptr(), out.data_ptr(), n, _st()))
    assert torch.equal(out.cpu(), av.float() + rv.float())
    cs = torch.zeros(64, device="cuda:0")
    _ck(lib.eab_colsum_f32(avd.data_ptr(), cs.data_ptr(), rows, 64, _st()))
    assert_close(cs.cpu().numpy(), av.sum(0).numpy(), TOL, "colsum")
    # filter-and-sum with padded rows and its backward
    B, T, Fq, M, ld = 2, 5, 161, 4, 64
    w = torch.randn(B, T, Fq, M, 2, generator=g, dtype=torch.float64, requires_grad=True)
    x = torch.randn(B, T, Fq, M, 2, generator=g, dtype=torch.float64)
    yr = (w[..., 0] * x[..., 0] - w[..., 1] * x[..., 1]).sum(-1)
    yi = (w[..., 0] * x[..., 1] + w[..., 1] * x[..., 0]).sum(-1)
    yy = torch.stack((yr, yi), 1)
    dout = torch.randn(yy.shape, generator=g, dtype=torch.float64)
    (yy * dout).sum().backward()
    wp = torch.zeros(B, T, Fq, ld, dtype=torch.float64)
    wp[..., :2 * M] = w.detach().reshape(B, T, Fq, 2 * M)
    yd = torch.empty(B, 2, T, Fq, device="cuda:0")
    wpd, xd, doutd = _dev(wp), _dev(x), _dev(dout)
    _ck(lib.eab_filter_sum_ld_f32(wpd.data_ptr(), xd.data_ptr(), yd.data_ptr(), B, T, Fq, M, ld, _st()))
    assert_close(yd.cpu().numpy(), yy.detach().numpy(), TOL, "filter_sum_ld")
    dw = torch.full((B, T, Fq, ld), float("nan"), device="cuda:0")
    _ck(lib.eab_filter_sum_bwd_f32(doutd.data_ptr(), xd.data_ptr(), dw.data_ptr(), B, T, Fq, M, ld, _st()))
    assert_close(dw[..., :2 * M].cpu().numpy(), w.grad.reshape(B, T, Fq, 2 * M).numpy(), TOL, "filter_sum_bwd")
    assert torch.count_nonzero(dw[..., 2 * M:]) == 0
    # LayerNorm(64)
    rows = 1001
    xl = torch.randn(rows, 64, generator=g, dtype=torch.float64, requires_grad=True)
    gl = (torch.rand(64, generator=g, dtype=torch.float64) + 0.5).requires_grad_(True)
    bl = torch.randn(64, generator=g, dtype=torch.float64, requires_grad=True)
    yl = F.layer_norm(xl, (64,), gl, bl, 1e-5)
    dyl = torch.randn(rows, 64, generator=g, dtype=torch.float64)
    (yl * dyl).sum().backward()
    y_d, mr = torch.empty(rows, 64, device="cuda:0"), torch.empty(rows, 2, device="cuda:0")
    xld, gld, bld, dyld = _dev(xl.detach()), _dev(gl.detach()), _dev(bl.detach()), _dev(dyl)
    _ck(lib.eab_layernorm64_fwd_f32(xld.data_ptr(), gld.data_ptr(), bld.data_ptr(), 1e-5, y_d.data_ptr(), mr.data_ptr(), rows, _st()))
    assert_close(y_d.cpu().numpy(), yl.detach().numpy(), TOL, "layernorm fwd")
    dxl, dgl, dbl = torch.empty(rows, 64, device="cuda:0"), torch.zeros(64, device="cuda:0"), torch.zeros(64, device="cuda:0")
    _ck(lib.eab_layernorm64_bwd_f32(dyld.data_ptr(), xld.data_ptr(), mr.data_ptr(), gld.data_ptr(), dxl.data_ptr(), dgl.data_ptr(),
                                    dbl.data_ptr(), rows, _st()))
    assert_close(dxl.cpu().numpy(), xl.grad.numpy(), TOL, "layernorm dx")
    assert_close(dgl.cpu().numpy(), gl.grad.numpy(), TOL, "layernorm dgamma")
    assert_close(dbl.cpu().numpy(), bl.grad.numpy(), TOL, "layernorm dbeta")


@pytest.mark.parametrize("B,T,Fq", [(1, 9, 21), (2, 40, 161), (13, 5, 161)])      # the last: > 2048 sequences, the 16-sequence kernel
def test_lstm_training_forward_and_reverse_time_kernel(lib, B, T, Fq):
    """nn.LSTM(64, 64) over time for B*F sequences (EaBNet.py:610-611): the training forward (gates kept) equals the
    layer, and the reverse-time kernel's dgates reproduce autograd's dW_ih, dW_hh, db and dx."""
    g = torch.Generator().manual_seed(7)
    lstm = torch.nn.LSTM(64, 64, batch_first=True).double()
    x = torch.randn(B * Fq, T, 64, generator=g, dtype=torch.float64, requires_grad=True)
    h, _ = lstm(x)
    dh = torch.randn(h.shape, generator=g, dtype=torch.float64)
    (h * dh).sum().backward()
    wcat = torch.cat((lstm.weight_ih_l0, lstm.weight_hh_l0), 1).detach()
    bias = (lstm.bias_ih_l0 + lstm.bias_hh_l0).detach()
    to_btf = lambda t: t.detach().view(B, Fq, T, -1).permute(0, 2, 1, 3)      # noqa: E731  (B*F, T, C) -> (B, T, F, C)
    xd, dhd, wd, bd = _dev(to_btf(x)), _dev(to_btf(dh)), _dev(wcat), _dev(bias)
    hd = torch.empty(B, T, Fq, 64, device="cuda:0")
    gates = torch.empty(B * Fq, T, 5, 64, device="cuda:0")
    _ck(lib.eab_lstm64_train_fwd_f32(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), hd.data_ptr(), gates.data_ptr(), B, T, Fq, _st()))
    assert_close(hd.cpu().numpy(), to_btf(h).numpy(), TOL, "h")
    dg = torch.empty(B, T, Fq, 256, device="cuda:0")
    _ck(lib.eab_lstm64_bwd_f32(gates.data_ptr(), dhd.data_ptr(), wd.data_ptr(), dg.data_ptr(), B, T, Fq, _st()))
    torch.cuda.synchronize()
    dgc = dg.cpu().double().permute(0, 2, 1, 3).reshape(B * Fq, T, 256)     # (seq, t, 256)
    hprev = torch.cat((torch.zeros(B * Fq, 1, 64, dtype=torch.float64), h.detach()[:, :-1]), 1)
    assert_close(torch.einsum("stk,stc->kc", dgc, x.detach()).numpy(), lstm.weight_ih_l0.grad.numpy(), TOL, "dW_ih")
    assert_close(torch.einsum("stk,stc->kc", dgc, hprev).numpy(), lstm.weight_hh_l0.grad.numpy(), TOL, "dW_hh")
    assert_close(dgc.sum((0, 1)).numpy(), lstm.bias_ih_l0.grad.numpy(), TOL, "db")
    assert_close((dgc @ lstm.weight_ih_l0.detach()).numpy(), x.grad.numpy(), TOL, "dx")
    # the weight gradients through the wgrad kernel: taps dt = 0 (input) and dt = -1 (previous hidden state)
    for name, src, dt, want in (("ih", xd, 0, lstm.weight_ih_l0.grad), ("hh", hd, -1, lstm.weight_hh_l0.grad)):
        got = _wgrad(lib, dg, src, None, 256, 64, B, T, Fq, Fq, Fq, 1, 0, 1, [dt], [0])
        assert_close(got.numpy(), want.numpy(), TOL, f"wgrad {name}")


def _bf(t):
    """round to bf16 (nearest even) and back, in the tensor's own dtype"""
    return t.float().bfloat16().to(t.dtype)


# (2093 sequences: the 16-sequence reverse kernel in bf16; 4186: also the 16-sequence forward instead of the 4-sequence form)
@pytest.mark.parametrize("B,T,Fq", [(1, 9, 21), (2, 40, 161), (13, 7, 161), (26, 4, 161)])
def test_lstm_training_kernels_in_bf16(lib, B, T, Fq):
    """The bf16 training programs run the LSTM's recurrent products on the bf16 matrix cores (precision EAB_PREC_BF16 of
    eab_lstm64_train_fwd_prec_f32 / eab_lstm64_bwd_prec_f32): operands rounded to bf16, everything else fp32.  Reference: the same
    recurrences in fp64 with exactly those roundings -- forward [x_t | h_{t-1}] and W, backward dgates_t and W_hh --, the backward
    fed with the kernel's own stored gates (above 2048 sequences; below, the reverse pass is the fp32 kernel in every mode); and the
    result stays within bf16 distance of the exact layer."""
    g = torch.Generator().manual_seed(11)
    lstm = torch.nn.LSTM(64, 64, batch_first=True).double()
    S = B * Fq
    x = torch.randn(S, T, 64, generator=g, dtype=torch.float64)
    dh = torch.randn(S, T, 64, generator=g, dtype=torch.float64)
    w_ih, w_hh = lstm.weight_ih_l0.detach(), lstm.weight_hh_l0.detach()
    bias = (lstm.bias_ih_l0 + lstm.bias_hh_l0).detach()
    wcat = torch.cat((w_ih, w_hh), 1)
    to_btf = lambda t: t.detach().view(B, Fq, T, -1).permute(0, 2, 1, 3)      # noqa: E731
    xd, dhd, wd, bd = _dev(to_btf(x)), _dev(to_btf(dh)), _dev(wcat), _dev(bias)
    hd = torch.empty(B, T, Fq, 64, device="cuda:0")
    gates = torch.empty(S, T, 5, 64, device="cuda:0")
    _ck(lib.eab_lstm64_train_fwd_prec_f32(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), hd.data_ptr(), gates.data_ptr(), B, T, Fq, 2, _st()))
    # forward reference with the kernel's roundings (the device holds fp32 x and weights: round THOSE)
    xr, wir, whr = _bf(xd.cpu().permute(0, 2, 1, 3).reshape(S, T, 64)).double(), _bf(wd.cpu()[:, :64]).double(), _bf(wd.cpu()[:, 64:]).double()
    bf = bd.cpu().double()
    h, c = torch.zeros(S, 64, dtype=torch.float64), torch.zeros(S, 64, dtype=torch.float64)
    hs, gs = [], []
    for t in range(T):
        pre = xr[:, t] @ wir.T + _bf(h.float()).double() @ whr.T + bf
        i, f, gg, o = torch.sigmoid(pre[:, :64]), torch.sigmoid(pre[:, 64:128]), torch.tanh(pre[:, 128:192]), torch.sigmoid(pre[:, 192:])
        c = f * c + i * gg
        h = o * torch.tanh(c)
        hs.append(h)
        gs.append(torch.stack((i, f, gg, o, c), 1))
    h_ref, g_ref = torch.stack(hs, 1), torch.stack(gs, 1)                     # (S, T, 64), (S, T, 5, 64)
    h_got = hd.cpu().double().permute(0, 2, 1, 3).reshape(S, T, 64)
    # (a value that lands on the other side of a bf16 rounding boundary of h_{t-1} moves the next step by 2^-9 of one operand)
    assert_close(h_got.numpy(), h_ref.numpy(), 2e-3, "h (bf16 recurrence)", tol_max=1e-2)
    assert_close(gates.cpu().double().numpy(), g_ref.numpy(), 2e-3, "stored gates", tol_max=1e-2)
    h_exact, _ = lstm(x)
    err = float((h_got - h_exact.detach()).norm() / h_exact.detach().norm())
    assert 1e-4 < err < 2e-2, err                                             # really bf16 products, and no worse than bf16
    # backward: the reverse recurrence on the kernel's own gates, dgates and W_hh rounded for the recurrent product
    dg = torch.empty(B, T, Fq, 256, device="cuda:0")
    _ck(lib.eab_lstm64_bwd_prec_f32(gates.data_ptr(), dhd.data_ptr(), wd.data_ptr(), dg.data_ptr(), B, T, Fq, 2, _st()))
    torch.cuda.synchronize()
    gk = gates.cpu().double()
    dhq = dhd.cpu().double().permute(0, 2, 1, 3).reshape(S, T, 64)
    dhr, dcc = torch.zeros(S, 64, dtype=torch.float64), torch.zeros(S, 64, dtype=torch.float64)
    want = torch.zeros(S, T, 256, dtype=torch.float64)
    for t in range(T - 1, -1, -1):
        i, f, gg, o, c = (gk[:, t, k] for k in range(5))
        cp = gk[:, t - 1, 4] if t > 0 else torch.zeros_like(c)
        d = dhq[:, t] + dhr
        tc = torch.tanh(c)
        d_o = d * tc * o * (1 - o)
        dc = dcc + d * o * (1 - tc * tc)
        d_i, d_g, d_f = dc * gg * i * (1 - i), dc * i * (1 - gg * gg), dc * cp * f * (1 - f)
        dcc = dc * f
        want[:, t] = torch.cat((d_i, d_f, d_g, d_o), 1)
        dhr = _bf(want[:, t].float()).double() @ whr
    dgc = dg.cpu().double().permute(0, 2, 1, 3).reshape(S, T, 256)
    dg32 = torch.empty_like(dg)
    _ck(lib.eab_lstm64_bwd_f32(gates.data_ptr(), dhd.data_ptr(), wd.data_ptr(), dg32.data_ptr(), B, T, Fq, _st()))
    torch.cuda.synchronize()
    if S > 2048:
        assert_close(dgc.numpy(), want.numpy(), 2e-3, "dgates (bf16 recurrence)", tol_max=1e-2)
        err = float((dg - dg32).norm() / dg32.norm())
        assert 1e-5 < err < 2e-2, err                     # bf16 distance from the fp32 kernel on the same gates, not more
    else:
        # up to 2048 sequences the reverse pass takes the 4-sequence fp32 kernel in every mode (faster there, and exact)
        assert torch.equal(dg, dg32)


def _wgrad_masked(lib, dz, src0, src1, N, Kpad, B, T, Fin, Fz, No, ostride, ophase, istride, dt, ioff, mask, C0, C1, dbias=None):
    from eabnet_amd import _lib
    d = _lib.WgradDesc()
    dw = torch.zeros(N, Kpad, device="cuda:0")
    d.dz, d.src0, d.src1, d.dw = dz.data_ptr(), src0.data_ptr(), (src1.data_ptr() if src1 is not None else None), dw.data_ptr()
    d.dbias = dbias.data_ptr() if dbias is not None else None
    d.N, d.C0, d.C1, d.Kpad = N, C0, C1, Kpad
    d.B, d.T, d.Fin, d.Fz, d.No, d.ostride, d.ophase, d.istride = B, T, Fin, Fz, No, ostride, ophase, istride
    d.ntaps = len(dt)
    d.precision = 2
    d.bf16_mask = mask
    for j in range(len(dt)):
        d.dt[j], d.ioff[j] = dt[j], ioff[j]
    _ck(lib.eab_wgrad_f32(C.byref(d), _st()), "eab_wgrad_f32")
    torch.cuda.synchronize()
    return dw.cpu()


@pytest.mark.parametrize("N,C0,C1,taps,B,T,Fin,No,ostride,istride", [
    (64, 64, 0, [(0, 0), (0, 1), (0, 2)], 2, 9, 39, 19, 1, 2),            # Conv2dunit (1,3) / stride 2
    (128, 64, 0, [(-1, 0), (-1, 1), (-1, 2), (0, 0), (0, 1), (0, 2)], 2, 7, 19, 9, 1, 2),   # GateConv2d (2,3)
    (64, 64, 64, [(0, 0), (0, -1)], 2, 6, 19, 20, 2, 1),                  # Deconv2dunit phase 0 on a concatenation
    (128, 16, 0, [(0, 0), (0, 1), (0, 2), (0, 3), (0, 4)], 1, 5, 161, 79, 1, 2),   # first convolution: fp32 input, bf16 dz only
])
def test_wgrad_with_bf16_stored_operands(lib, N, C0, C1, taps, B, T, Fin, No, ostride, istride):
    g = torch.Generator().manual_seed(7)
    Fz = (No - 1) * ostride + 1 + (1 if ostride == 2 else 0)
    dz = torch.randn(B, T, Fz, N, generator=g)
    x0 = torch.randn(B, T, Fin, C0, generator=g)
    x1 = torch.randn(B, T, Fin, C1, generator=g) if C1 else None
    upt = (C0 + C1 + 15) // 16
    Kpad = len(taps) * upt * 16
    dt, io = [a for a, _ in taps], [c for _, c in taps]
    full = (C0 % 16 == 0)
    db32, db16 = torch.zeros(N, device="cuda:0"), torch.zeros(N, device="cuda:0")
    ref = _wgrad_masked(lib, _dev(dz), _dev(x0), _dev(x1) if C1 else None, N, Kpad, B, T, Fin, Fz, No, ostride, 0, istride, dt, io, 0,
                        C0, C1, dbias=db32)
    h = lambda t: t.to("cuda:0").bfloat16().contiguous()        # noqa: E731 (round to nearest even, 2-byte elements)
    mask = 1 | ((2 | (4 if C1 else 0)) if full else 0)
    got = _wgrad_masked(lib, h(dz), h(x0) if full else _dev(x0), (h(x1) if full else _dev(x1)) if C1 else None, N, Kpad, B, T, Fin, Fz, No,
                        ostride, 0, istride, dt, io, mask, C0, C1, dbias=db16)
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 2e-5 * scale, "same operand bits: only the order of the fp32 atomics may differ"
    want_db = dz.bfloat16().float().sum((0, 1, 2))
    rows = torch.zeros(B, T, Fz, dtype=torch.bool)
    rows[:, :, 0:(No - 1) * ostride + 1:ostride] = True
    want_db = (dz.bfloat16().float() * rows[..., None]).sum((0, 1, 2))
    assert_close(db16.cpu().numpy(), want_db.numpy(), 1e-4, "dbias from the stored (bf16) gradient")


def _conv_plain(lib, src0, src1, w, N, C0, C1, B, T, Fin, Fout, No, ostride, ophase, istride, dt, ioff, src_bf16, bm=128):
    from eabnet_amd import _lib
    d = _lib.ConvDesc()
    out = torch.full((B, T, Fout, N), float("nan"), device="cuda:0")
    d.src0, d.src1, d.w, d.dst = src0.data_ptr(), (src1.data_ptr() if src1 is not None else None), w.data_ptr(), out.data_ptr()
    d.C0, d.C1, d.N, d.Kpad, d.Cout = C0, C1, N, w.shape[1], N
    d.B, d.T, d.Fin, d.Fout, d.No, d.ostride, d.ophase, d.istride = B, T, Fin, Fout, No, ostride, ophase, istride
    d.ntaps = len(dt)
    for j in range(len(dt)):
        d.dt[j], d.ioff[j] = dt[j], ioff[j]
    d.epi, d.bm, d.precision, d.korder, d.src_bf16 = 0, bm, 2, 0, src_bf16
    _ck(lib.eab_conv_f32(C.byref(d), _st()), "eab_conv_f32")
    torch.cuda.synchronize()
    return out.cpu()


@pytest.mark.parametrize("N,C0,C1,taps,B,T,Fin,No,ostride,istride,bm", [
    (64, 64, 0, [(0, 0), (0, 1), (0, 2)], 2, 40, 39, 19, 1, 2, 128),      # unit convolution, 128-row tiles (two register sets)
    (64, 64, 64, [(0, 0), (0, -1)], 2, 17, 19, 20, 2, 1, 64),             # transposed unit, phase 0, two sources, 64-row tiles
    (128, 128, 0, [(1, 0), (1, -1), (0, 0), (0, -1)], 1, 21, 9, 10, 2, 1, 64),   # a dgrad-like launch with look-ahead taps
    (64, 128, 0, [(0, 0)], 2, 12, 79, 79, 1, 1, 128),                     # 1x1
])
def test_conv_with_bf16_stored_sources_is_bit_identical(lib, N, C0, C1, taps, B, T, Fin, No, ostride, istride, bm):
    """EAB_PREC_BF16 gather on sources STORED as bf16 (walked in 32-channel units) == the same launch on the fp32 tensors holding
    the same (already rounded) values: same operand bits, same k order -> identical fp32 results."""
    g = torch.Generator().manual_seed(11)
    x0 = torch.randn(B, T, Fin, C0, generator=g).bfloat16()
    x1 = torch.randn(B, T, Fin, C1, generator=g).bfloat16() if C1 else None
    upt = (C0 + C1 + 15) // 16
    w = torch.randn(N, len(taps) * upt * 16, generator=g).to("cuda:0")
    Fout = (No - 1) * ostride + 1
    dt, io = [a for a, _ in taps], [c for _, c in taps]
    a32 = _conv_plain(lib, x0.float().to("cuda:0").contiguous(), x1.float().to("cuda:0").contiguous() if C1 else None, w, N, C0, C1, B, T,
                      Fin, Fout, No, ostride, 0, istride, dt, io, 0, bm)
    a16 = _conv_plain(lib, x0.to("cuda:0").contiguous(), x1.to("cuda:0").contiguous() if C1 else None, w, N, C0, C1, B, T, Fin, Fout, No,
                      ostride, 0, istride, dt, io, 1 | (2 if C1 else 0), bm)
    written = torch.isfinite(a32)
    assert written.any() and torch.equal(torch.isfinite(a16), written)
    assert torch.equal(a16[written], a32[written])
    if C1:   # the sources of a launch are stored alike (the 32-channel walk is a compile-time property): a mixed mask is refused
        from eabnet_amd import _lib
        with pytest.raises(_lib.EabError):
            _conv_plain(lib, x0.to("cuda:0").contiguous(), x1.float().to("cuda:0").contiguous(), w, N, C0, C1, B, T, Fin, Fout, No,
                        ostride, 0, istride, dt, io, 1, bm)
