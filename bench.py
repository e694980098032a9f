#!/usr/bin/env python
"""Headline benchmark: enhanced STFT frames/s of the EaBNet hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of synthetic input that is
already resident in HBM: (B, M, L) waves -> fused STFT/compression ->
EaBNet.forward -> (B, 2, T, F).  Workload = BASELINE.json configs[1]/[2]
(batch of 16 four-second 8-mic 16 kHz utterances per GPU, fp32, full
hand-written HIP path).  Utterances are independent, so N GPUs run N
independent shards (weak scaling, no data-path collective).

Prints ONE JSON line on rank 0 (see the keys below); `roofline` is measured in
an instrumented replay (HIP events around every op of the program, on the
stream the kernels run on) right after the timed region, `cpu_baseline` times
the oracle (oracle/eabnet_oracle.py, test infrastructure) on the host cores on
a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import eabnet_amd  # noqa: E402
from eabnet_amd import dist  # noqa: E402
from eabnet_amd import program as prg  # noqa: E402

B_PER_GPU, MICS, SR, SECONDS = 16, 8, 16000, 4.0
N_FFT, HOP = 320, 160
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense fp32 matrix peak (= vector peak)
PEAK_HBM_GBS = 8000.0


def kernel_source_sha() -> str:
    """sha256 (first 16 hex digits) over the kernel sources: committed PMC summaries carry it, and `roofline.traffic`
    is reported only when the summary was taken on the kernels of this tree (there is no git on the GPU box)"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "eabnet_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "eabnet_amd", "csrc", "*.h"))
                  + glob.glob(os.path.join(ROOT, "eabnet_amd", "csrc", "*.inc"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def mac_per_frame(M: int) -> int:
    """SURVEY §0 / BASELINE.md: conv + linear + LSTM-gate MACs per STFT frame."""
    return 44_404_736 + 222_848 * M


def conv_kernel_mac_per_frame(M: int) -> int:
    """The share executed by conv_gemm_kernel: everything except the two LSTM
    layers (2*161*256*128) and the w_dnn MLP (161*64*(64 + 2M), fused with the filter-and-sum)."""
    return mac_per_frame(M) - 2 * 161 * 256 * 128 - 161 * 64 * (64 + 2 * M)


def make_model(M: int, device, seed: int = 0):
    torch.manual_seed(seed)
    net = eabnet_amd.EaBNet(M=M).eval()
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():                      # norm / PReLU parameters away from 1 / 0 / 0.25
        for n, p in net.named_parameters():
            if n.endswith("norm.weight"):
                p.copy_(torch.empty_like(p).uniform_(0.5, 1.5, generator=g))
            elif n.endswith("norm.bias"):
                p.copy_(torch.empty_like(p).uniform_(-0.3, 0.3, generator=g))
    state = {k: v.detach().clone() for k, v in net.state_dict().items()}
    return net.to(device), state


def synth_waves(B: int, M: int, L: int, seed: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return 0.05 * torch.randn(B, M, L, generator=g)


def instrumented_replay(net, ns, reps: int):
    """Per-op device time with HIP events on the launch stream.  A spin kernel
    keeps the GPU busy while the host enqueues, so that event timestamps bracket
    kernels and not host latency."""
    bound = net._last[0]
    ops = bound.prog.ops
    stream = torch.cuda.current_stream()
    totals = np.zeros(len(ops))
    for _ in range(reps):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(ops) + 1)]
        torch.cuda._sleep(int(4e8))
        for k in range(len(ops)):
            evs[k].record(stream)
            bound.run(stream.cuda_stream, k, 1)
        evs[-1].record(stream)
        torch.cuda.synchronize()
        totals += np.array([evs[k].elapsed_time(evs[k + 1]) for k in range(len(ops))])
    return ops, totals / reps            # ms per op


def cpu_baseline(state, M: int, L: int, budget_s: float = 15.0):
    """Oracle on the host cores (kind "port"): B=2 utterances of the same shape,
    repeated until ~budget_s of CPU work, best pass reported."""
    from oracle import eabnet_oracle as orc          # test infrastructure: checker / baseline only
    Bc = 4
    wav = synth_waves(Bc, M, L, 4321)
    T = 1 + L // HOP
    # a 1-GPU box exposes every host CPU but the job's share is 16 cores
    # (oversubscribing oneDNN with 128 threads ran 6x slower): use min(16, visible)
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    torch.set_num_threads(min(16, ncpu))
    best, spent, passes = float("inf"), 0.0, 0
    with torch.no_grad():
        while passes < 2 or (spent < budget_s and passes < 40):
            t0 = time.perf_counter()
            ns, _ = orc.prepare_data_oracle(wav, None, N_FFT, HOP, N_FFT)
            orc.eabnet_forward(state, ns, fast_lstm=True)
            dt = time.perf_counter() - t0
            best, spent, passes = min(best, dt), spent + dt, passes + 1
            print(f"[bench] cpu_baseline pass {passes}: {dt:.2f} s", file=sys.stderr, flush=True)
    return {"value": Bc * T / best, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{Bc} utterances x {SECONDS:.0f} s x {M} mics (T={T}), best of {passes} passes, "
                      f"{spent:.1f} s CPU total; oracle = PyTorch-CPU (ATen/oneDNN) restatement; "
                      f"{ncpu} CPUs visible, 16-core job share"}


_CALIBRATED: list = []


def _operator_path():
    """tests/operator_path.py: the PyTorch-ROCm operator evaluation of the networks, a comparator (in-run loss check of the
    training rows, --train-operator-path); the package itself has no such backend"""
    tdir = os.path.join(ROOT, "tests")
    if tdir not in sys.path:
        sys.path.insert(0, tdir)
    import operator_path
    return operator_path


def train_measure(rank, world, dev, is_dist, *, precision="f32", two_stage=False, operator_path=False, ddp=False,
                  steps=5, warmup=2, per_op="", roofline=True, check=False):
    """BASELINE configs[3] / train_distributed.py:214-230: per step prepare_data (noisy + target STFT) -> net(noisy) ->
    com_mag_mse_loss (two_stage: eabnet_with_postnet_loss of the model train_distributed.py:181 builds) -> backward ->
    clip_grad_norm_(1.0) -> Adam(5e-4), forward and backward on the HIP training programs (eabnet_amd/train.py, train_gag.py),
    gradients averaged over the ranks.  Returns the result dict of one configuration (nothing is printed).
    check=True: before the timed steps, the loss of the first step is compared with the PyTorch-ROCm operator comparator
    (tests/operator_path.py -- test infrastructure, never timed in the default run) on the same parameters and batch, and
    every gradient must be finite.  operator_path=True times that comparator instead (the --train-operator-path line)."""
    from eabnet_amd import train as tr
    B, M, seconds = 6, MICS, 6.0                            # train_distributed.py:273,279 (batch 6, wav_len 6 s)
    L = int(seconds * SR)
    T = 1 + L // HOP
    net, _ = make_model(M, dev)
    net.train()
    net.precision = "bf16" if precision == "bf16" else "f32"
    two = None
    if two_stage:
        # the model train_distributed.py:181 actually builds: beam-former + GaGNet post-filter, both trained
        # (eabnet_with_postnet_loss, :225), the post-filter fed esti0.detach() (EaBNet.py:142).  Both stages run on their HIP
        # training programs (train.py, train_gag.py); operator_path puts both on PyTorch-ROCm operators.
        pa = argparse.Namespace(
            k1=(2, 3), k2=(1, 3), c=64, M=M, embed_dim=64, kd1=5, cd1=64, d_feat=256, p=6, q=3, is_causal=True, is_u2=True,
            bf_type="lstm", topo_type="mimo", intra_connect="cat", norm_type="IN", ref_mic=0, freeze_eabnet=False,
            gagnet_k1=(2, 3), gagnet_k2=(1, 3), gagnet_c=64, gagnet_kd1=3, gagnet_cd1=64, gagnet_d_feat=256, gagnet_p=2,
            gagnet_q=3, gagnet_dilas=[1, 2, 5, 9], gagnet_fft_num=320, gagnet_is_u2=True, gagnet_is_causal=True,
            gagnet_is_squeezed=False, gagnet_acti_type="sigmoid", gagnet_intra_connect="cat", gagnet_norm_type="IN")
        torch.manual_seed(1)
        two = eabnet_amd.make_eabnet_with_postnet(pa).to(dev).train()
        two.eabnet.load_state_dict(net.state_dict(), strict=True)
        two.eabnet.precision = two.postnet.precision = net.precision
        net = two.eabnet
    pd_args = argparse.Namespace(mics=M, sr=SR, wav_len=seconds, win_size=0.020, win_shift=0.010, fft_num=N_FFT)
    wav = synth_waves(B, M, L, 1234 + rank).to(dev)
    tgt = synth_waves(B, 1, L, 4321 + rank).to(dev)
    model = net
    if is_dist and two is not None:
        # data-parallel two-stage training: one flat gradient all-reduce per stage and step (RCCL)
        tr.broadcast_parameters(two)
        tr.enable_flat_allreduce(two.eabnet)
        tr.enable_flat_allreduce(two.postnet)
    elif is_dist:
        tr.broadcast_parameters(net)
        if ddp:
            inner = _operator_path().OperatorPath(net) if operator_path else net
            model = torch.nn.parallel.DistributedDataParallel(inner, device_ids=[dev.index], bucket_cap_mb=64,
                                                              gradient_as_bucket_view=True, static_graph=True)
        else:
            tr.enable_flat_allreduce(net)
    trained = two if two is not None else net
    opt = torch.optim.Adam(trained.parameters(), lr=5e-4)
    frames_list = [T] * B
    # the PyTorch-ROCm operator evaluation of the same modules on the same parameter objects (comparator only)
    op_fwd = _operator_path().OperatorPath(trained) if (operator_path or check) else None

    def loss_of(noisy, target, fwd=None):
        if fwd is None:
            fwd = (op_fwd if model is net else model) if operator_path else (two if two is not None else model)
        if two is not None:
            return eabnet_amd.eabnet_with_postnet_loss(fwd(noisy), target, frames_list)["final"]
        return eabnet_amd.com_mag_mse_loss(fwd(noisy), target, frames_list)

    def step():
        opt.zero_grad(set_to_none=True)
        noisy, target = eabnet_amd.prepare_data(wav, tgt, dev, pd_args)
        loss = loss_of(noisy, target)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(trained.parameters(), 1.0)
        opt.step()
        return loss

    checked = None
    if check and not operator_path:
        # in-run check on the untouched initial parameters: the HIP programs' loss against the operator path's on the same
        # batch (fp32: 1e-4 relative; bf16 products: its stated 5e-2 bound), all gradients finite
        noisy, target = eabnet_amd.prepare_data(wav, tgt, dev, pd_args)
        opt.zero_grad(set_to_none=True)
        l_hip = loss_of(noisy, target)
        l_hip.backward()
        finite = all(bool(torch.isfinite(p.grad).all()) for p in trained.parameters() if p.grad is not None)
        n_grads = sum(1 for p in trained.parameters() if p.grad is not None)
        mods = [two.eabnet, two.postnet] if two is not None else [net]
        hip_engaged = all(m_.training_backend == "hip" for m_ in mods)
        l_op = loss_of(noisy, target, op_fwd)              # operator comparator, same parameters and batch (no backward)
        rel = abs(float(l_hip.detach()) - float(l_op.detach())) / max(abs(float(l_op.detach())), 1e-12)
        tol = 1e-4 if precision == "f32" else 5e-2
        checked = {"loss_hip": float(l_hip.detach()), "loss_operator_path": float(l_op.detach()), "rel_diff": rel, "tolerance": tol,
                   "gradients_finite": finite, "gradient_tensors": n_grads, "hip_programs_engaged": hip_engaged,
                   "ok": bool(rel <= tol and finite and hip_engaged)}
        del l_op, l_hip
        opt.zero_grad(set_to_none=True)
        torch.cuda.empty_cache()
        if not checked["ok"]:
            return {"check": checked, "failed": True}

    for _ in range(max(1, warmup)):
        loss = step()
    import gc
    gc.collect()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    per_rank = dist.gather_over_ranks(el, dev)
    elapsed = dist.max_over_ranks(el, dev)
    assert bool(torch.isfinite(loss)), "training diverged"
    # data-parallel sanity the line carries: after the same number of averaged-gradient steps every rank must hold the same
    # parameters (each rank trained on its own shard, so the local losses differ -- the replicas must not)
    with torch.no_grad():
        chk = float(sum(p.detach().double().sum() for p in trained.parameters()))
    chk_ranks = dist.gather_over_ranks(chk, dev)
    loss_ranks = dist.gather_over_ranks(float(loss.detach()), dev)
    replicas = {"param_checksum_per_rank": chk_ranks, "final_loss_per_rank": loss_ranks,
                "replicas_identical": bool(max(chk_ranks) - min(chk_ranks) <= 1e-9 * max(1.0, abs(chk_ranks[0])))}
    if not operator_path:
        assert net.training_backend == "hip", "the HIP training programs did not engage (see the RuntimeWarning)"
    frames = world * B * T * steps
    if two_stage:
        out = {"mode": "training step of the two-stage model (train_distributed.py:181,225): beam-former and GaGNet "
                       "post-filter on " + ("PyTorch-ROCm operators (comparison line)" if operator_path else
                                            "their HIP training programs"),
               "dtype": net.precision,
               "value": frames / elapsed, "unit": "frames/s (trained)", "n_gpus": world, "steps": steps,
               "ms_per_step": 1e3 * elapsed / steps, "final_loss": float(loss.detach()),
               "params": eabnet_amd.numParams(two)}
        if not operator_path:
            fl = sum(next(iter(m_._train_bound.values())).prog.flops_fwd + next(iter(m_._train_bound.values())).prog.flops_bwd
                     for m_ in (two.eabnet, two.postnet))
            out["whole_step_tflops"] = world * fl / (elapsed / steps) / 1e12
        if checked:
            out["check"] = checked
        out["ranks"] = {"world_size_seen_by_torch_distributed": torch.distributed.get_world_size() if is_dist else 1,
                        "frames_per_s_per_rank": [B * T * steps / e for e in per_rank], **replicas}
        return out
    if operator_path:
        return {"mode": "training step on PyTorch-ROCm operators (tests/operator_path.py): comparison line", "value": frames / elapsed,
                "unit": "frames/s (trained)", "n_gpus": world, "steps": steps, "ms_per_step": 1e3 * elapsed / steps,
                "final_loss": float(loss.detach())}
    bound = next(reversed(net._train_bound.values()))
    prog = bound.prog
    out = {
        "metric": "enhanced frames/sec (16 kHz, 8-mic) at 1/2/4/8 MI355X; RTF per utterance",
        "mode": "training step (BASELINE configs[3])",
        "value": frames / elapsed, "unit": "frames/s (trained)", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if net.precision == "f32" else "bf16 products (forward, dgrad and wgrad contractions), fp32 accumulate / LSTM / norms / gradients / Adam",
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[3] (train_distributed.py:214-230) for the beam-former stage: per-GPU batch 6 x 6 s x "
                               "8 mics; prepare_data (noisy + target STFT), EaBNet forward, com_mag_mse_loss, backward, "
                               "clip_grad_norm_(1.0), Adam(5e-4); forward and backward on the HIP training programs",
                   "batch_per_gpu": B, "global_batch": world * B, "mics": M, "frames_per_utt": T,
                   "parallelism": f"dp{world}: " + ("torch DDP, one 64 MB bucket" if ddp else
                                                    "one flat RCCL all-reduce of the 2.84 M-float gradient per step")},
        "final_loss": float(loss.detach()),
        "gflop_per_step": {"forward": prog.flops_fwd / 1e9, "backward": prog.flops_bwd / 1e9},
        "whole_step_tflops": world * (prog.flops_fwd + prog.flops_bwd) / (elapsed / steps) / 1e12,
        "ranks": {"world_size_seen_by_torch_distributed": torch.distributed.get_world_size() if is_dist else 1,
                  "frames_per_s_per_rank": [B * T * steps / e for e in per_rank], **replicas},
        "workspace_GB": prog.a_floats * 4 / 1e9,
    }
    if checked:
        out["check"] = checked
    try:
        if rank == 0 and roofline:
            # per-op device time of both programs (HIP events on the launch stream, one more forward/backward state is live)
            stream = torch.cuda.current_stream()
            res = {}
            for which, ops in (("fwd", prog.fwd), ("bwd", prog.bwd)):
                evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(ops) + 1)]
                if which == "bwd":
                    bound.g.zero_()
                torch.cuda._sleep(int(4e8))
                for k in range(len(ops)):
                    evs[k].record(stream)
                    bound.run(which, stream.cuda_stream, k, 1)
                evs[-1].record(stream)
                torch.cuda.synchronize()
                ms = np.array([evs[k].elapsed_time(evs[k + 1]) for k in range(len(ops))])
                by = {}
                for k, o in enumerate(ops):
                    nm = {prg.OP_CONV: "conv_gemm(dgrad)" if which == "bwd" else "conv_gemm", tr.OP_WGRAD: "wgrad", tr.OP_NORM_BWD: "norm_bwd",
                          tr.OP_LSTM_BWD: "lstm_bwd", tr.OP_LSTM_TRAIN: "lstm_fwd", tr.OP_TR_NORM_ACT: "norm_act", tr.OP_COLSUM: "colsum",
                          tr.OP_GLU_BWD: "glu_bwd", prg.OP_IN_FINALIZE: "in_finalize"}.get(o.kind, "other")
                    by[nm] = by.get(nm, 0.0) + float(ms[k])
                res[which] = {"ms_total": float(ms.sum()), "ms_by_kernel": {k: round(v, 3) for k, v in sorted(by.items(), key=lambda kv: -kv[1])}}
                if per_op:
                    with open(per_op + "." + which, "w") as f:
                        f.write("idx kind ms gflop tflops name geometry\n")
                        for k, o in enumerate(ops):
                            gf, geo = 0.0, ""
                            if o.kind in (prg.OP_CONV, tr.OP_WGRAD):
                                gf = 2e-9 * o.B * o.T * o.No * o.N * len(o.dt) * (o.C0 + o.C1)
                                geo = f"N={o.N} C={o.C0}+{o.C1} taps={len(o.dt)} Fin={o.Fin} No={o.No}"
                            f.write(f"{k} {o.kind} {ms[k]:.4f} {gf:.3f} {gf / max(ms[k], 1e-9):.2f} {o.name} {geo}\n")
                if which == "bwd":
                    wg = [k for k, o in enumerate(ops) if o.kind == tr.OP_WGRAD]
                    wg_fl = sum(2.0 * o.B * o.T * o.No * o.N * len(o.dt) * (o.C0 + o.C1) for o in ops if o.kind == tr.OP_WGRAD)
                    # the weight gradients sit at the end of the backward program, sorted by geometry; eab_run_program
                    # serves every run of identical geometry with one launch -- time the block as it runs in production
                    assert wg == list(range(wg[0], len(ops)))
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    bound.g.zero_()
                    e0.record(stream)
                    bound.run(which, stream.cuda_stream, wg[0], len(wg))
                    e1.record(stream)
                    torch.cuda.synchronize()
                    wg_ms = float(e0.elapsed_time(e1))

                    def geo(o):
                        return (o.N, o.C0, o.C1, o.Kpad, o.Fin, o.Fz, o.No, o.ostride, o.ophase, o.istride, tuple(o.dt), tuple(o.ioff),
                                o.src1 is None, o.dbias is None)
                    launches, prev, run = 0, None, 0
                    for o in (ops[k] for k in wg):
                        if geo(o) == prev and run < 24:
                            run += 1
                        else:
                            launches, run = launches + 1, 1
                        prev = geo(o)
                    res[which]["ms_total"] += wg_ms - float(ms[wg].sum())
                    res[which]["ms_by_kernel"]["wgrad"] = round(wg_ms, 3)
                    res[which]["ms_by_kernel"]["wgrad (one launch per descriptor)"] = round(float(ms[wg].sum()), 3)
                    # operand bytes as they are STORED (bf16-stored operands, eab_wgrad_desc.bf16_mask, are two bytes each)
                    def wbytes(o):
                        m = getattr(o, "bf16_mask", 0)
                        return float(o.B * o.T * o.No) * (o.N * (2 if m & 1 else 4)
                                                          + len(o.dt) * (o.C0 * (2 if m & 2 else 4) + o.C1 * (2 if m & 4 else 4)))
                    wg_by = sum(wbytes(o) for o in ops if o.kind == tr.OP_WGRAD)
                    wg_by32 = sum(4.0 * o.B * o.T * o.No * (o.N + len(o.dt) * (o.C0 + o.C1)) for o in ops if o.kind == tr.OP_WGRAD)
                    if net.precision == "bf16":
                        # bf16 products: the matrix work is 1/16 of the fp32 form's, the kernel is bound by its operand traffic
                        # (rows x (N + taps x C) x 4 B per descriptor: dz once, the gathered activations once per tap) and by the
                        # split-K atomics
                        out["roofline"] = {"kernel": "wgrad_bf_kernel (bf16 MFMA 32x32x16, fp32 accumulate, split-K, atomics)",
                                           "bound": "hbm", "achieved": wg_by / (wg_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                           "frac": wg_by / (wg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "traffic": None,
                                           "achieved_tflops": wg_fl / (wg_ms * 1e-3) / 1e12,
                                           "weight_gradients_per_step": len(wg), "launches_per_step": launches, "ms_per_step": wg_ms,
                                           "bf16_stored_operand_descriptors": sum(1 for k in wg if getattr(ops[k], "bf16_mask", 0)),
                                           "achieved_if_all_operands_were_fp32": wg_by32 / (wg_ms * 1e-3) / 1e9,
                                           "note": "all weight gradients of the step: operand bytes as stored (bf16-stored operands two "
                                                   "bytes; dz once, the gathered activations once per tap) / HIP-event time of the block as "
                                                   "it runs in production (descriptors of identical geometry share a launch).  The kernel "
                                                   "is not byte-bound: halving operands did not shorten it (DESIGN.md, round 4 item 3)"}
                    else:
                        out["roofline"] = {"kernel": "wgrad_kernel (fp32 MFMA 32x32x2, split-K, atomics)", "bound": "mfma",
                                           "achieved": wg_fl / (wg_ms * 1e-3) / 1e12, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                           "frac": wg_fl / (wg_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, "traffic": None,
                                           "weight_gradients_per_step": len(wg), "launches_per_step": launches, "ms_per_step": wg_ms,
                                           "note": "all weight gradients of the step: exact FLOPs / HIP-event time of the block as it "
                                                   "runs in production (descriptors of identical geometry share a launch)"}
                    dg = [k for k, o in enumerate(ops) if o.kind == prg.OP_CONV]
                    dg_fl = sum(2.0 * o.B * o.T * o.No * o.N * len(o.dt) * (o.C0 + o.C1) for o in ops if o.kind == prg.OP_CONV)
                    out["dgrad"] = {"achieved_tflops": dg_fl / (float(ms[dg].sum()) * 1e-3) / 1e12, "ms_per_step": float(ms[dg].sum())}
            out["programs"] = res
    except Exception as e:  # noqa: BLE001
        out["secondary_sections_error"] = repr(e)
    return out


def main_train(a, rank, world, dev, is_dist):
    """`bench.py --train [...]`: one training configuration, printed as the job's JSON line"""
    out = train_measure(rank, world, dev, is_dist, precision=a.precision, two_stage=a.train_two_stage,
                        operator_path=a.train_operator_path, ddp=a.train_ddp, steps=a.steps, warmup=a.warmup,
                        per_op=a.per_op, roofline=not a.no_roofline)
    if rank == 0:
        print(json.dumps(out))
    if is_dist:
        dist.barrier()
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the secondary f16x3 measurement")
    ap.add_argument("--no-c1", action="store_true", help="skip the single-utterance (B=1) latency measurement")
    ap.add_argument("--no-graph", action="store_true", help="direct kernel launches instead of hipGraph replay")
    ap.add_argument("--pipeline", type=int, default=3,
                    help="batches in flight on separate HIP streams (eabnet_amd.Pipeline); 1 = strictly one step after the other")
    ap.add_argument("--no-next", action="store_true", help="skip the next-row measurements (post-filter, ISTFT)")
    ap.add_argument("--no-train", action="store_true", help="skip the training-step section of the next rows (BASELINE configs[3])")
    ap.add_argument("--precision", choices=("f32", "f16x3", "bf16"), default="f32",
                    help="MFMA arithmetic of the timed path (DESIGN.md §4.4)")
    ap.add_argument("--per-op", type=str, default="", help="write the per-op timing table (instrumented replay) here")
    ap.add_argument("--train", action="store_true",
                    help="BASELINE configs[3]: training step (prepare_data, forward, loss, backward, clip, Adam) on the HIP "
                         "training programs, per-GPU batch 6 x 6 s x 8 mics, one flat RCCL gradient all-reduce per step")
    ap.add_argument("--train-two-stage", action="store_true",
                    help="with --train: the two-stage model of train_distributed.py (beam-former + GaGNet post-filter, both on "
                         "their HIP training programs; --gpus N: one flat RCCL all-reduce per stage and step)")
    ap.add_argument("--train-operator-path", action="store_true",
                    help="with --train: forward/backward on PyTorch-ROCm operators (tests/operator_path.py, MIOpen) -- the comparison line")
    ap.add_argument("--train-ddp", action="store_true", help="with --train: wrap in torch DistributedDataParallel (one 64 MB bucket, "
                                                             "gradient_as_bucket_view, static_graph) instead of the flat all-reduce")
    a = ap.parse_args()

    rank, world, local = dist.env_rank()
    if world == 1 and a.gpus > 1:
        # no launcher around us: start the ranks ourselves, as the reference's mp.spawn does
        # (train_distributed.py:363-366) -- fresh processes, one GPU each, before this one makes any GPU call;
        # rank 0's JSON line is the job's.  (Under torch.distributed.run WORLD_SIZE is set and this is skipped.)
        sys.exit(dist.launch_local(a.gpus, [os.path.abspath(__file__)] + sys.argv[1:], timeout=1500.0))
    if world != a.gpus:
        sys.exit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    is_dist = dist.init("nccl", dev)                        # RCCL; only barriers and one scalar MAX
    if a.train:
        return main_train(a, rank, world, dev, is_dist)

    L = int(SECONDS * SR)
    T = 1 + L // HOP
    net, state = make_model(MICS, dev)
    net.precision = a.precision
    net.use_graph = not a.no_graph
    wav = synth_waves(B_PER_GPU, MICS, L, 1234 + rank).to(dev)    # resident in HBM before timing
    window = torch.hann_window(N_FFT)

    def step():
        ns = eabnet_amd.stft_compress(wav, N_FFT, HOP, window)
        return net(ns), ns

    def timed_steps(depth: int, warmup: int, steps: int):
        """`steps` passes wave -> STFT -> EaBNet over the resident batch with `depth` of them in flight
        (eabnet_amd.Pipeline; depth 1 = one after the other), bracketed by barrier + synchronize; returns
        (max-over-ranks seconds, last output)."""
        pipe = eabnet_amd.Pipeline(net, depth=depth, front_end=(N_FFT, HOP, window))
        with torch.no_grad():
            if depth > 1 and not _CALIBRATED:
                pipe.calibrate(wav)                      # untimed: pick streams that actually overlap on this box
                _CALIBRATED.append(True)
            for _ in range(max(warmup, depth)):
                pipe.submit(wav)
                yy = pipe.collect()
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                if pipe.outstanding == depth:
                    yy = pipe.collect()
                pipe.submit(wav)
            while pipe.outstanding:
                yy = pipe.collect()
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
        per_rank.clear()
        per_rank.extend(dist.gather_over_ranks(el, dev))
        return dist.max_over_ranks(el, dev), yy

    per_rank: list = []                   # elapsed seconds of every rank in the most recent timed region
    depth = max(1, a.pipeline)
    elapsed, y = timed_steps(depth, a.warmup, a.steps)
    rank_elapsed = list(per_rank)
    assert torch.isfinite(y).all()
    seq_elapsed = timed_steps(1, 2, a.steps)[0] if depth > 1 else elapsed
    if seq_elapsed < elapsed:
        # streams that happen to share a hardware queue do not overlap; then the plain loop IS the better executor
        # and the headline is its (equally complete) timing of the same K steps
        elapsed, depth, rank_elapsed = seq_elapsed, 1, list(per_rank)
    with torch.no_grad():
        y_seq, ns = step()
    assert torch.equal(y_seq, y), "the pipelined executor must return exactly what net(x) returns"

    def pipelined_results_all_equal(want, n=8) -> bool:
        """untimed: every one of n pipelined results (not only the last, which runs while the pipeline drains)
        equals the direct call's output"""
        pipe = eabnet_amd.Pipeline(net, depth=max(2, depth), front_end=(N_FFT, HOP, window))
        ok = True
        with torch.no_grad():
            for yy in pipe.map([(wav,)] * n):
                ok = ok and bool(torch.equal(yy, want))
        torch.cuda.synchronize()
        return ok
    if a.pipeline > 1:             # (--pipeline 1 = the profiling recipe: one batch in flight, nothing pipelined anywhere)
        assert pipelined_results_all_equal(y_seq), "a pipelined result differs from the direct call (f32)"

    frames = world * B_PER_GPU * T * a.steps
    out = {
        "metric": "enhanced frames/sec (16 kHz, 8-mic) at 1/2/4/8 MI355X; RTF per utterance",
        "value": frames / elapsed, "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if a.precision == "f32" else "f32 storage/accumulate, f16x3 products", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]/[2]: batch of 16 four-second 8-mic 16 kHz utterances per GPU, "
                               "wave -> STFT+compress -> EaBNet.forward, inference, full hand-written HIP path",
                   "batch_per_gpu": B_PER_GPU, "global_batch": world * B_PER_GPU, "mics": MICS, "frames_per_utt": T,
                   "freq_bins": N_FFT // 2 + 1, "parallelism": f"dp{world} (independent shards, no collective)",
                   "pipeline_depth": depth},
        "one_step_at_a_time": {"ms_per_step": 1e3 * seq_elapsed / a.steps, "value": frames / seq_elapsed,
                               "note": "same protocol with a single batch in flight (pipeline depth 1): the latency of "
                                       "a step; the headline keeps `pipeline_depth` batches in flight on separate HIP "
                                       "streams (eabnet_amd.Pipeline), results bit-identical"},
        "rtf_per_utterance": elapsed / a.steps / (B_PER_GPU * SECONDS),
        "ranks": {"world_size_seen_by_torch_distributed": torch.distributed.get_world_size() if is_dist else 1,
                  "backend": "nccl (RCCL)" if is_dist else "none (single process)",
                  "frames_per_s_per_rank": [B_PER_GPU * T * a.steps / e for e in rank_elapsed],
                  "min": B_PER_GPU * T * a.steps / max(rank_elapsed), "max": B_PER_GPU * T * a.steps / min(rank_elapsed)},
        "gflop_per_step_algorithmic": 2e-9 * mac_per_frame(MICS) * B_PER_GPU * T,
    }
    # whole path against the fp32 matrix peak (frames/s x FLOP/frame, summed over the ranks' GPUs)
    ms_step = out["ms_per_step"]
    out["whole_path"] = {"achieved_tflops": out["gflop_per_step_algorithmic"] * world / ms_step,
                         "frac_of_fp32_mfma_peak": out["gflop_per_step_algorithmic"] / ms_step / PEAK_FP32_MFMA_TFLOPS,
                         "one_step_at_a_time_frac": out["gflop_per_step_algorithmic"]
                         / out["one_step_at_a_time"]["ms_per_step"] / PEAK_FP32_MFMA_TFLOPS}

    # secondary measurement: the same timed protocol with the f16x3 contraction mode
    # (DESIGN.md §4.4).  The headline `value` stays on exact-fp32 arithmetic.
    if a.precision == "f32" and not a.no_alt:
        y32 = y.clone()
        net.precision = "f16x3"
        el2, y = timed_steps(depth, max(2, a.warmup), a.steps)
        el2_seq = timed_steps(1, 2, a.steps)[0] if depth > 1 else el2
        with torch.no_grad():
            y_seq2, _ = step()                  # the same batch through a direct (unpipelined) call
        same = bool(torch.equal(y_seq2, y)) and (a.pipeline <= 1 or pipelined_results_all_equal(y_seq2))
        dev_rel = float((y - y32).abs().max() / y32.abs().max())
        if same and dev_rel < 1e-4:
            out["alt_precision"] = {
                "dtype": "f32 storage/accumulate, products as 3 x f16 MFMA on fp16 hi+lo splits (f16x3)",
                "value": frames / el2, "unit": "frames/s", "ms_per_step": 1e3 * el2 / a.steps,
                "one_step_at_a_time_ms_per_step": 1e3 * el2_seq / a.steps,
                "max_rel_deviation_from_f32_mode": dev_rel, "pipelined_equals_direct": same,
                "note": "checked in this run: pipelined output == direct output bit for bit and within 1e-4 of the "
                        "f32 mode's output on the same batch; parity tests: tests/test_hip_parity.py"}
        else:
            # an unverified number is not reported (round-1 shipped one: a cross-wave LDS race in lstm64_h3_kernel)
            out["alt_precision_failed"] = {"pipelined_equals_direct": same, "max_rel_deviation_from_f32_mode": dev_rel,
                                           "note": "f16x3 output failed its in-bench check; no throughput reported"}
        net.precision = "f32"
        with torch.no_grad():
            y, ns = step()                      # rebind the f32 program for the instrumented replay
        torch.cuda.synchronize()

    # Everything below is secondary reporting by rank 0 (the other ranks wait at the final barrier): a failure
    # there must not cost the headline line, so it is recorded instead of raised.
    try:
        if rank == 0 and not a.no_roofline:
            with torch.no_grad():
                ops, ms = instrumented_replay(net, ns, reps=3)

            def valid_flop(o):
                """exact MACs*2 of a conv launch: taps that fall on zero padding are not counted"""
                tot = 0
                oo = np.arange(o.No)
                if o.epi == prg.EPI_PHASE2:
                    # both output-column phases of a transposed convolution in one launch: the phase-0 columns (N / 2) take
                    # every tap, the phase-1 columns only the taps of p2_mask1 and only where the output column 2o+1 exists
                    for j, (dt, io) in enumerate(zip(o.dt, o.ioff)):
                        fi = oo * o.istride + io
                        ok = (fi >= 0) & (fi < o.Fin)
                        tot += max(o.T + dt, 0) * int(ok.sum())
                        if (o.p2_mask1 >> j) & 1:
                            tot += max(o.T + dt, 0) * int((ok & (2 * oo + 1 < o.Fout)).sum())
                    return 2.0 * o.B * tot * (o.N // 2) * (o.C0 + o.C1)
                for dt, io in zip(o.dt, o.ioff):
                    fi = oo * o.istride + io
                    tot += max(o.T + dt, 0) * int(((fi >= 0) & (fi < o.Fin)).sum())
                return 2.0 * o.B * tot * o.N * (o.C0 + o.C1)

            if a.per_op:
                with open(a.per_op, "w") as f:
                    f.write("idx kind ms gflop tflops name geometry\n")
                    for k, o in enumerate(ops):
                        gf, geo = 0.0, ""
                        if o.kind == prg.OP_CONV:
                            gf = valid_flop(o) * 1e-9
                            geo = (f"N={o.N} C={o.C0}+{o.C1} taps={len(o.dt)} Fin={o.Fin} No={o.No} bm={o.bm} xf={o.xf_mode} "
                                   f"epi={o.epi} sets={o.nsets} prec={o.precision} korder={o.korder}")
                        f.write(f"{k} {o.kind} {ms[k]:.4f} {gf:.3f} {gf / max(ms[k], 1e-9):.2f} {o.name} {geo}\n")
            conv = [k for k, o in enumerate(ops) if o.kind == prg.OP_CONV]
            # dominant kernel = the 128x128-tile gated instantiation (GateConv2d / GateConvTranspose2d)
            dom = [k for k in conv if ops[k].epi == prg.EPI_GLU and ops[k].bm == 128 and ops[k].C0 % 4 == 0
                   and ops[k].precision == (prg.PREC_F32 if a.precision == "f32" else prg.PREC_F16X3)]
            dom_ms = float(ms[dom].sum())
            dom_flop = float(sum(valid_flop(ops[k]) for k in dom))
            achieved = dom_flop / (dom_ms * 1e-3) / 1e12
            conv_ms = float(ms[conv].sum())
            by_kind = {}
            for k, o in enumerate(ops):
                nm = {prg.OP_CONV: "conv_gemm", prg.OP_IN_FINALIZE: "in_finalize", prg.OP_NORM_ACT: "norm_act",
                      prg.OP_LSTM64: "lstm64", prg.OP_BFW_FS: "bfw_filter_sum", prg.OP_MEMSET0: "memset"}[o.kind]
                by_kind[nm] = by_kind.get(nm, 0.0) + float(ms[k])
            # the convolution time by block (what VERDICT r02 asks for: S-TCN, 64-column unit convolutions, gated convolutions)
            by_block = {}
            for k in conv:
                o = ops[k]
                blk = ("stcn" if o.name.startswith("stcns.") else "gated_128col" if o.epi == prg.EPI_GLU else "unit_64col")
                e = by_block.setdefault(blk, {"launches": 0, "ms": 0.0, "gflop": 0.0})
                e["launches"] += 1
                e["ms"] += float(ms[k])
                e["gflop"] += valid_flop(o) * 1e-9 + (2e-9 * o.B * o.T * o.f2_N * o.N if o.f2_w is not None else 0.0)
            for e in by_block.values():
                e["tflops"] = e["gflop"] / max(e["ms"], 1e-9)
                e["ms"], e["gflop"] = round(e["ms"], 4), round(e["gflop"], 2)
            peak = PEAK_FP32_MFMA_TFLOPS
            # HBM traffic of the dominant kernel from the committed PMC summary of the same command
            # (tools/prof.sh + tools/summarize_profiles.py; PMC passes are separate runs by necessity).
            # MI355X_MICROARCH.md §HBM: bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- FETCH_SIZE reads
            # half of a wide coalesced read stream on gfx950.
            traffic = None
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", "r04_final", "pmc_summary.json")))
                if pmc.get("_meta", {}).get("kernel_source_sha16") != kernel_source_sha():
                    raise LookupError("the committed PMC summary was taken on other kernel sources")
                tag = "conv_gemm_kernel<2, 2, 1, 1, 0, true, %d, true, false>" % (0 if a.precision == "f32" else 1)
                ent = next(v for k, v in pmc.items() if tag in k)
                c = ent["counters"]
                nd = ent.get("dispatches_of", {})
                traffic = (2.0 * c["FETCH_SIZE"] / nd.get("FETCH_SIZE", ent["dispatches"])
                           + c["WRITE_SIZE"] / nd.get("WRITE_SIZE", ent["dispatches"])) * 1024.0
            except Exception:                                   # noqa: BLE001 - no committed profile: leave null
                traffic = None
            out["roofline"] = {
                "kernel": "conv_gemm_kernel<MI=2,NI=2,KU=1,GLU,XF=0,VEC," + ("f32" if a.precision == "f32" else "f16x3")
                          + "> (128x128-tile gated gather-GEMM convolution)",
                "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                "traffic": traffic, "traffic_unit": "bytes per launch (PMC, profiles/r04_final; null when that profile is of other kernel sources)",
                "launches_per_step": len(dom), "avg_launch_ms": dom_ms / max(len(dom), 1),
                "algorithmic_gflop_per_launch": dom_flop / max(len(dom), 1) / 1e9,
                "share_of_program_time": dom_ms / float(ms.sum()),
                "all_conv_launches": {"launches_per_step": len(conv), "ms": conv_ms,
                                      "achieved_tflops": 2.0 * conv_kernel_mac_per_frame(MICS) * B_PER_GPU * T / (conv_ms * 1e-3) / 1e12},
                "program_ms_by_kernel": {k: round(v, 4) for k, v in sorted(by_kind.items(), key=lambda kv: -kv[1])},
                "conv_ms_by_block": by_block,
                "program_ms_total": float(ms.sum()),
                "note": "achieved = exact valid-tap FLOPs of the dominant instantiation's launches / their summed "
                        "duration (HIP events per op on the launch stream, instrumented replay after the timed region: every "
                        "op's figure includes ~5 us of event overhead, which inflates the many short launches of conv_ms_by_block); "
                        "peak = fp32 MFMA dense (MI355X_MICROARCH.md); traffic: see profiles/ (PMC passes are separate runs)",
            }
        if rank == 0 and not a.no_roofline:
            # HBM-side kernels of the path (SURVEY §8d asks for both fractions): algorithmic bytes / time
            def timed(fn, reps=10):
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) / reps * 1e-3
            with torch.no_grad():
                t_stft = timed(lambda: eabnet_amd.stft_compress(wav, N_FFT, HOP, window))
                wts = torch.randn_like(ns)
                t_fs = timed(lambda: eabnet_amd.filter_and_sum(wts, ns))
                est = torch.randn(B_PER_GPU, 2, T, 161, device=dev)
                t_is = timed(lambda: eabnet_amd.istft(est, N_FFT, HOP, window))
            fr = B_PER_GPU * T
            by_stft = fr * (160 * MICS * 4 + 161 * MICS * 2 * 4)
            by_fs = fr * 161 * (4 * MICS + 2) * 4
            by_is = fr * (2 * 161 + 160) * 4
            by_na = fr * 3 * 64 * 4 * sum(o.P for o in ops if o.kind == prg.OP_NORM_ACT and o.b is not None) / T \
                + fr * 2 * 64 * 4 * sum(o.P for o in ops if o.kind == prg.OP_NORM_ACT and o.b is None) / T
            t_na = 1e-3 * sum(float(ms[k]) for k, o in enumerate(ops) if o.kind == prg.OP_NORM_ACT)
            out["hbm_kernels"] = {
                "peak_GBs": PEAK_HBM_GBS,
                "stft_compress": {"us": 1e6 * t_stft, "achieved_GBs": by_stft / t_stft / 1e9, "frac": by_stft / t_stft / 1e9 / PEAK_HBM_GBS,
                                  "bytes_per_frame": 160 * MICS * 4 + 161 * MICS * 2 * 4},
                "filter_sum": {"us": 1e6 * t_fs, "achieved_GBs": by_fs / t_fs / 1e9, "frac": by_fs / t_fs / 1e9 / PEAK_HBM_GBS,
                               "bytes_per_frame": 161 * (4 * MICS + 2) * 4},
                "istft(back end, next row N2)": {"us": 1e6 * t_is, "achieved_GBs": by_is / t_is / 1e9,
                                                 "frac": by_is / t_is / 1e9 / PEAK_HBM_GBS, "bytes_per_frame": (2 * 161 + 160) * 4},
                "norm_act(all 10 launches)": {"us": 1e6 * t_na, "achieved_GBs": by_na / t_na / 1e9, "frac": by_na / t_na / 1e9 / PEAK_HBM_GBS},
            }

        if rank == 0 and not a.no_c1:
            # BASELINE configs[0] shape on the GPU: ONE 4-s 8-mic utterance, wave -> output, latency and RTF
            wav1 = wav[:1].contiguous()
            c1 = {}
            for prec in (("f32", "f16x3") if (a.precision == "f32" and not a.no_alt) else (a.precision,)):
                net.precision = prec
                with torch.no_grad():
                    for _ in range(3):
                        y1 = net(eabnet_amd.stft_compress(wav1, N_FFT, HOP, window))
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(20):
                        y1 = net(eabnet_amd.stft_compress(wav1, N_FFT, HOP, window))
                    torch.cuda.synchronize()
                    dt1 = (time.perf_counter() - t0) / 20
                    # independent single-utterance requests, two in flight (throughput of a request server)
                    p1 = eabnet_amd.Pipeline(net, depth=2, front_end=(N_FFT, HOP, window))
                    for _ in range(4):
                        p1.submit(wav1)
                        p1.collect()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(40):
                        if p1.outstanding == 2:
                            p1.collect()
                        p1.submit(wav1)
                    while p1.outstanding:
                        p1.collect()
                    torch.cuda.synchronize()
                    dt2 = (time.perf_counter() - t0) / 40
                    p1 = None
                c1[prec] = {"ms_per_utterance": 1e3 * dt1, "rtf": dt1 / SECONDS, "frames_per_s": T / dt1,
                            "two_in_flight_ms_per_utterance": 1e3 * dt2, "two_in_flight_frames_per_s": T / dt2}
            net.precision = a.precision
            out["single_utterance_c1"] = c1

        if rank == 0 and not a.no_next:
            # the boundary as the reference drives it: the wave arrives in HOST memory and prepare_data moves it
            # (train_distributed.py:76-77) -- the PCIe-inclusive rate, never the headline
            pd_args = argparse.Namespace(mics=MICS, sr=SR, wav_len=SECONDS, win_size=0.020, win_shift=0.010, fft_num=N_FFT)
            net.precision = a.precision
            host = {"pageable": wav.cpu(), "pinned": wav.cpu().pin_memory()}
            pc = {}
            depth = max(1, a.pipeline)
            for kind, hw in host.items():
                tg = hw[:, :1].contiguous()
                if kind == "pinned":
                    tg = tg.pin_memory()
                with torch.no_grad():
                    # (a) as the reference's loop does it: one step at a time, blocking copies (train_distributed.py:76-77)
                    for _ in range(3):
                        net(eabnet_amd.prepare_data(hw, tg, dev, pd_args)[0])
                    torch.cuda.synchronize()
                    rounds = []
                    for _ in range(3):                      # the host-side memcpy into the staging ring is noisy on a shared box
                        t0 = time.perf_counter()
                        for _ in range(10):
                            net(eabnet_amd.prepare_data(hw, tg, dev, pd_args)[0])
                        torch.cuda.synchronize()
                        rounds.append((time.perf_counter() - t0) / 10)
                    # (b) the same work per step (both uploads, both STFTs, the network) with `depth` batches in flight:
                    # eabnet_amd.Pipeline(prepare=args) stages batch k+1 (pinned ring, copy kernel on its own stream) while
                    # batch k computes -- the protocol of the headline, with the input in HOST memory
                    pipe = eabnet_amd.Pipeline(net, depth=depth, prepare=pd_args)
                    for _ in range(depth + 2):
                        pipe.submit(hw, tg)
                        pipe.collect()
                    torch.cuda.synchronize()
                    prounds = []
                    for _ in range(3):
                        t0 = time.perf_counter()
                        for _ in range(12):
                            if pipe.outstanding == depth:
                                pipe.collect()
                            pipe.submit(hw, tg)
                        while pipe.outstanding:
                            y_p, _t = pipe.collect()
                        torch.cuda.synchronize()
                        prounds.append((time.perf_counter() - t0) / 12)
                    y_b = net(eabnet_amd.prepare_data(hw, tg, dev, pd_args)[0])
                    same = bool(torch.equal(y_p, y_b))
                    pipe = None
                dth = sorted(rounds)[len(rounds) // 2]       # the median round (the best one flattered the path in round 2)
                dtp = sorted(prounds)[len(prounds) // 2]
                pc[kind] = {"ms_per_step": 1e3 * dtp, "frames_per_s": B_PER_GPU * T / dtp, "batches_in_flight": depth,
                            "ms_per_step_rounds": [round(1e3 * r, 3) for r in prounds],
                            "pipelined_equals_blocking": same,
                            "blocking_ms_per_step": 1e3 * dth, "blocking_frames_per_s": B_PER_GPU * T / dth,
                            "blocking_ms_per_step_rounds": [round(1e3 * r, 3) for r in rounds]}
            pc["note"] = (f"host wave ({wav.numel() * 4 / 1e6:.1f} MB per step) -> prepare_data (noisy + target STFT) -> EaBNet.  "
                          f"ms_per_step: {depth} batches in flight through eabnet_amd.Pipeline(prepare=args) -- the upload of batch "
                          "k+1 (pinned staging ring + copy kernel on its own stream) runs under batch k's compute; "
                          "blocking_*: one step at a time with blocking copies, as train_distributed.py:76-77 drives it")
            out.setdefault("next_rows", {})["pcie_inclusive"] = pc

        if rank == 0 and not a.no_next:
            # SURVEY §8f rows built after the hot path: what enhance.py actually runs -- wave -> STFT ->
            # EaBNetWithPostNet (beam-former + GaGNet post-filter) -> ISTFT -> wave, same batch, same protocol
            net = None
            torch.cuda.empty_cache()
            pa = argparse.Namespace(
                k1=(2, 3), k2=(1, 3), c=64, M=MICS, embed_dim=64, kd1=5, cd1=64, d_feat=256, p=6, q=3, is_causal=True,
                is_u2=True, bf_type="lstm", topo_type="mimo", intra_connect="cat", norm_type="IN", ref_mic=0,
                freeze_eabnet=False, gagnet_k1=(2, 3), gagnet_k2=(1, 3), gagnet_c=64, gagnet_kd1=3, gagnet_cd1=64,
                gagnet_d_feat=256, gagnet_p=2, gagnet_q=3, gagnet_dilas=[1, 2, 5, 9], gagnet_fft_num=320, gagnet_is_u2=True,
                gagnet_is_causal=True, gagnet_is_squeezed=False, gagnet_acti_type="sigmoid", gagnet_intra_connect="cat",
                gagnet_norm_type="IN")
            torch.manual_seed(1)
            two = eabnet_amd.make_eabnet_with_postnet(pa).to(dev).eval()
            two.eabnet.load_state_dict(state, strict=True)
            nxt = {"pipeline": "wave -> stft_compress -> EaBNetWithPostNet -> istft -> wave (enhance.py:45-62)",
                   "params": eabnet_amd.numParams(two)}
            for prec in (("f32", "f16x3") if not a.no_alt else ("f32",)):
                two.eabnet.precision = two.postnet.precision = prec

                def step():
                    o = two(eabnet_amd.stft_compress(wav, N_FFT, HOP, window))
                    return eabnet_amd.istft(o["esti_stft"], N_FFT, HOP, window)
                with torch.no_grad():
                    for _ in range(3):
                        step()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(10):
                        w_out = step()
                    torch.cuda.synchronize()
                    dtw = (time.perf_counter() - t0) / 10
                    inp = two.postnet._last[1:]
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10):
                        two.postnet(*inp)
                    e1.record()
                    torch.cuda.synchronize()
                    # the same with two batches in flight (eabnet_amd.Pipeline over the two-stage model)
                    pipe2 = eabnet_amd.Pipeline(two, depth=2, front_end=(N_FFT, HOP, window))
                    for _ in range(3):
                        pipe2.submit(wav)
                        eabnet_amd.istft(pipe2.collect()["esti_stft"], N_FFT, HOP, window)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(12):
                        if pipe2.outstanding == 2:
                            w_out = eabnet_amd.istft(pipe2.collect()["esti_stft"], N_FFT, HOP, window)
                        pipe2.submit(wav)
                    while pipe2.outstanding:
                        w_out = eabnet_amd.istft(pipe2.collect()["esti_stft"], N_FFT, HOP, window)
                    torch.cuda.synchronize()
                    dtp2 = (time.perf_counter() - t0) / 12
                    pipe2 = None
                assert bool(torch.isfinite(w_out).all())
                nxt[prec] = {"ms_per_step": 1e3 * dtw, "frames_per_s": B_PER_GPU * T / dtw,
                             "two_in_flight_ms_per_step": 1e3 * dtp2, "two_in_flight_frames_per_s": B_PER_GPU * T / dtp2,
                             "postfilter_ms_per_step": e0.elapsed_time(e1) / 10}
            # BASELINE config 5 shape: 16 microphones, 8-s utterance, frame-synchronous (BatchNorm norms, causal):
            # latency of one step of `chunk` 10-ms frames = one hipGraph replay of the windowed program
            torch.manual_seed(2)
            sn = eabnet_amd.EaBNet(M=16, norm_type="BN").to(dev).eval()
            stream_rows = {"config": "B=1, M=16, T_max=801 (8 s), norm_type=BN; fp32 and (suffix _bf16) bf16 products", "hop_ms": 10.0}
            for chunk, sprec in ((1, "f32"), (16, "f32"), (1, "bf16"), (16, "bf16")):
                sn.precision = sprec
                st = sn.stream_begin(1, T_max=801, chunk=chunk)
                xs = 0.3 * torch.randn(1, chunk, 161, 16, 2, device=dev)
                for _ in range(3):
                    st.step(xs)
                torch.cuda.synchronize()
                nstep = min(40, (801 - 3 * chunk) // chunk)
                t0 = time.perf_counter()
                for _ in range(nstep):
                    ys = st.step(xs)
                torch.cuda.synchronize()
                dts = (time.perf_counter() - t0) / nstep
                assert bool(torch.isfinite(ys).all())
                stream_rows[f"chunk{chunk}" + ("" if sprec == "f32" else "_bf16")] = {"ms_per_step": 1e3 * dts, "rtf": dts / (chunk * 0.010),
                                                "algorithmic_latency_ms": 10.0 * chunk + 10.0}
                st = None
            nxt["streaming"] = stream_rows
            sn = None
            # the same shape with the causal norm the reference means to offer (norm_type="cLN", fixed constructor): running
            # sums are the only norm state; statistics and normalisation are separate ops here, hence more launches per step
            torch.manual_seed(3)
            sc = eabnet_amd.EaBNet(M=16, norm_type="cLN").to(dev).eval()
            cln_rows = {"config": "B=1, M=16, T_max=801 (8 s), norm_type=cLN; fp32 and (suffix _bf16) bf16 products", "hop_ms": 10.0}
            for chunk, sprec in ((1, "f32"), (16, "f32"), (1, "bf16")):
                sc.precision = sprec
                st = sc.stream_begin(1, T_max=801, chunk=chunk)
                xs = 0.3 * torch.randn(1, chunk, 161, 16, 2, device=dev)
                for _ in range(3):
                    st.step(xs)
                torch.cuda.synchronize()
                nstep = min(40, (801 - 3 * chunk) // chunk)
                t0 = time.perf_counter()
                for _ in range(nstep):
                    ys = st.step(xs)
                torch.cuda.synchronize()
                dts = (time.perf_counter() - t0) / nstep
                assert bool(torch.isfinite(ys).all())
                cln_rows[f"chunk{chunk}" + ("" if sprec == "f32" else "_bf16")] = {"ms_per_step": 1e3 * dts, "rtf": dts / (chunk * 0.010)}
                st = None
            nxt["streaming_cln"] = cln_rows
            sc = None
            # the same, wave in -> wave out through the two-stage model (StreamingEnhancer: STFT windows, streamed
            # beam-former + post-filter, ISTFT windows), one 10-ms hop of 16-microphone samples per push
            pb = argparse.Namespace(**{**vars(pa), "M": 16, "norm_type": "BN", "gagnet_norm_type": "BN"})
            tw = eabnet_amd.make_eabnet_with_postnet(pb).to(dev).eval()
            enh = eabnet_amd.StreamingEnhancer(tw, B=1, seconds=8.0, chunk=1)
            hop_samples = 0.05 * torch.randn(1, 16, 160, device=dev)
            for _ in range(5):
                enh.push(hop_samples)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(40):
                wv = enh.push(hop_samples)
            torch.cuda.synchronize()
            dtp = (time.perf_counter() - t0) / 40
            assert wv.shape == (1, 160) and bool(torch.isfinite(wv).all())
            nxt["streaming_wave_to_wave"] = {"config": "B=1, M=16, two-stage model with BN norms, one hop (160 samples) per push",
                                             "ms_per_push": 1e3 * dtp, "rtf": dtp / 0.010,
                                             "latency_ms": 20.0 + 1e3 * dtp}
            tw = enh = None
            out.setdefault("next_rows", {}).update(nxt)
            two = None
            torch.cuda.empty_cache()

        if rank == 0 and world == 1 and not a.no_next and not a.no_train:
            # BASELINE configs[3] inside the driver-timed line: the training step of the beam-former and of the two-stage model
            # (train_distributed.py:181,214-230; batch 6 x 6 s x 8 mics, Adam, clip, loss) on the HIP training programs, exact
            # fp32 and bf16 products.  Each configuration first passes an in-run check (loss against the PyTorch-ROCm operator
            # path on the same parameters and batch, all gradients finite); a configuration that fails it reports no number.
            net = None
            import gc
            gc.collect()            # nets of the earlier sections: their bound programs drain the device when they are finalised
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            trn = {"config": "per-GPU batch 6 x 6 s x 8 mics (T = 601): prepare_data (noisy + target STFT), forward, loss, backward, "
                             "clip_grad_norm_(1.0), Adam(5e-4) inside every timed step; 5 timed steps after 2 warm-up steps; "
                             "`bench.py --train [--train-two-stage] [--precision bf16] [--gpus N]` runs one configuration alone"}
            for two_ in (False, True):
                for prec_ in ("f32", "bf16"):
                    key = ("two_stage" if two_ else "beam_former") + "_" + prec_
                    t_cfg = time.perf_counter()
                    try:
                        r = train_measure(0, 1, dev, False, precision=prec_, two_stage=two_, steps=5, warmup=2, roofline=not two_,
                                          check=True)
                        if r.get("failed"):
                            trn[key] = {"dropped": "in-run check failed; no throughput reported", "check": r["check"]}
                        else:
                            keep = ("ms_per_step", "value", "unit", "whole_step_tflops", "final_loss", "check", "roofline", "dgrad",
                                    "gflop_per_step", "workspace_GB", "params")
                            trn[key] = {k: r[k] for k in keep if k in r}
                            if "programs" in r:
                                trn[key]["program_ms_by_kernel"] = {w: r["programs"][w]["ms_by_kernel"] for w in r["programs"]}
                    except Exception as e:  # noqa: BLE001
                        trn[key] = {"error": repr(e)}
                    trn[key]["bench_seconds"] = round(time.perf_counter() - t_cfg, 1)
                    torch.cuda.empty_cache()
            out.setdefault("next_rows", {})["training"] = trn

        if rank == 0 and world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(state, MICS, L)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    except Exception as e:  # noqa: BLE001
        out["secondary_sections_error"] = repr(e)
    if rank == 0:
        print(json.dumps(out))
    if is_dist:
        dist.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
