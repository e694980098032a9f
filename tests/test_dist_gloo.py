"""N > 1 path on CPU: two gloo ranks shard a set of utterances the way bench.py
shards them over GPUs, and the gathered result equals the single-process one
(utterances are independent, so no data-path collective is needed)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import paramgen
from eabnet_amd import dist
from eabnet_amd.spec import NetConfig, param_specs


def test_shard_partitions_exactly_once():
    for n in (0, 1, 7, 16, 17):
        for world in (1, 2, 3, 8):
            seen = [i for r in range(world) for i in dist.shard(n, r, world)]
            assert seen == list(range(n))
            sizes = [len(dist.shard(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_utt, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from oracle import eabnet_oracle as orc           # checker standing in for the device path on CPU
    assert dist.init("gloo")
    cfg = NetConfig(M=2, p=1, q=1)
    P = {k: torch.from_numpy(v) for k, v in paramgen.make_params(param_specs(cfg), 7).items()}
    x = torch.from_numpy(paramgen.make_spec_input(n_utt, 6, 161, 2, 8))
    mine = dist.shard(n_utt, rank, world)
    dist.barrier()
    with torch.no_grad():
        y = orc.eabnet_forward(P, x[mine.start:mine.stop], p=1, q=1)
    dist.barrier()
    t = dist.max_over_ranks(1.0 + rank)
    loss = dist.mean_over_ranks(torch.tensor(float(rank)))
    q.put((rank, mine.start, y.numpy(), t, float(loss)))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharded_inference_matches_single_process():
    from oracle import eabnet_oracle as orc
    world, n_utt = 2, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_utt, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = NetConfig(M=2, p=1, q=1)
    P = {k: torch.from_numpy(v) for k, v in paramgen.make_params(param_specs(cfg), 7).items()}
    x = torch.from_numpy(paramgen.make_spec_input(n_utt, 6, 161, 2, 8))
    with torch.no_grad():
        ref = orc.eabnet_forward(P, x, p=1, q=1).numpy()
    out = np.zeros_like(ref)
    for rank, start, y, t, loss in got:
        out[start:start + y.shape[0]] = y
        assert t == 2.0                  # MAX over ranks of (1 + rank)
        assert abs(loss - 0.5) < 1e-12   # SUM / world_size
    np.testing.assert_allclose(out, ref, rtol=0, atol=2e-6 * np.abs(ref).max())


def _ddp_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    torch.manual_seed(0)
    import eabnet_amd
    from torch.nn.parallel import DistributedDataParallel as DDP
    assert dist.init("gloo")
    from operator_path import OperatorPath
    net = eabnet_amd.EaBNet(M=2, p=1, q=1)
    # reference: DDP(net, device_ids=[device]) train_distributed.py:198.  CPU ranks have no HIP programs: the differentiable
    # forward is the PyTorch-operator comparator (tests/operator_path.py) on the module's own parameters -- what is under test
    # here is the distributed plumbing (hooks, bucket all-reduce, identical replicas), which does not care
    ddp = DDP(OperatorPath(net))
    opt = torch.optim.Adam(ddp.parameters(), lr=5e-4)
    losses = []
    for it in range(2):
        x = torch.from_numpy(paramgen.make_spec_input(1, 5, 161, 2, 200 + 10 * it + rank))     # rank-specific shard
        label = torch.from_numpy(paramgen.make_spec_input(1, 5, 161, 1, 300 + 10 * it + rank)[..., 0, :]).permute(0, 3, 1, 2)
        opt.zero_grad()
        out = ddp(x)
        loss = eabnet_amd.com_mag_mse_loss(out, label, [5])
        loss.backward()                                # gradient all-reduce happens here
        torch.nn.utils.clip_grad_norm_(ddp.parameters(), 1.0)
        opt.step()
        losses.append(float(dist.mean_over_ranks(loss.detach())))
    g = torch.cat([p.grad.flatten() for p in net.parameters()])
    w = torch.cat([p.detach().flatten() for p in net.parameters()])
    q.put((rank, losses, float(g.double().norm()), float(w.double().sum())))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_ddp_training_step():
    """BASELINE config 4 on CPU ranks: DDP gradient all-reduce keeps the replicas identical."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=500) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, l0, g0, w0), (_, l1, g1, w1) = got
    assert l0 == l1 and all(np.isfinite(l0))          # the averaged loss is the same number on both ranks
    assert abs(g0 - g1) <= 1e-9 * max(g0, 1.0)        # all-reduced gradients are identical
    assert abs(w0 - w1) <= 1e-9 * max(abs(w0), 1.0)   # so are the updated parameters


@pytest.mark.timeout(300)
def test_self_launcher_two_ranks():
    """bench.py --gpus N without torchrun: dist.launch_local starts N fresh processes with the
    torch.distributed.run environment; only rank 0's line reaches stdout; a failing rank fails the job."""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys; sys.path.insert(0, %r); from eabnet_amd import dist; "
            "sys.exit(dist.launch_local(2, [%r] + sys.argv[1:], timeout=240))") % (os.path.dirname(here), os.path.join(here, "_launch_worker.py"))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip() and not l.startswith("[Gloo]")]   # gloo's own banner
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out == {"world": 2, "local": 0, "per_rank": [10.0, 11.0], "max": 11.0, "master": "127.0.0.1"}
    r = subprocess.run([sys.executable, "-c", code, "fail"], capture_output=True, text=True, env=env, timeout=280)
    assert r.returncode == 3 and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def _flat_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    assert dist.init("gloo")
    from eabnet_amd.train import finish_flat_gradient
    shapes = [(3, 2), (), (4,), (2, 1, 2)]
    dtypes = [torch.float32, torch.float32, torch.float64, torch.float32]
    needs = [True, True, True, False]                      # a frozen parameter gets None
    n = sum(int(np.prod(s)) if s else 1 for s in shapes)
    g = torch.arange(n, dtype=torch.float32) * (rank + 1) + 100.0 * rank     # known per-rank gradient
    plain = finish_flat_gradient(g.clone(), None, shapes, dtypes, needs)     # no synchronisation: the rank's own values
    synced = finish_flat_gradient(g.clone(), True, shapes, dtypes, needs)    # default group: the mean over the ranks
    q.put((rank, [None if t is None else t.double().numpy() for t in plain],
           [None if t is None else (t.double().numpy(), str(t.dtype)) for t in synced]))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_flat_gradient_allreduce_on_two_gloo_ranks():
    """The post-processing of the training autograd nodes (eabnet_amd.train.finish_flat_gradient: one all-reduce of the flat
    gradient, average, one view per parameter, None for frozen parameters) on two CPU ranks with known per-rank values."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_flat_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = q.get(timeout=240)
        res[r[0]] = r
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    shapes = [(3, 2), (), (4,), (2, 1, 2)]
    n = 6 + 1 + 4 + 4
    per_rank = [np.arange(n, dtype=np.float64) * (r + 1) + 100.0 * r for r in range(world)]
    mean = sum(per_rank) / world
    for r in range(world):
        _, plain, synced = res[r]
        off = 0
        for k, shp in enumerate(shapes):
            cnt = int(np.prod(shp)) if shp else 1
            if k == 3:
                assert plain[k] is None and synced[k] is None
            else:
                assert np.array_equal(plain[k].reshape(-1), per_rank[r][off:off + cnt]) and plain[k].shape == shp
                val, dt = synced[k]
                assert np.allclose(val.reshape(-1), mean[off:off + cnt], rtol=0, atol=1e-5) and val.shape == shp
                assert dt == ("torch.float64" if k == 2 else "torch.float32")
            off += cnt
    assert all(np.array_equal(a[0], b[0]) for a, b in zip(res[0][2][:3], res[1][2][:3]))     # identical on both ranks
