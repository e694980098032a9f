"""Replay of op programs with parallel branches as a set of SINGLE-STREAM hipGraphs.

Programs mark independent chains (the post-filter's glance / gaze S-TCM chains, forward and backward) with fork / join
points and a stream lane per op (program.Program.lanes / .sync, train.TrainProgram.lanes / .sync).  Rounds 1-3 captured such
a program into ONE hipGraph with internal branches and let the HIP runtime pick the streams of a replay.  That ends in a
segmentation fault inside the runtime under a condition the caller cannot see (ROCm 7.2,
profiles/r03_graph_branch_segfault_backtrace.txt; DESIGN.md section 7 has the disassembly):

    hip::Graph::UpdateStreams(launch_stream, parallel_streams)          libamdhip64.so + 0xaed90
        streams_.resize(max_streams_); streams_[0] = launch_stream;
        for (i = 1, j = 0; i < streams_.size(); ++j) {
            if (parallel_streams[j]->vdev()->f() == launch_stream->vdev()->f()) continue;   // <- crash: + 0xb1, j unchecked
            streams_[i++] = parallel_streams[j];
        }

A parallel stream of the graph that maps to the same device queue as the launch stream is skipped, and nothing bounds j by
parallel_streams.size(): when more of the graph's own streams collide with the launch stream than the runtime created spares
for, the loop reads past the vector and dereferences what it finds.  Which queue a stream maps to depends on every stream the
process created and destroyed before (RCCL communicators create and destroy their own) -- not on anything in this package.
A graph captured on ONE stream has max_streams_ == 1 and never enters that loop.

So the branches are kept out of the runtime's hands: every maximal run of ops on one lane between two fork / join marks is
captured as its own single-stream hipGraph, and a replay launches those graphs on streams this object owns, with ordinary
events for the forks and joins.  Same kernels, same order per lane, same concurrency between lanes; 13 graph launches
instead of 1 for the default post-filter."""
from __future__ import annotations

import threading
from typing import Callable, Dict, List, Sequence, Tuple

import torch

# plan entries
FORK, JOIN, RUN = "fork", "join", "run"


def plan_segments(n_ops: int, lanes: Sequence[int], sync: Dict[int, list]) -> List[tuple]:
    """The replay plan of a program: a list of ("fork", [lanes]) / ("join", [lanes]) / ("run", lane, first, count).
    `sync[k]` applies BEFORE op k (k == n_ops: after the last op); a run never crosses a mark or a lane change.
    Pure Python: tested on a machine without a GPU."""
    if len(lanes) != n_ops:
        raise ValueError("one lane per op")
    open_lanes = set()
    plan: List[tuple] = []
    k = 0
    while k <= n_ops:
        for what, ls in sync.get(k, ()):
            ls = [int(l) for l in ls]
            if what == FORK:
                if 0 in ls or open_lanes & set(ls):
                    raise ValueError(f"fork of lane 0 or of an open lane at op {k}")
                open_lanes |= set(ls)
            elif what == JOIN:
                if not set(ls) <= open_lanes:
                    raise ValueError(f"join of a lane that was not forked at op {k}")
                open_lanes -= set(ls)
            else:
                raise ValueError(what)
            plan.append((what, ls))
        if k == n_ops:
            break
        j = k + 1
        while j < n_ops and lanes[j] == lanes[k] and j not in sync:
            j += 1
        if lanes[k] != 0 and lanes[k] not in open_lanes:
            raise ValueError(f"op {k} runs on lane {lanes[k]} outside a fork / join pair")
        plan.append((RUN, int(lanes[k]), k, j - k))
        k = j
    if open_lanes:
        raise ValueError(f"lanes {sorted(open_lanes)} are never joined")
    return plan


def single_lane(n_ops: int) -> List[tuple]:
    return [(RUN, 0, 0, n_ops)] if n_ops else []


# Side streams that really run beside the launch stream.  HIP maps streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by
# default) in creation order, and two streams on one queue execute back to back: a freshly created side stream overlaps with the
# launch stream -- or not -- depending on every stream the process created before (measured: the post-filter's three chains per
# streamed hop, 2.11 ms with distinct queues, 2.64 ms = the single-lane time without).  The runtime's own branch replay picked
# distinct queues itself (that is what the loop quoted above is for); with the streams in our hands we do: candidates are timed
# pairwise with a spin kernel, once per (device, launch stream), and the first that overlap with the launch stream AND with
# each other are kept for every program of the process.
_SIDE_POOL: Dict[Tuple[str, int], List[torch.cuda.Stream]] = {}
_POOL_LOCK = threading.Lock()      # (two threads binding programs at once would otherwise probe -- and create streams -- twice)
_PROBE_CYCLES = 400_000        # torch.cuda._sleep: ~0.2 ms


def _overlap(a: torch.cuda.Stream, b: torch.cuda.Stream, device: torch.device) -> float:
    """seconds for one spin kernel on each of two streams, started together"""
    import time
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for st in (a, b):
        with torch.cuda.stream(st):
            torch.cuda._sleep(_PROBE_CYCLES)
    a.synchronize()
    b.synchronize()
    return time.perf_counter() - t0


def _pick(device: torch.device, n: int, fixed: List[torch.cuda.Stream], pool: List[torch.cuda.Stream], priority: int = 0) -> None:
    """extend `pool` to n streams that overlap with every stream of `fixed` and with each other (best effort: when fewer
    such streams are found among the candidates the rest are ordinary new streams -- correctness never depends on overlap)"""
    def make():
        return torch.cuda.Stream(device=device, priority=priority)
    if not hasattr(torch.cuda, "_sleep") or torch.cuda.is_current_stream_capturing():
        pool += [make() for _ in range(n - len(pool))]
        return
    with torch.cuda.device(device):
        ref = fixed[0] if fixed else torch.cuda.current_stream(device)
        _overlap(ref, torch.cuda.Stream(device=device), device)             # (first launch: module load)
        alone = min(_overlap(ref, ref, device) for _ in range(2)) / 2.0      # one spin kernel
        cands = [make() for _ in range(8)]
        for c in cands:
            if len(pool) >= n:
                break
            if all(_overlap(o, c, device) < 1.5 * alone for o in fixed + pool):
                pool.append(c)
        spare = [c for c in cands if c not in pool]
        pool += spare[:max(0, n - len(pool))]


def side_streams(device: torch.device, n: int) -> List[torch.cuda.Stream]:
    """n streams that overlap with the current stream of `device` and with each other."""
    if n <= 0:
        return []
    main = torch.cuda.current_stream(device)
    key = (str(device), main.cuda_stream)
    with _POOL_LOCK:
        if key not in _SIDE_POOL and sum(1 for k in _SIDE_POOL if k[0] == key[0]) >= 4:
            key = next(k for k in _SIDE_POOL if k[0] == key[0])             # many launch streams: stop creating streams, reuse
        pool = _SIDE_POOL.setdefault(key, [])
        if len(pool) < n:
            _pick(device, n, [main], pool)
        return pool[:n]


def overlapping_streams(device: torch.device, n: int, beside: Sequence[torch.cuda.Stream] = (), priority: int = 0) -> List[torch.cuda.Stream]:
    """n new streams that overlap with each other and with every stream of `beside` (the executor's batches in flight and the
    upload stream next to them, model.Pipeline / model._HostStager)."""
    pool: List[torch.cuda.Stream] = []
    _pick(device, n, list(beside), pool, priority)
    return pool[:n]


class LaneGraphs:
    """The captured form of one program: one single-stream hipGraph per ("run", ...) entry of the plan.

    launch(stream_ptr, first, count) must enqueue ops [first, first + count) on the raw stream it is given and nothing
    else (no allocation, no other stream)."""

    def __init__(self, device: torch.device, plan: List[tuple], launch: Callable[[int, int, int], None]):
        self.device, self.plan, self.launch = device, plan, launch
        self.side: Dict[int, torch.cuda.Stream] = {}
        self.graphs: Dict[Tuple[int, int], torch.cuda.CUDAGraph] = {}
        lanes = []
        for e in plan:                          # every lane a run or a mark names (a forked lane may carry no op)
            for l in ([e[1]] if e[0] == RUN else e[1]):
                if l != 0 and l not in lanes:
                    lanes.append(l)
        self.side = dict(zip(lanes, side_streams(device, len(lanes))))

    @property
    def n_graphs(self) -> int:
        return len(self.graphs)

    def run_direct(self) -> None:
        """The plan with direct kernel launches (warm-up; programs that must not be captured)."""
        self._walk(lambda lane, first, count, st: self.launch(st.cuda_stream, first, count))

    def capture(self) -> None:
        """Capture every run on the capture stream torch provides: nothing crosses streams inside a capture, so every
        graph is a single-stream graph.  thread_local: other threads of the process (RCCL's watchdog, a data loader) may
        issue HIP calls while this thread captures; in the default global mode such a call invalidates the capture."""
        for e in self.plan:
            if e[0] != RUN:
                continue
            _, lane, first, count = e
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self.launch(torch.cuda.current_stream().cuda_stream, first, count)
            self.graphs[(first, count)] = g

    def replay(self) -> None:
        self._walk(lambda lane, first, count, st: self.graphs[(first, count)].replay())

    def _walk(self, do) -> None:
        main = torch.cuda.current_stream(self.device)
        for e in self.plan:
            if e[0] == FORK:
                for l in e[1]:
                    self.side[l].wait_stream(main)
            elif e[0] == JOIN:
                for l in e[1]:
                    main.wait_stream(self.side[l])
            else:
                _, lane, first, count = e
                if lane == 0:
                    do(lane, first, count, main)
                else:
                    with torch.cuda.stream(self.side[lane]):
                        do(lane, first, count, self.side[lane])
