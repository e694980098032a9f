"""CPU tests of the replay plan for programs with parallel branches (eabnet_amd/graphs.py): every op is launched exactly
once, in program order within its lane, every lane runs between its fork and its join, and a plan never needs more than
single-stream captures (one graph per run)."""
import numpy as np
import pytest

import paramgen
from eabnet_amd import graphs, program as prg, train_gag
from eabnet_amd.spec import GagConfig, gag_param_specs


def _check_plan(plan, n_ops, lanes):
    seen = []
    open_lanes = set()
    for e in plan:
        if e[0] == graphs.FORK:
            assert not (set(e[1]) & open_lanes) and 0 not in e[1]
            open_lanes |= set(e[1])
        elif e[0] == graphs.JOIN:
            assert set(e[1]) <= open_lanes
            open_lanes -= set(e[1])
        else:
            _, lane, first, count = e
            assert count > 0 and all(lanes[k] == lane for k in range(first, first + count))
            assert lane == 0 or lane in open_lanes, "a side lane runs only between its fork and its join"
            seen += list(range(first, first + count))
    assert seen == list(range(n_ops)), "every op exactly once, in program order"
    assert not open_lanes


def test_plan_of_a_hand_written_program():
    lanes = [0, 0, 0, 1, 1, 2, 0, 0]
    sync = {2: [("fork", [1, 2])], 6: [("join", [1, 2])]}
    plan = graphs.plan_segments(8, lanes, sync)
    assert plan == [("run", 0, 0, 2), ("fork", [1, 2]), ("run", 0, 2, 1), ("run", 1, 3, 2), ("run", 2, 5, 1),
                    ("join", [1, 2]), ("run", 0, 6, 2)]
    _check_plan(plan, 8, lanes)
    assert graphs.single_lane(5) == [("run", 0, 0, 5)] and graphs.single_lane(0) == []
    # a join directly followed by the next fork, and a join after the last op
    lanes = [0, 1, 0, 1]
    sync = {0: [("fork", [1])], 2: [("join", [1]), ("fork", [1])], 4: [("join", [1])]}
    _check_plan(graphs.plan_segments(4, lanes, sync), 4, lanes)


@pytest.mark.parametrize("bad", ["unjoined", "unforked_op", "fork_main", "double_fork", "join_closed", "length"])
def test_plan_refuses_inconsistent_marks(bad):
    lanes, sync, n = [0, 1, 0], {1: [("fork", [1])], 2: [("join", [1])]}, 3
    if bad == "unjoined":
        sync = {1: [("fork", [1])]}
    elif bad == "unforked_op":
        sync = {}
    elif bad == "fork_main":
        sync = {1: [("fork", [0, 1])], 2: [("join", [1])]}
    elif bad == "double_fork":
        sync = {0: [("fork", [1])], 1: [("fork", [1])], 2: [("join", [1])]}
    elif bad == "join_closed":
        sync = {1: [("fork", [1])], 2: [("join", [1, 2])]}
    elif bad == "length":
        n = 4
    with pytest.raises(ValueError):
        graphs.plan_segments(n, lanes, sync)


@pytest.mark.parametrize("squeezed", [False, True])
def test_plan_of_the_post_filter_programs(squeezed):
    """Inference and both training programs of GaGNet: q stages x (glance | gaze_r | gaze_i) chains."""
    cfg = GagConfig(p=1, q=2, dilas=(1, 2), is_squeezed=squeezed)
    P = paramgen.make_params(gag_param_specs(cfg), 7)
    prog = prg.lower(cfg, P, 1, 10, 161)
    plan = graphs.plan_segments(len(prog.ops), prog.lanes, prog.sync)
    _check_plan(plan, len(prog.ops), prog.lanes)
    n_side = 1 if squeezed else 2
    assert sum(e[0] == graphs.FORK for e in plan) == cfg.q
    assert sum(e[0] == graphs.RUN and e[1] != 0 for e in plan) == cfg.q * n_side
    # back to back (Pipeline): no marks at all -> one run
    one = prg.lower(cfg, P, 1, 10, 161, parallel_chains=False)
    assert not one.sync and set(one.lanes) == {0}
    tp = train_gag.lower_train(cfg, 1, 10, 161, "f32")
    for which, ops in (("fwd", tp.fwd), ("bwd", tp.bwd)):
        plan = graphs.plan_segments(len(ops), tp.lanes[which], tp.sync[which])
        _check_plan(plan, len(ops), tp.lanes[which])
        assert any(e[0] == graphs.RUN and e[1] != 0 for e in plan)
        # every lane a mark names gets a stream, also one that carries no op (the squeezed gaze block forks a lane it never uses)
        named = {l for e in plan if e[0] != graphs.RUN for l in e[1]}
        assert named <= _lanes_with_streams(plan)


def _lanes_with_streams(plan):
    """the lanes graphs.LaneGraphs creates side streams for (its constructor's rule, without a GPU)"""
    out = set()
    for e in plan:
        out |= {l for l in ([e[1]] if e[0] == graphs.RUN else e[1]) if l != 0}
    return out


def test_branch_switch_no_longer_depends_on_a_process_group(monkeypatch):
    """Rounds 1-3 fenced the runtime crash by switching branches off while torch.distributed was initialised -- a decision
    taken at capture time that a later init_process_group could not undo.  Branches now replay as separate single-stream
    graphs, so the switch is the environment variable alone."""
    from eabnet_amd import model
    import torch.distributed as td
    monkeypatch.delenv("EAB_GRAPH_BRANCHES", raising=False)
    monkeypatch.setattr(td, "is_initialized", lambda: True)
    assert model.graph_branches_allowed()
    monkeypatch.setenv("EAB_GRAPH_BRANCHES", "0")
    assert not model.graph_branches_allowed()


def test_nola_check_is_keyed_by_window_content():
    """istft's host-side NOLA check caches its verdict; two windows that happen to live at the same address (the allocator
    hands a freed window's storage to the next one) must not share it."""
    import torch
    import torch.nn.functional as F
    from eabnet_amd import model
    bad = F.pad(torch.hann_window(100), (110, 110))
    with pytest.raises(RuntimeError, match="NOLA"):
        model._check_nola(bad, 320, 160, 12)
    bad.copy_(torch.hann_window(320))               # same storage, same object: now a valid window
    model._check_nola(bad, 320, 160, 12)
    model._check_nola(torch.ones(256), 256, 256, 11)
    with pytest.raises(RuntimeError, match="NOLA"):
        model._check_nola(torch.hann_window(256), 256, 256, 11)
    model._check_nola(torch.hann_window(256), 256, 256, 1)      # a single frame has no trimmed sample to divide
