"""dcvc_amd.lanes.LanePool on a CPU box (device=None: the lanes are plain threads): dealing logic,
result order, lane-private state, error propagation. The GPU behaviour (same bytes whatever the
number of lanes) is in tests/test_dmci_gpu.py::test_lanes_code_the_same_bytes."""
import threading
import time

import pytest

from dcvc_amd.lanes import LanePool


class _State:
    def __init__(self, k):
        self.k, self.seen, self.threads = k, [], set()

    def code(self, unit):
        self.threads.add(threading.get_ident())
        self.seen.append(unit)
        time.sleep(0.001 * (unit % 3))
        return unit * unit


def test_map_keeps_unit_order_and_lane_private_state():
    pool = LanePool(3, _State)
    assert len(pool) == 3 and [s.k for s in pool.states] == [0, 1, 2]
    units = list(range(40))
    assert pool.map(lambda st, u: st.code(u), units) == [u * u for u in units]
    seen = sorted(u for s in pool.states for u in s.seen)
    assert seen == units                                   # every unit exactly once
    assert all(len(s.threads) == 1 for s in pool.states)   # a lane's state is touched by one thread only
    assert len({t for s in pool.states for t in s.threads}) == 3
    assert pool.map(lambda st, u: u, []) == []


def test_run_each_and_warm():
    pool = LanePool(4, _State)
    main = threading.get_ident()
    assert pool.warm(lambda k, st: (k, threading.get_ident())) == [(k, main) for k in range(4)]
    together = threading.Barrier(4, timeout=10)             # passes only if the four lanes run at the same time

    def fn(k, st):
        together.wait()
        return k, st.k, threading.get_ident()

    got = pool.run_each(fn)
    assert [g[:2] for g in got] == [(k, k) for k in range(4)]
    assert len({g[2] for g in got}) == 4 and main not in {g[2] for g in got}


def test_lanes_overlap_in_time():
    pool = LanePool(4, _State)
    t0 = time.perf_counter()
    pool.map(lambda st, u: time.sleep(0.05), range(8))
    assert time.perf_counter() - t0 < 0.3                  # 8 x 50 ms on 4 lanes, not 400 ms


def test_errors_reach_the_caller_and_stop_the_queue():
    pool = LanePool(2, _State)

    def fn(st, u):
        if u == 3:
            raise RuntimeError("unit 3 is broken")
        time.sleep(0.002)
        return st.code(u)

    with pytest.raises(RuntimeError, match="unit 3"):
        pool.map(fn, range(200))
    assert sum(len(s.seen) for s in pool.states) < 199
    with pytest.raises(ValueError):
        LanePool(0, _State)
