"""Several independent coding lanes on ONE GPU.

The reference codes one picture at a time on one stream (test_video.py:186-366); a picture's
decode has four GPU -> CPU -> GPU round trips (dmci_proxy.cpp:857-871: each masked group needs the
entropy decoder's symbols before the next prior can be computed) and every kernel boundary drains
the chip. Pictures of an all-intra sequence, GOPs of an inter sequence and the qps of a rate sweep
are independent units (the same units `dcvc_amd.sharding` deals over GPUs), so one process per GPU
can keep `n` of them in flight:

  lane = its own codec objects (own resident buffers, compute stream, entropy-coding worker - the
         objects `factory(lane_index)` returns), one torch side stream for the tensors that cross
         the plugin boundary, one host thread issuing the reference-surface calls.

Measured on MI355X at 1080p (profiles/README.md, v5), two lanes against one lane: intra +5..15 %,
LD +3..5 %, HT-S / HT-L -15 %; more than two lanes lose everywhere - the contraction kernel owns a
whole CU, lanes only fill boundary and round-trip gaps and share L2 / MALL. Use it to serve several
streams from one GPU, not as a throughput trick; bench.py runs one lane.

Nothing is shared between lanes and every unit runs through exactly the single-lane path, so the
bytes and reconstructions of a unit do not depend on the number of lanes
(tests/test_lanes_cpu.py, tests/test_dmci_gpu.py::test_lanes_code_the_same_bytes).
"""
import contextlib
import threading


class LanePool:
    def __init__(self, n, factory, device=None):
        """factory(lane_index) -> the lane's private state (e.g. an (encoder, decoder) pair); called
        on the caller's thread, lane after lane. device: a torch cuda device, or None on a CPU box
        (the lanes are then plain threads - used by the CPU tests of the dealing logic)."""
        if n < 1:
            raise ValueError("at least one lane")
        self.device = device
        self.streams = [None] * n
        if device is not None:
            import torch
            self.streams = [torch.cuda.Stream(device=device) for _ in range(n)]
        self.states = []
        self._caller_sync()
        for k in range(n):
            with self._on_lane(k):
                self.states.append(factory(k))

    def __len__(self):
        return len(self.states)

    @contextlib.contextmanager
    def _on_lane(self, k):
        if self.device is None:
            yield
            return
        import torch
        torch.cuda.set_device(self.device)      # the current device is per host thread
        with torch.cuda.stream(self.streams[k]):
            yield
            self.streams[k].synchronize()

    def _caller_sync(self):
        if self.device is not None:
            import torch
            torch.cuda.current_stream(self.device).synchronize()    # what the caller queued for the lanes is complete

    def _run(self, bodies):
        errors = []

        def guarded(k, body):
            try:
                with self._on_lane(k):
                    body()
            except BaseException as e:      # noqa: BLE001 - re-raised on the caller's thread
                errors.append(e)

        self._caller_sync()
        threads = [threading.Thread(target=guarded, args=(k, b), name="dcvc-lane-%d" % k) for k, b in enumerate(bodies)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]

    def run_each(self, fn):
        """fn(lane_index, state) once on every lane, concurrently -> [result per lane]."""
        out = [None] * len(self)

        def body(k):
            def b():
                out[k] = fn(k, self.states[k])
            return b

        self._run([body(k) for k in range(len(self))])
        return out

    def warm(self, fn):
        """fn(lane_index, state) on every lane, ONE LANE AFTER THE OTHER on the caller's thread
        (first calls allocate the resident buffers and capture the stage graphs)."""
        out = []
        self._caller_sync()
        for k in range(len(self)):
            with self._on_lane(k):
                out.append(fn(k, self.states[k]))
        return out

    def map(self, fn, units):
        """fn(state, unit) for every unit -> results in unit order. Units are pulled from a shared
        queue, so a lane that finishes early takes the next one (units of unequal cost, e.g. GOPs of
        different length); which lane coded a unit does not influence its result."""
        units = list(units)
        out = [None] * len(units)
        lock = threading.Lock()
        cursor = [0]
        failed = threading.Event()

        def body(k):
            def b():
                while not failed.is_set():
                    with lock:
                        i = cursor[0]
                        cursor[0] += 1
                    if i >= len(units):
                        return
                    try:
                        out[i] = fn(self.states[k], units[i])
                    except BaseException:
                        failed.set()            # the other lanes stop after their current unit
                        raise
            return b

        self._run([body(k) for k in range(len(self))])
        return out
