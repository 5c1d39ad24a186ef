"""bench.py's control flow and JSON contract on a CPU box: the GPU pieces (codec, streams, events,
kernel stamps) are replaced by stand-ins, everything else - argument handling, the timed region,
the separate encode / decode timing pass, the other-workload runs, the fields of the one JSON line
the driver parses - is the real code."""
import json
import os
import subprocess
import sys
import time

import pytest
import torch

import bench


class _FakeStream:
    def __init__(self, *a, **k):
        pass

    def synchronize(self):
        pass


class _FakeEvent:
    def __init__(self, *a, **k):
        self.t = 0.0

    def record(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


class _FakeNet:
    """stands in for a DMCI object on the device (get_padding_size for everyone; compress / decompress for the rate sweep)"""

    def get_padding_size(self, h, w, p):
        return (-w) % p, (-h) % p

    def compress(self, x, qp, pad_b, pad_r):
        return {"bit_stream": b"i" * (100 + qp), "ec_parallel": 1, "x_hat": torch.full((1, 3, 2, 2), float(qp))}

    def decompress(self, bit_stream, sps, qp, ec):
        return {"x_hat": torch.full((1, 3, 2, 2), float(len(bit_stream) - 100))}


class _FakePicture:
    """stands in for a [1, 3, H, W] device tensor: the workloads only look at its shape"""

    def __init__(self, height, width):
        self.shape = (1, 3, height, width)


class _FakeWork:
    kind, default_graphs = "intra", True
    made = []

    height, width = 1080, 1920

    def __init__(self, *a, **k):
        self.frames = 1
        type(self).made.append(self)
        self.calls = []
        # objects made for the two-stage pipeline: IntraWorkload(net, pics, pb, pr, dec_net, prioritised) /
        # InterWorkload(kind, device, pics, net, pb, pr, prioritised)
        self.overlapped = bool(k.get("prioritised", False)) or (len(a) >= 6 and a[-1] is True)
        self.enc_prepared, self.dec_prepared = [], []
        for arg in a:                     # the picture list, if the caller passed one
            if isinstance(arg, list) and arg and isinstance(arg[0], _FakePicture):
                self.height, self.width = arg[0].shape[2], arg[0].shape[3]

    def prepare(self, i):
        pass

    def prepare_enc(self, i):
        self.enc_prepared.append(i)

    def prepare_dec(self, i):
        self.dec_prepared.append(i)

    def compress(self, i, qp):
        assert qp in bench.QPS
        self.calls.append(("c", i))
        time.sleep(0.004)
        return {"bit_stream": b"x" * 1000, "ec_parallel": 1}

    def decompress(self, i, qp, enc):
        self.calls.append(("d", i))
        time.sleep(0.008)

    def set_use_graphs(self, on):
        pass

    def closure(self, i, qp):
        self.closures = getattr(self, "closures", 0) + 1
        self.decompress(i, qp, self.compress(i, qp))
        return True


class _FakeCodecObject:
    """one side (encoder or decoder) of a stand-in inter codec: its temporal state is one int64, every coded unit folds
    its index into it - so bytes depend on the whole history, and a hand-off that lost or mixed up state shows"""

    def __init__(self):
        self.state = torch.zeros(1, dtype=torch.int64)

    def _ensure_proxy(self):
        return self

    def export_state(self):
        return self.state.clone().view(torch.uint8)

    def import_state(self, state, height, width):
        self.state = state.clone().view(torch.int64)

    def step(self, i):
        self.state = (self.state * 1000003 + i + 1) % 2147483647
        return int(self.state.item())

    def debug_read(self, name, dtype):
        import numpy as np
        return np.asarray([int(self.state.item())], dtype=np.int64)

    # the codec-object surface the rate sweep drives directly
    def add_ref_feature_from_frame(self, frame, apply_feature_adaptor=True):
        self.state = torch.full((1,), int(frame.flatten()[0].item()) + 11, dtype=torch.int64)

    def compress(self, x, qp, reset, pad_b, pad_r):
        return {"bit_stream": (b"%020d" % self.step(qp)) * (10 + qp), "ec_parallel": 1}

    def decompress(self, bit_stream, sps, qp, ec, reset):
        assert bit_stream[:20] == b"%020d" % self.step(qp)
        return {"x_hat": None}


class _FakeInter(_FakeWork):
    gop = 12

    def __init__(self, kind, *a, **k):
        super().__init__(kind, *a, **k)
        self.kind = kind
        self.frames = 1 if kind == "ld" else 8
        self.enc, self.dec = _FakeCodecObject(), _FakeCodecObject()
        self.inputs = [None]

    def prepare(self, i):
        if i % self.gop == 0:
            self.enc.state = torch.full((1,), 7, dtype=torch.int64)
            self.dec.state = torch.full((1,), 7, dtype=torch.int64)

    def compress(self, i, qp):
        r = super().compress(i, qp)
        r["bit_stream"] = (b"%020d" % self.enc.step(i)) * 50
        return r

    def decompress(self, i, qp, enc):
        super().decompress(i, qp, enc)
        assert enc["bit_stream"] == (b"%020d" % self.dec.step(i)) * 50, "decoder out of step with the encoder"


@pytest.fixture
def fake_gpu(monkeypatch):
    import __graft_entry__
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    monkeypatch.setattr(torch.cuda, "Stream", _FakeStream)
    monkeypatch.setattr(torch.cuda, "Event", _FakeEvent)
    monkeypatch.setattr(torch.cuda, "set_stream", lambda s: None)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    monkeypatch.setattr(torch.cuda, "stream", lambda s: __import__("contextlib").nullcontext())
    monkeypatch.setattr(torch.cuda, "synchronize", lambda d=None: None)
    monkeypatch.setattr(torch.cuda, "empty_cache", lambda: None)
    monkeypatch.setattr(__graft_entry__, "build", lambda: None)
    monkeypatch.setattr(bench, "build_model", lambda device: (_FakeNet(), _FakeNet()))
    monkeypatch.setattr(bench, "_to_gpu", lambda net, device: _FakeNet())
    monkeypatch.setattr(bench, "make_pictures", lambda n, rank, device, height=1080, width=1920: [_FakePicture(height, width)] * n)
    monkeypatch.setattr(bench, "IntraWorkload", _FakeWork)
    monkeypatch.setattr(bench, "InterWorkload", _FakeInter)
    monkeypatch.setattr(bench, "roofline", lambda work, n=5: {"bound": "mfma", "kernel": "dcb_nsplit8_kernel<384, 384, 64 px>", "achieved": 1.0,
                                                             "peak": 2500.0, "unit": "TFLOP/s", "frac": 0.0004, "traffic": None,
                                                             "all_contractions": {"achieved": 0.9, "frac": 0.00036}})
    monkeypatch.setattr(bench, "cpu_baseline", lambda net, device: {"value": 1e-3, "unit": "frames/s", "cores": 1,
                                                                    "kind": "port", "sample": "stand-in"})
    monkeypatch.setattr(bench, "sweep64_block", lambda env, kind, h, w, units: {"value": 1.0, "closure_ok": True, "rate_points": 64})
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    _FakeWork.made = []


def _run(monkeypatch, capsys, argv):
    monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
    bench.main()
    lines = [l for l in capsys.readouterr().out.splitlines() if l.strip()]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_one_json_line_with_the_contract_fields(fake_gpu, monkeypatch, capsys):
    d = _run(monkeypatch, capsys, ["--steps", "7", "--warmup", "2", "--min-seconds", "0.3"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "encode_fps", "decode_fps",
                "other_workloads", "sustained", "uhd", "pipelined", "box", "loop"):
        assert key in d, key
    # the reference's own metric rides inside `config` as well (a driver that keeps only the contract keys keeps it)
    assert d["config"]["encode_fps"] == d["encode_fps"] and d["config"]["decode_fps"] == d["decode_fps"]
    assert d["config"]["loop"] == "sequential" and "sequential" in d["metric"]
    # the two-stage pipeline is reported beside `value`, never as `value`
    assert "two-stage pipeline" in d["pipelined"]["loop"] and "one call after the other" in d["loop"]
    assert d["pipelined"]["closure_ok"] is True and d["pipelined"]["sustained"]["seconds"] >= 0.25
    # the K timed steps stay exactly K; the longer region behind them is reported beside, never instead
    assert d["sustained"]["seconds"] >= 0.25 and d["sustained"]["steps"] >= 7 and d["sustained"]["value"] > 0
    assert d["uhd"]["resolution"] == "3840x2160" and set(d["uhd"]) == {"resolution", "intra", "ld", "hts", "htl", "sweep64"}
    assert d["roofline"]["kernel"].startswith("dcb_nsplit8_kernel") and "all_contractions" in d["roofline"]
    assert all("roofline" in o for o in d["other_workloads"].values())
    assert d["n_gpus"] == 1 and d["steps"] == 7 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f16"
    assert d["config"]["pictures_per_step"] == 1 and "workload" in d["config"] and "model" not in d["config"]
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert d["value"] == pytest.approx(7 / (d["ms_per_step"] * 7 / 1e3), rel=1e-6)
    assert d["bytes_per_picture"] == 1000
    assert all("pipelined" in o for o in d["other_workloads"].values())
    # compress takes 4 ms and decompress 8 ms in the stand-in: the two rates are measured separately (wide bounds:
    # sleep() overshoots on a busy host)
    assert d["encode_fps"] > d["decode_fps"] > 0
    assert 1.3 < d["encode_fps"] / d["decode_fps"] < 2.8
    assert set(d["other_workloads"]) == {"ld", "hts", "htl"}
    # closure (decoder output == encoder output for every rate point) is checked behind every timed region and reported
    assert d["closure_ok"] is True and all(o["closure_ok"] is True for o in d["other_workloads"].values())
    assert all(d["uhd"][k]["closure_ok"] is True for k in ("intra", "ld", "hts", "htl"))
    for kind, o in d["other_workloads"].items():
        assert set(o) >= {"value", "encode_fps", "decode_fps", "ms_per_step"} and o["value"] > 0
    assert d["other_workloads"]["hts"]["value"] > d["other_workloads"]["ld"]["value"]     # 8 pictures per call


def test_timed_region_runs_exactly_k_steps(fake_gpu, monkeypatch, capsys):
    d = _run(monkeypatch, capsys, ["--steps", "5", "--warmup", "3", "--no-extras", "--no-roofline", "--no-cpu-baseline",
                                   "--min-seconds", "0"])
    w = _FakeWork.made[0]
    timed_and_warm = [c for c in w.calls if c[0] == "c" and c[1] < 8]
    assert len(timed_and_warm) == 8                       # 3 warm-up + 5 timed, then the per-call timing pass
    assert "roofline" not in d and "cpu_baseline" not in d and "other_workloads" not in d and "sustained" not in d


def test_resolution_flag(fake_gpu, monkeypatch, capsys):
    d = _run(monkeypatch, capsys, ["--steps", "3", "--warmup", "1", "--resolution", "3840x2160", "--no-extras",
                                   "--no-cpu-baseline", "--min-seconds", "0"])
    assert d["config"]["resolution"] == "3840x2160" and "3840x2160" in d["metric"] and "uhd" not in d
    assert _FakeWork.made[0].height == 2160 and _FakeWork.made[0].width == 3840
    assert d["bpp"] == pytest.approx(8.0 * 1000 / (3840 * 2160))
    with pytest.raises(SystemExit):
        monkeypatch.setattr(sys, "argv", ["bench.py", "--resolution", "1920"])
        bench.main()


def test_reset_cadence_of_the_inter_workloads():
    """test_video.py:232-235 with reset_interval 32: LD resets the feature memory on every 32nd picture of a GOP,
    the 8-picture models on every 4th chunk (frame_idx + 8) % 32 == 1."""
    w = bench.InterWorkload.__new__(bench.InterWorkload)
    w.frames, w.gop = 1, 96
    assert [i for i in range(96) if w._reset(i)] == [31, 63, 95]
    w.frames, w.gop = 8, 12
    assert [i for i in range(24) if w._reset(i)] == [3, 7, 11, 15, 19, 23]


def test_inter_workload_line(fake_gpu, monkeypatch, capsys):
    d = _run(monkeypatch, capsys, ["--steps", "6", "--warmup", "1", "--workload", "hts", "--no-extras", "--min-seconds", "0"])
    assert d["config"]["pictures_per_step"] == 8 and "HT-S" in d["metric"]
    assert d["value"] == pytest.approx(8 * 6 / (d["ms_per_step"] * 6 / 1e3), rel=1e-6)
    assert "cpu_baseline" not in d                        # the CPU baseline belongs to the headline workload
    # `value`: compress (4 ms in the stand-in), then decompress (8 ms), one after the other: >= 12 ms per step
    assert "one call after the other" in d["loop"] and d["ms_per_step"] > 11.0
    plain, piped = _FakeWork.made[0], _FakeWork.made[1]
    assert not plain.overlapped and piped.overlapped
    assert [c for c in plain.calls if c[1] < 7] == [(k, i) for i in range(7) for k in ("c", "d")]
    # `pipelined`: separate objects made for it; the timed steps run as a two-stage pipeline (a step costs ~ 8 ms instead of
    # 12), every step coded and decoded exactly once, in order, on both sides
    p = d["pipelined"]
    assert "two-stage pipeline" in p["loop"] and p["ms_per_step"] < 11.0 and p["closure_ok"] is True
    # step 0 = the plain warm-up, steps 1 - 2 = the pipeline's own untimed warm-up, steps 3 - 8 = the six timed steps
    assert piped.enc_prepared == piped.dec_prepared == list(range(1, 9))
    timed = [c for c in piped.calls if 3 <= c[1] <= 8]
    assert sorted(timed) == sorted([("c", i) for i in range(3, 9)] + [("d", i) for i in range(3, 9)])
    assert [i for k, i in timed if k == "d"] == list(range(3, 9))


def test_no_pipeline_switch(fake_gpu, monkeypatch, capsys):
    d = _run(monkeypatch, capsys, ["--steps", "3", "--warmup", "1", "--workload", "ld", "--no-extras", "--min-seconds", "0",
                                   "--no-pipeline"])
    assert "pipelined" not in d and "one call after the other" in d["loop"] and d["ms_per_step"] > 11.0
    assert len(_FakeWork.made) == 1


def test_a_failing_encoder_thread_surfaces(fake_gpu, monkeypatch):
    w = _FakeInter("ld", prioritised=True)

    def boom(i, qp):
        if i == 2:
            raise RuntimeError("encoder failed")
        return {"bit_stream": b"x", "ec_parallel": 1}

    w.compress = boom
    w.decompress = lambda i, qp, enc: None
    with pytest.raises(RuntimeError, match="encoder failed"):
        bench.run_steps_overlapped(w, 0, 5)


class _FakeDist:
    """stands in for torch.distributed inside bench.main(): one process plays one rank of a 2-GPU launch"""
    ReduceOp = type("ReduceOp", (), {"MAX": "max", "MIN": "min"})

    def __init__(self, rank, world):
        self.rank, self.world, self.barriers, self.reduced = rank, world, 0, 0

    def init_process_group(self, backend=None, device_id=None, **k):
        assert backend == "nccl"

    def barrier(self):
        self.barriers += 1

    def all_reduce(self, t, op=None):
        assert op in ("max", "min")
        if op == "max":
            self.reduced += 1
        else:
            self.closure_votes = getattr(self, "closure_votes", 0) + 1      # every rank's closure flag, MIN over ranks

    def get_rank(self):
        return self.rank

    def get_world_size(self):
        return self.world

    def destroy_process_group(self):
        pass


@pytest.mark.parametrize("rank", [0, 1])
def test_one_rank_of_a_two_gpu_launch(fake_gpu, monkeypatch, capsys, rank):
    """The driver launches `bench.py --gpus 2` under torch.distributed.run: every rank codes its own share of the
    2 K steps (sharding.shard_range), rank 0 alone prints the line, with the aggregate over both ranks."""
    fake = _FakeDist(rank, 2)
    monkeypatch.setitem(sys.modules, "torch.distributed", fake)
    monkeypatch.setattr(torch, "distributed", fake, raising=False)
    monkeypatch.setattr(torch, "tensor", lambda data, dtype=None, device=None: torch.as_tensor(data, dtype=dtype))
    monkeypatch.setenv("RANK", str(rank))
    monkeypatch.setenv("LOCAL_RANK", str(rank))
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "4", "--warmup", "1", "--no-roofline", "--min-seconds", "0.1"])
    bench.main()
    lines = [l for l in capsys.readouterr().out.splitlines() if l.strip()]
    w = _FakeWork.made[0]
    timed = [c[1] for c in w.calls if c[0] == "c"][1:5]
    assert timed == [1 + 4 * rank + i for i in range(4)]          # warm-up step 0, then this rank's share of the 8 steps
    assert fake.barriers >= 3 and fake.reduced == 2          # the K timed steps and the sustained region
    assert fake.closure_votes == 1 and w.closures == len(bench.QPS)          # every rank checks its own codec objects
    if rank == 1:
        assert lines == []
        return
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 4
    assert d["value"] == pytest.approx(2 * 4 / (d["ms_per_step"] * 4 / 1e3), rel=1e-6)
    assert "other_workloads" not in d and "cpu_baseline" not in d          # single-GPU extras only at N = 1


def test_fan_out_line(fake_gpu, monkeypatch, capsys):
    """`--fanout` (one hierarchical stream over all ranks): strong scaling, every rank takes part in the per-call
    timing loop, the value counts the stream's pictures once."""
    fake = _FakeDist(0, 2)
    monkeypatch.setitem(sys.modules, "torch.distributed", fake)
    monkeypatch.setattr(torch, "distributed", fake, raising=False)
    monkeypatch.setattr(torch, "tensor", lambda data, dtype=None, device=None: torch.as_tensor(data, dtype=dtype))
    monkeypatch.setattr(bench, "FanoutWorkload", lambda kind, device, pics, net, pb, pr, dist: _FakeInter(kind))
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("LOCAL_RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--workload", "hts", "--fanout", "--steps", "3", "--warmup", "1", "--min-seconds", "0"])
    bench.main()
    d = json.loads([l for l in capsys.readouterr().out.splitlines() if l.strip()][0])
    assert d["scaling"] == "strong" and d["n_gpus"] == 2 and d["config"]["sharding"] == "recon-head fan-out"
    assert d["closure_ok"] is None                          # the shared stream is covered by the fan-out tests, not here
    assert d["value"] == pytest.approx(8 * 3 / (d["ms_per_step"] * 3 / 1e3), rel=1e-6)
    assert "roofline" not in d
    with pytest.raises(SystemExit):
        monkeypatch.setattr(sys, "argv", ["bench.py", "--workload", "ld", "--fanout"])
        bench.main()


def test_sweep64_line(monkeypatch, capsys, fake_gpu):
    """BASELINE configs[4]: all 64 rate points, closure per rate point, 3840x2160 unless told otherwise"""
    monkeypatch.undo()                       # the fixture replaced sweep64_block by a constant: this test wants the real one
    import __graft_entry__
    for name, val in (("is_available", lambda: True), ("set_device", lambda d: None), ("Stream", _FakeStream), ("Event", _FakeEvent),
                      ("set_stream", lambda s: None), ("current_device", lambda: 0), ("synchronize", lambda d=None: None),
                      ("empty_cache", lambda: None)):
        monkeypatch.setattr(torch.cuda, name, val)
    monkeypatch.setattr(__graft_entry__, "build", lambda: None)
    monkeypatch.setattr(bench, "build_model", lambda device: (_FakeNet(), _FakeNet()))
    monkeypatch.setattr(bench, "_to_gpu", lambda net, device: _FakeNet())
    monkeypatch.setattr(bench, "make_pictures", lambda n, rank, device, height=1080, width=1920: [_FakePicture(height, width)] * n)
    monkeypatch.setattr(bench, "InterWorkload", _FakeInter)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    d = _run(monkeypatch, capsys, ["--sweep64", "--workload", "ld", "--sweep-units", "3"])
    sw = d["sweep64"]
    assert d["config"]["resolution"] == "3840x2160" and "64-point rate sweep" in d["metric"] and d["scaling"] == "strong"
    assert sw["rate_points"] == 64 and sw["pictures_per_rate_point"] == 4 and len(sw["closure_ok_per_q"]) == 64
    assert d["closure_ok"] is True and all(sw["closure_ok_per_q"])
    assert d["steps"] == 64 and d["value"] == pytest.approx(64 * 4 / sw["seconds"])
    assert sw["bpp_per_q"] == sorted(sw["bpp_per_q"]) and sw["bpp_per_q"][0] < sw["bpp_per_q"][-1]      # the stand-in's rate grows with q


@pytest.mark.timeout(300)
def test_gop_hand_off_mode_over_two_ranks():
    """`bench.py --gpus 2 --workload ld --handoff 3`: ONE stream, its temporal state (encoder and decoder objects) moves to
    the other rank every 3 coded units by sharding.send_state / recv_state (gloo here, RCCL on the GPU box); rank 0 collects
    the bytes of every unit and checks them against the same stream coded on one rank. The stand-in codec's bytes depend on
    the whole history of its state, so a lost or stale hand-off fails `bit_exact_continuation`."""
    env = dict(os.environ, DCVC_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_launch_child.py")
    res = subprocess.run([sys.executable, child, "--gpus", "2", "--workload", "ld", "--handoff", "3", "--steps", "14", "--warmup", "2"],
                         env=env, capture_output=True, text=True, timeout=280)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    h = d["handoff"]
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["sharding"].startswith("GOP hand-off")
    assert h["bit_exact_continuation"] is True and d["closure_ok"] is True
    # units 0 .. 15 in groups of 3: boundaries behind units 2, 5, 8, 11, 14; the one behind unit 2 is the first timed one
    assert h["count"] == 5 and h["every_units"] == 3 and h["bytes_per_handoff"] == 16 and h["us_per_handoff"] > 0
    assert d["value"] == pytest.approx(14 / (d["ms_per_step"] * 14 / 1e3), rel=1e-6)


@pytest.mark.timeout(300)
def test_rate_sweep_sharded_over_two_ranks():
    """`bench.py --gpus 2 --sweep64`: the 64 rate points are split over the ranks (sharding.shard_range), every rank reports
    its points with its closure flags (all_gather_object over gloo here), rank 0 prints the one line with all 64 in order."""
    env = dict(os.environ, DCVC_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_launch_child.py")
    res = subprocess.run([sys.executable, child, "--gpus", "2", "--sweep64", "--workload", "ld", "--sweep-units", "2"],
                         env=env, capture_output=True, text=True, timeout=280)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    sw = d["sweep64"]
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 64 and d["config"]["resolution"] == "3840x2160"
    assert sw["rate_points"] == 64 and len(sw["closure_ok_per_q"]) == 64 and all(sw["closure_ok_per_q"]) and d["closure_ok"] is True
    assert sw["pictures_per_rate_point"] == 3 and sw["bpp_per_q"] == sorted(sw["bpp_per_q"])
    assert res.stderr.count("done in") == 2


@pytest.mark.timeout(300)
def test_gpus_flag_spawns_the_ranks_itself():
    """`python bench.py --gpus 2` with NO launcher around it (how the driver calls it): the process re-launches
    itself under torch.distributed.run, both ranks rendezvous on 127.0.0.1, run the real main() - barriers, shard of
    the 2 K steps, max over ranks - and rank 0 alone prints the one line with n_gpus 2. gloo + stand-in codec here;
    on the GPU box the same path runs nccl (= RCCL)."""
    env = dict(os.environ, DCVC_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_launch_child.py")
    res = subprocess.run([sys.executable, child, "--gpus", "2", "--steps", "4", "--warmup", "1", "--no-roofline", "--min-seconds", "0.2"],
                         env=env, capture_output=True, text=True, timeout=280)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "weak"
    assert d["value"] == pytest.approx(2 * 4 / (d["ms_per_step"] * 4 / 1e3), rel=1e-6)
    assert d["sustained"]["seconds"] >= 0.15               # sized for 0.2 s from the K timed steps
    assert res.stderr.count("done in") == 2                # two rank processes ran main() to the end
