"""bench.py's control flow and JSON contract on a CPU box: the GPU pieces (codec, streams, kernel
stamps) are replaced by stand-ins, everything else - argument handling, lane dealing, the timed
region, the fields of the one JSON line the driver parses - is the real code."""
import contextlib
import json
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


class _FakeNet:
    def get_padding_size(self, h, w, p):
        return (-w) % p, (-h) % p


class _FakeWork:
    frames, kind, graphs = 1, "intra", True
    made = 0

    def __init__(self, *a, **k):
        type(self).made += 1
        self.calls = []

    def step(self, i, qp):
        assert qp in bench.QPS
        self.calls.append((i, qp))
        time.sleep(0.001)
        return 1000

    def set_use_graphs(self, on):
        pass


@pytest.fixture
def fake_gpu(monkeypatch):
    import __graft_entry__
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    monkeypatch.setattr(torch.cuda, "Stream", _FakeStream)
    monkeypatch.setattr(torch.cuda, "set_stream", lambda s: None)
    monkeypatch.setattr(torch.cuda, "stream", lambda s: contextlib.nullcontext())
    monkeypatch.setattr(torch.cuda, "current_stream", lambda d=None: _FakeStream())
    monkeypatch.setattr(torch.cuda, "synchronize", lambda d=None: None)
    monkeypatch.setattr(__graft_entry__, "build", lambda: None)
    monkeypatch.setattr(bench, "build_model", lambda device: (_FakeNet(), _FakeNet()))
    monkeypatch.setattr(bench, "_to_gpu", lambda net, device: net)
    monkeypatch.setattr(bench, "make_pictures", lambda n, rank, device: [None] * n)
    monkeypatch.setattr(bench, "IntraWorkload", _FakeWork)
    monkeypatch.setattr(bench, "roofline", lambda work: {"bound": "mfma", "achieved": 1.0, "peak": 2500.0,
                                                        "unit": "TFLOP/s", "frac": 0.0004, "traffic": None})
    monkeypatch.setattr(bench, "cpu_baseline", lambda net: {"value": 1e-3, "unit": "frames/s", "cores": 1,
                                                            "kind": "port", "sample": "stand-in"})
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "DCVC_BENCH_LANES", "DCVC_BENCH_EAGER", "DCVC_BENCH_POOL1"):
        monkeypatch.delenv(k, raising=False)
    _FakeWork.made = 0


@pytest.mark.parametrize("lanes", [None, 1, 3])
def test_one_json_line_with_the_contract_fields(fake_gpu, monkeypatch, capsys, lanes):
    argv = ["bench.py", "--steps", "7", "--warmup", "2"] + ([] if lanes is None else ["--lanes", str(lanes)])
    monkeypatch.setattr(sys, "argv", argv)
    bench.main()
    lines = [l for l in capsys.readouterr().out.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    n = lanes or 1
    assert d["n_gpus"] == 1 and d["steps"] == 7 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f16"
    assert d["config"]["lanes"] == n and d["config"]["pictures_per_step"] == n and "workload" in d["config"]
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert d["value"] == pytest.approx(n * 7 / (d["ms_per_step"] * 7 / 1e3), rel=1e-6)
    assert d["bytes_per_picture"] == 1000
    assert _FakeWork.made == n
    assert ("one_lane" in d) == (n > 1)
