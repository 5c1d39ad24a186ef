"""bench.py's multi-rank launch path with the REAL codec on a 1-GPU box (-m gpu).

`python bench.py --gpus 2` re-launches itself under torch.distributed.run (bench.launch_ranks); here both ranks are
pinned to cuda:0 (DCVC_BENCH_ONE_DEVICE=1) and talk over gloo (DCVC_BENCH_BACKEND=gloo): real sharding
(sharding.shard_range), the file-locked build path of two ranks in one tree, barriers, max-over-ranks, the closure flag
of every rank, one JSON line - everything of an N-GPU run but RCCL itself (which needs a second GPU:
tests/test_dmcht_gpu.py::test_recon_head_fan_out_over_rccl)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv):
    env = dict(os.environ, DCVC_BENCH_BACKEND="gloo", DCVC_BENCH_ONE_DEVICE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=850)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_two_ranks_code_their_own_pictures():
    d = _bench("--gpus", "2", "--steps", "4", "--warmup", "1", "--no-extras", "--no-cpu-baseline", "--no-roofline",
               "--min-seconds", "0")
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "weak"
    assert d["closure_ok"] is True                     # MIN over both ranks' own codec objects
    assert d["value"] > 0 and d["bytes_per_picture"] > 1000
    assert d["value"] == pytest.approx(2 * 4 / (d["ms_per_step"] * 4 / 1e3), rel=1e-6)


@pytest.mark.timeout(900)
def test_two_ranks_share_one_hierarchical_stream():
    d = _bench("--gpus", "2", "--workload", "hts", "--fanout", "--steps", "3", "--warmup", "1", "--no-extras", "--min-seconds", "0")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["sharding"] == "recon-head fan-out"
    assert d["value"] > 0 and d["config"]["pictures_per_step"] == 8


@pytest.mark.timeout(900)
def test_two_ranks_hand_a_running_gop_to_each_other():
    """north_star's temporal-context exchange, rehearsed with the REAL proxies: `--handoff 3` moves the temporal state of the
    LD encoder and decoder objects (sharding.send_state / recv_state -> dcvc_dmcld_export_state / _import_state) to the other
    rank every 3 coded pictures - five q indexes -, and the bytes of all 16 coded pictures
    equal those of the same stream coded on one rank without any hand-off. Over gloo (host-staged) here; on an N-GPU node the
    same command line runs over RCCL point-to-point."""
    d = _bench("--gpus", "2", "--workload", "ld", "--handoff", "3", "--steps", "14", "--warmup", "2", "--resolution", "640x360")
    h = d["handoff"]
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["sharding"].startswith("GOP hand-off")
    assert h["bit_exact_continuation"] is True and d["closure_ok"] is True
    assert h["count"] == 5 and h["bytes_per_handoff"] > 1_000_000 and h["us_per_handoff"] > 0 and h["backend"] == "gloo"
    assert d["value"] > 0


@pytest.mark.timeout(900)
def test_two_ranks_hand_a_hierarchical_stream_to_each_other():
    """the same with HT-S chunks of 8 pictures; chunk 3 resets the feature memory (the reference's cadence) right behind a
    hand-off"""
    d = _bench("--gpus", "2", "--workload", "hts", "--handoff", "2", "--steps", "6", "--warmup", "1", "--resolution", "640x360")
    assert d["handoff"]["bit_exact_continuation"] is True and d["handoff"]["count"] == 3
    assert d["config"]["pictures_per_step"] == 8


@pytest.mark.timeout(900)
def test_two_ranks_share_the_64_point_rate_sweep():
    """BASELINE configs[4]'s sweep (q_index = linspace(0, 63, 64)), rate points sharded over two ranks, at a small picture
    size here: closure at every one of the 64 rate points, the rate grows with q."""
    d = _bench("--gpus", "2", "--sweep64", "--workload", "ld", "--sweep-units", "2", "--resolution", "640x360")
    sw = d["sweep64"]
    assert d["n_gpus"] == 2 and d["closure_ok"] is True and sw["rate_points"] == 64 and all(sw["closure_ok_per_q"])
    assert sw["pictures_per_rate_point"] == 3 and sw["bpp_per_q"][63] > sw["bpp_per_q"][0] > 0
