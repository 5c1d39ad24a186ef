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
