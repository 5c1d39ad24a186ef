"""Child script of tests/test_bench_cpu.py::test_gpus_flag_spawns_the_ranks_itself: bench.main() with the GPU pieces
replaced by CPU stand-ins (the same ones as the in-process tests), so that the REAL launch path - `--gpus N` without
WORLD_SIZE re-launching this very script under torch.distributed.run, rendezvous, barriers, max-over-ranks, one JSON
line from rank 0 - runs on a CPU box with the gloo backend. Not collected by pytest."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

import bench  # noqa: E402
import test_bench_cpu as fakes  # noqa: E402


def install():
    import __graft_entry__
    torch.cuda.is_available = lambda: True
    torch.cuda.set_device = lambda d: None
    torch.cuda.Stream = fakes._FakeStream
    torch.cuda.Event = fakes._FakeEvent
    torch.cuda.set_stream = lambda s: None
    torch.cuda.current_device = lambda: 0
    torch.cuda.stream = lambda s: __import__("contextlib").nullcontext()
    torch.cuda.synchronize = lambda d=None: None
    torch.cuda.empty_cache = lambda: None
    __graft_entry__.build = lambda: None
    bench.build_model = lambda device: (fakes._FakeNet(), fakes._FakeNet())
    bench._to_gpu = lambda net, device: fakes._FakeNet()
    bench.make_pictures = lambda n, rank, device, height=0, width=0: [fakes._FakePicture(height, width)] * n
    bench.IntraWorkload = fakes._FakeWork
    bench.InterWorkload = fakes._FakeInter


if __name__ == "__main__":
    install()
    t0 = time.time()
    bench.main()
    print("rank %s done in %.1f s" % (os.environ.get("RANK", "-"), time.time() - t0), file=sys.stderr)
