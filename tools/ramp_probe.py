#!/usr/bin/env python
"""Throughput of one lane over time (windows of a few steps) - is the bench measuring a clock ramp?
   python tools/ramp_probe.py ld|hts|intra [null|side] [windows] [steps_per_window]"""
import contextlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "ld"
user = sys.argv[2] if len(sys.argv) > 2 else "null"
windows = int(sys.argv[3]) if len(sys.argv) > 3 else 30
per = int(sys.argv[4]) if len(sys.argv) > 4 else 20
device = torch.device("cuda", 0)
torch.cuda.set_device(device)
cpu_net, gpu_net = bench.build_model(device)
pics = bench.make_pictures(5, 0, device)
pad_r, pad_b = gpu_net.get_padding_size(bench.HEIGHT, bench.WIDTH, 16)
work = bench.IntraWorkload(gpu_net, pics, pad_b, pad_r) if kind == "intra" else \
    bench.InterWorkload(kind, device, pics, gpu_net, pad_b, pad_r)
stream = torch.cuda.Stream(device=device)
ctx = (lambda: torch.cuda.stream(stream)) if user == "side" else contextlib.nullcontext
torch.cuda.synchronize()


def ordering_check(n=8):
    """a null-stream consumer right behind decompress (no host sync) must see the finished picture"""
    bad = 0
    for i in range(n):
        x = work.inputs[i % len(work.inputs)]
        enc = work.enc.compress(x, 32, 0, work.pad_b, work.pad_r)
        d = work.dec.decompress(enc["bit_stream"], {"height": bench.HEIGHT, "width": bench.WIDTH}, 32,
                                enc["ec_parallel"], 0)["x_hat"]
        c = d.clone()
        torch.cuda.synchronize()
        bad += int(not torch.equal(c, d))
    return bad


modes = os.environ.get("PROBE_NULL_MODES", "").split(",") if os.environ.get("PROBE_NULL_MODES") else [None]
step = 0
for mode in modes:
  if mode is not None:
      os.environ["DCVC_NULL_STREAM_MODE"] = mode
  out = []
  with ctx():
    for w in range(windows):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(per):
            work.step(step, bench.QPS[step % 5])
            step += 1
        torch.cuda.synchronize()
        out.append(per * work.frames / (time.perf_counter() - t0))
  if os.environ.get("PROBE_CHECK"):
      print("mode", mode, "consumer saw an unfinished picture in", ordering_check(), "of 8 calls", flush=True)
  print(kind, user, "null-stream mode", mode, "pictures/s per window of %d steps:" % per, " ".join("%.0f" % v for v in out), flush=True)
