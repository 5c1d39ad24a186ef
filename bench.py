#!/usr/bin/env python
"""bench.py - DCVC-UF-Intra 1080p YUV420 encode+decode throughput on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): DCVC-UF-Intra (DMCI, 42.2 M parameters, seeded synthetic
weights of the reference architecture), 1920x1080 YUV420 synthetic pictures, q_index cycling over
{0, 16, 32, 48, 63}, skip_thres 0.15 (the reference's runtime setting, test_compress_time.py:41).
One step = compress one picture to a real rANS bit stream + decompress it again
(DMCI.compress + DMCI.decompress of the reference surface, host entropy coding included),
pictures already resident in HBM as fp16 NHWC tensors. Every rank codes its own pictures
(all-intra pictures are independent: no data-path collective), value = pictures/s of the job.

One JSON line on rank 0 with the fields of the driver contract plus
  roofline     - conv_gemm (the matrix-core contraction kernel, >99 % of the FLOPs): algorithmic
                 FLOPs (2*M*N*K per launch) / HIP-event time of those launches, measured live on
                 the codec's stream in an extra eager pass of the same workload
  cpu_baseline - the CPU oracle (oracle/codec.py, OpenMP C contractions) timed on this host on a
                 bounded sample (one 160x160 crop, encode + decode), scaled to 1080p pictures
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HEIGHT, WIDTH = 1080, 1920
QPS = (0, 16, 32, 48, 63)
SKIP_THRES = 0.15
MFMA_PEAK_TFLOPS = 2500.0        # dense fp16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=40)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--frames", type=int, default=5, help="distinct synthetic pictures per rank")
    return p.parse_args()


def build_model(device):
    from dcvc_amd import arch, models, synthetic
    net = models.DMCI()
    net.load_state_dict(synthetic.synthetic_state_dict(arch.dmci_spec(), 0))
    net.update(SKIP_THRES)
    cpu_net = net
    import copy
    gpu_net = copy.deepcopy(net).half().to(device)
    gpu_net.proxy = None
    return cpu_net, gpu_net


def make_pictures(n, rank, device):
    from dcvc_amd import synthetic
    pics = []
    for i in range(n):
        y, uv = synthetic.synthetic_frame_yuv420(HEIGHT, WIDTH, index=i, seed=rank)
        x = synthetic.yuv420_to_x(y, uv).half().to(device)
        pics.append(x.contiguous(memory_format=torch.channels_last))
    return pics


def step(net, x, qp, pad_b, pad_r):
    enc = net.compress(x, qp, pad_b, pad_r)
    dec = net.decompress(enc["bit_stream"], {"height": HEIGHT, "width": WIDTH}, qp, enc["ec_parallel"])
    return enc, dec


def cpu_baseline(cpu_net):
    """The oracle (kind "port") on a bounded sample of the same workload."""
    from oracle import codec
    from dcvc_amd import synthetic
    h = w = 160
    y, uv = synthetic.synthetic_frame_yuv420(h, w, 0, 0)
    x = synthetic.yuv420_to_x(y, uv)[0].permute(1, 2, 0).contiguous().numpy().astype(np.float16)
    o = codec.DMCIOracle(cpu_net.state_dict(), SKIP_THRES, cpu_net.get_cdf_info())
    t0 = time.time()
    r = o.compress(x, 32)
    o.decompress(r["bit_stream"], 32, h, w, r["ec_parallel"])
    dt = time.time() - t0
    frames_1080p = (h * w) / float(HEIGHT * WIDTH)
    return {
        "value": frames_1080p / dt, "unit": "frames/s", "cores": os.cpu_count(), "kind": "port",
        "sample": "oracle encode+decode of one %dx%d crop (qp 32) in %.1f s, scaled by pixel count to "
                  "1080p pictures" % (h, w, dt),
    }


def roofline(gpu_net, pics, pad_b, pad_r):
    """conv_gemm launches of encode+decode bracketed by HIP events (eager pass)."""
    from dcvc_amd import _lib
    en = _lib.fn("dcvc_gemm_profile_enable", ctypes.c_int, [ctypes.c_int])
    rs = _lib.fn("dcvc_gemm_profile_reset", ctypes.c_int, [])
    co = _lib.fn("dcvc_gemm_profile_collect", ctypes.c_int,
                 [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                  ctypes.POINTER(ctypes.c_longlong)])
    gpu_net.proxy.set_use_graphs(False)
    step(gpu_net, pics[0], QPS[2], pad_b, pad_r)      # eager warm-up
    torch.cuda.synchronize()
    _lib.check(en(1))
    _lib.check(rs())
    n = 0
    for i, qp in enumerate(QPS):
        step(gpu_net, pics[i % len(pics)], qp, pad_b, pad_r)
        n += 1
    torch.cuda.synchronize()
    ms, fl, ln = ctypes.c_double(), ctypes.c_double(), ctypes.c_longlong()
    _lib.check(co(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(ln)))
    if os.environ.get("DCVC_BENCH_SHAPES"):
        rec = np.dtype([("M", np.int32), ("N", np.int32), ("K", np.int32), ("variant", np.int32), ("ms", np.float32)])
        buf = np.zeros(int(ln.value), dtype=rec)
        _lib.fn("dcvc_gemm_profile_launches", ctypes.c_longlong, [ctypes.c_void_p, ctypes.c_longlong])(
            buf.ctypes.data, len(buf))
        agg = {}
        for r in buf:
            k = (int(r["M"]), int(r["N"]), int(r["K"]), int(r["variant"]))
            a = agg.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += float(r["ms"])
        with open(os.environ["DCVC_BENCH_SHAPES"], "w") as f:
            f.write("M,N,K,variant,calls_per_step,avg_us,tflops,ms_per_step\n")
            for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                f.write("%d,%d,%d,%d,%.1f,%.2f,%.1f,%.3f\n" % (
                    k + (a[0] / n, 1e3 * a[1] / a[0], 2.0 * k[0] * k[1] * k[2] / (a[1] / a[0] * 1e-3) / 1e12, a[1] / n)))
    _lib.check(en(0))
    gpu_net.proxy.set_use_graphs(True)
    achieved = fl.value / (ms.value * 1e-3) / 1e12
    return {
        "bound": "mfma", "kernel": "conv_gemm_kernel", "achieved": achieved, "peak": MFMA_PEAK_TFLOPS,
        "unit": "TFLOP/s", "frac": achieved / MFMA_PEAK_TFLOPS, "traffic": None,
        "launches_per_step": ln.value / n, "avg_launch_us": ms.value * 1e3 / ln.value,
        "gflop_per_step": fl.value / n / 1e9, "gemm_ms_per_step": ms.value / n,
    }


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=device)
    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    if dist is not None:
        dist.barrier()

    cpu_net, gpu_net = build_model(device)
    pics = make_pictures(args.frames, rank, device)
    pad_r, pad_b = gpu_net.get_padding_size(HEIGHT, WIDTH, 16)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(gpu_net, pics[i % len(pics)], QPS[i % len(QPS)], pad_b, pad_r)
    sync()
    t_enc = t_dec = 0.0
    nbytes = 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        x, qp = pics[i % len(pics)], QPS[i % len(QPS)]
        a = time.perf_counter()
        enc = gpu_net.compress(x, qp, pad_b, pad_r)
        b = time.perf_counter()
        gpu_net.decompress(enc["bit_stream"], {"height": HEIGHT, "width": WIDTH}, qp, enc["ec_parallel"])
        c = time.perf_counter()
        t_enc += b - a
        t_dec += c - b
        nbytes += len(enc["bit_stream"])
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        out = {
            "metric": "1080p YUV420 intra encode+decode pictures per second (DCVC-UF-Intra, real rANS "
                      "bit streams, q_index in {0,16,32,48,63})",
            "value": world * args.steps / elapsed, "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16",
            "data": "synthetic (seeded low-pass noise + pan, 8-bit YUV420; seeded random weights of the "
                    "reference DMCI architecture)",
            "config": {"workload": "DCVC-UF-Intra 1080p YUV420 on 1xMI355X per rank, q_index cycling "
                                   "{0,16,32,48,63}, skip_thres 0.15, one picture per step: compress + decompress",
                       "pictures_per_step": 1, "resolution": "%dx%d" % (WIDTH, HEIGHT)},
            # host-side split of one step (the decode of step i cannot overlap its own encode);
            # compress() returns when the bit stream is ready, its reconstruction may still be running
            "encode_fps_host_view": args.steps / t_enc, "decode_fps_host_view": args.steps / t_dec,
            "bytes_per_picture": nbytes / args.steps,
            "bpp": 8.0 * nbytes / args.steps / (HEIGHT * WIDTH),
        }
        if not args.no_roofline:
            out["roofline"] = roofline(gpu_net, pics, pad_b, pad_r)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cpu_net)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
