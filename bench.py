#!/usr/bin/env python
"""bench.py - DCVC-UF 1080p YUV420 encode+decode throughput on MI355X (default: DCVC-UF-Intra).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): DCVC-UF-Intra (DMCI, 42.2 M parameters, seeded synthetic
weights of the reference architecture), 1920x1080 YUV420 synthetic pictures, q_index cycling over
{0, 16, 32, 48, 63}, skip_thres 0.15 (the reference's runtime setting, test_compress_time.py:41).
One step = compress one picture to a real rANS bit stream + decompress it again
(DMCI.compress + DMCI.decompress of the reference surface, host entropy coding included),
pictures already resident in HBM as fp16 NHWC tensors. Every rank codes its own pictures
(all-intra pictures are independent: no data-path collective), value = pictures/s of the job.
--lanes L (default 1 = the reference's sequential loop) keeps L independent pictures in flight per
GPU (dcvc_amd/lanes.py: own codec objects, stream and host thread per lane; one step = every lane
codes one picture); the one-picture-in-flight rate is then reported beside it as "one_lane".

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
    p.add_argument("--steps", type=int, default=60)
    p.add_argument("--warmup", type=int, default=15)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--frames", type=int, default=5, help="distinct synthetic pictures per rank")
    p.add_argument("--lanes", type=int, default=int(os.environ.get("DCVC_BENCH_LANES", "0")),
                   help="independent coding lanes per GPU: each lane owns its codec objects, a HIP stream and a host "
                        "thread and codes its own pictures; one step = every lane codes one unit (a batch of "
                        "`lanes` units in flight on the GPU). 0 = default = 1: the reference's sequential "
                        "loop (measured sweep in profiles/README.md: 2 lanes +5..15 %% intra, +3..5 %% LD, "
                        "-15 %% HT over one lane)")
    p.add_argument("--workload", default="intra", choices=("intra", "ld", "hts", "htl"),
                   help="intra = the headline configuration (BASELINE.json configs[1]); ld / hts / htl = the inter "
                        "models of configs[2] (one step = one call: 1 picture for ld, a chunk of 8 for hts / htl)")
    return p.parse_args()


def _to_gpu(net, device):
    import copy
    g = copy.deepcopy(net).half().to(device)       # finalize_model, test_video.py:27-29
    g.proxy = None
    return g


def build_model(device):
    from dcvc_amd import arch, models, synthetic
    net = models.DMCI()
    net.load_state_dict(synthetic.synthetic_state_dict(arch.dmci_spec(), 0))
    net.update(SKIP_THRES)
    return net, _to_gpu(net, device)


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


class InterWorkload:
    """configs[2]: P pictures with the inter models. One step = one compress + one decompress call
    (separate encoder / decoder objects, the decoder sees only the bytes): 1 picture for LD, a chunk
    of 8 for HT-S / HT-L. Every `gop` steps both sides are re-seeded from an intra reconstruction
    (add_ref_feature_from_frame; the I picture itself is coded outside the timed region), LD resets
    its feature memory every 32 pictures like the reference default (test_video.py:148,232)."""

    def __init__(self, kind, device, pics, gpu_intra, pad_b, pad_r):
        from dcvc_amd import arch, models, synthetic
        self.kind, self.pics, self.pad_b, self.pad_r = kind, pics, pad_b, pad_r
        if kind == "ld":
            net = models.DMC()
            net.load_state_dict(synthetic.synthetic_state_dict(arch.dmc_ld_spec(), 0))
            self.frames, self.gop = 1, 96
        else:
            net = models.DMCHT(kind)
            net.load_state_dict(synthetic.synthetic_state_dict(arch.dmc_ht_spec(kind == "hts"), 0))
            self.frames, self.gop = 8, 12
        net.update(SKIP_THRES)
        self.enc, self.dec = _to_gpu(net, device), _to_gpu(net, device)
        ref = gpu_intra.compress(pics[0], 32, pad_b, pad_r)["x_hat"]
        self.ref = ref.clone()
        if self.frames == 1:
            self.inputs = pics
        else:
            self.inputs = [torch.cat([pics[(i + j) % len(pics)] for j in range(8)], dim=1).contiguous(
                memory_format=torch.channels_last) for i in range(len(pics))]
        # launch mode of the timed region: the codec's own default (LD launches eagerly, dmc_ld.hip:24-26)
        self.graphs = (kind != "ld" or bool(os.environ.get("DCVC_BENCH_GRAPHS"))) and not os.environ.get("DCVC_BENCH_EAGER")

    def step(self, i, qp):
        if i % self.gop == 0:
            self.enc.add_ref_feature_from_frame(self.ref)
            self.dec.add_ref_feature_from_frame(self.ref, apply_feature_adaptor=False)
        # picture index inside the GOP (0 = the I picture); (frame_idx + g_frame_delay) % reset_interval == 1
        reset = 1 if (self.frames == 1 and (i % self.gop + 1) % 32 == 0) else 0
        x = self.inputs[i % len(self.inputs)]
        enc = self.enc.compress(x, qp, reset, self.pad_b, self.pad_r)
        self.dec.decompress(enc["bit_stream"], {"height": HEIGHT, "width": WIDTH}, qp, enc["ec_parallel"], reset)
        return len(enc["bit_stream"])

    def set_use_graphs(self, on):
        for g in (self.enc, self.dec):
            g._ensure_proxy().set_use_graphs(on)


class IntraWorkload:
    frames, kind = 1, "intra"

    def __init__(self, gpu_net, pics, pad_b, pad_r):
        self.net, self.pics, self.pad_b, self.pad_r = gpu_net, pics, pad_b, pad_r
        self.graphs = not os.environ.get("DCVC_BENCH_EAGER")

    def step(self, i, qp):
        enc, _ = step(self.net, self.pics[i % len(self.pics)], qp, self.pad_b, self.pad_r)
        return len(enc["bit_stream"])

    def set_use_graphs(self, on):
        self.net._ensure_proxy().set_use_graphs(on)


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


def roofline(work):
    """conv_gemm launches of the workload's steps bracketed by HIP events (eager pass)."""
    from dcvc_amd import _lib
    en = _lib.fn("dcvc_gemm_profile_enable", ctypes.c_int, [ctypes.c_int])
    rs = _lib.fn("dcvc_gemm_profile_reset", ctypes.c_int, [])
    co = _lib.fn("dcvc_gemm_profile_collect", ctypes.c_int,
                 [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                  ctypes.POINTER(ctypes.c_longlong)])
    work.set_use_graphs(False)
    work.step(1, QPS[2])                                 # eager warm-up
    torch.cuda.synchronize()
    _lib.check(en(1))
    _lib.check(rs())
    n = 0
    for i, qp in enumerate(QPS):
        work.step(2 + i, qp)
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
    work.set_use_graphs(work.graphs)
    achieved = fl.value / (ms.value * 1e-3) / 1e12
    # HBM bytes per launch: PMC counters cannot be read from inside this process; the figure is the
    # rocprofv3 FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE average over the conv_gemm launches
    # of this same command, collected by tools/pmc_session.sh and committed under profiles/
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")) as f:
            traffic = json.load(f)["kernels"]["conv_gemm"]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    return {
        "bound": "mfma", "kernel": "conv_gemm_kernel", "achieved": achieved, "peak": MFMA_PEAK_TFLOPS,
        "unit": "TFLOP/s", "frac": achieved / MFMA_PEAK_TFLOPS, "traffic": traffic,
        "algorithmic_bytes_per_launch": None,
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
    if os.environ.get("DCVC_BENCH_USER_STREAM", "side") != "null":
        # as the reference harness (test_video.py:423-425): the process works on a non-default stream.
        # (From the legacy null stream the codec joins its results through a blocking stream,
        # CodecBase::leave; measured LD 266 pictures/s there vs 277 from a stream like this one.)
        torch.cuda.set_stream(torch.cuda.Stream(device))
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
    def make_work(lane):
        lane_pics = pics if lane == 0 else make_pictures(args.frames, rank + 1000 * lane, device)
        if args.workload == "intra":
            return IntraWorkload(gpu_net if lane == 0 else _to_gpu(cpu_net, device), lane_pics, pad_b, pad_r)
        return InterWorkload(args.workload, device, lane_pics, gpu_net, pad_b, pad_r)

    # args.lanes independent lanes per GPU (dcvc_amd/lanes.py): lane 0 is also the single-lane reference run
    import contextlib
    from dcvc_amd.lanes import LanePool
    if args.lanes == 0:
        args.lanes = 1
    one = args.lanes == 1 and not os.environ.get("DCVC_BENCH_POOL1")       # plain loop on the caller's stream
    pool = LanePool(args.lanes, (lambda k: None) if one else make_work, device)
    if one:
        pool.states[0] = make_work(0)
    work = pool.states[0]
    # the single-lane passes (one_lane, roofline) of a multi-lane run go through lane 0's own stream, as
    # the lane does in the timed region
    lane0 = (lambda: pool._on_lane(0)) if args.lanes > 1 else contextlib.nullcontext

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if os.environ.get("DCVC_BENCH_EAGER") or os.environ.get("DCVC_BENCH_GRAPHS"):
        for w in pool.states:      # experiments: force eager launches / hipGraph replay (LD launches eagerly by default)
            w.set_use_graphs(w.graphs)

    def run_steps(first, n):
        """every lane codes n units (lane k starts at qp offset k) -> coded bytes of all lanes."""
        if len(pool) == 1:
            with lane0():
                return sum(work.step(first + i, QPS[i % len(QPS)]) for i in range(n))
        return sum(pool.run_each(
            lambda k, w: sum(w.step(first + i, QPS[(i + k) % len(QPS)]) for i in range(n))))

    if one:
        with lane0():
            for i in range(args.warmup):
                work.step(i, QPS[i % len(QPS)])
    else:
        pool.warm(lambda k, w: [w.step(i, QPS[i % len(QPS)]) for i in range(args.warmup)])
    # Untimed single-lane passes on lane 0 (multi-lane runs: before the timed region, through lane 0's
    # own stream): the one-unit-in-flight rate and the per-launch kernel times.
    single, roof = None, None
    if rank == 0:
        if len(pool) > 1:
            n1 = max(5, args.steps // 2)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            with lane0():
                for i in range(n1):
                    work.step(args.warmup + i, QPS[i % len(QPS)])
            torch.cuda.synchronize()
            single = (time.perf_counter() - t1) / n1
        if not args.no_roofline and len(pool) > 1:
            with lane0():
                roof = roofline(work)
    sync()
    t0 = time.perf_counter()
    nbytes = run_steps(args.warmup, args.steps)
    sync()
    elapsed = time.perf_counter() - t0
    if rank == 0 and not args.no_roofline and len(pool) == 1:
        with lane0():
            roof = roofline(work)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        names = {"intra": "DCVC-UF-Intra (DMCI)", "ld": "DCVC-UF inter LD (DMC low-delay)",
                 "hts": "DCVC-UF inter HT-S (8-picture chunks)", "htl": "DCVC-UF inter HT-L (8-picture chunks)"}
        units = args.steps * args.lanes                 # units coded per rank in the timed region
        fps = world * units * work.frames / elapsed
        out = {
            "metric": "1080p YUV420 %s encode+decode pictures per second (%s, real rANS bit streams, "
                      "q_index in {0,16,32,48,63})" % ("intra" if args.workload == "intra" else "inter", names[args.workload]),
            "value": fps, "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16",
            "data": "synthetic (seeded low-pass noise + pan, 8-bit YUV420; seeded random weights of the "
                    "reference architecture)",
            "config": {"workload": "%s 1080p YUV420 on 1xMI355X per rank, q_index cycling {0,16,32,48,63}, "
                                   "skip_thres 0.15, one step = compress + decompress of %d picture(s)%s"
                                   % (names[args.workload], work.frames * args.lanes,
                                      "" if args.lanes == 1 else " (%d independent lanes x %d in flight)" % (args.lanes, work.frames)),
                       "pictures_per_step": work.frames * args.lanes, "lanes": args.lanes,
                       "resolution": "%dx%d" % (WIDTH, HEIGHT)},
            "bytes_per_picture": nbytes / units / work.frames,
            "bpp": 8.0 * nbytes / units / work.frames / (HEIGHT * WIDTH),
        }
        if single is not None:
            out["one_lane"] = {"value": work.frames / single, "unit": "frames/s per GPU", "ms_per_unit": 1e3 * single,
                               "note": "same workload, one unit in flight on the GPU (the reference's sequential loop)"}
        if roof is not None:
            out["roofline"] = roof
        if not args.no_cpu_baseline and args.workload == "intra" and world == 1:      # N = 1 only (contract)
            out["cpu_baseline"] = cpu_baseline(cpu_net)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
