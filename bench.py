#!/usr/bin/env python
"""bench.py - DCVC-UF 1080p YUV420 encode + decode throughput on MI355X.

  python bench.py --gpus N --steps K --warmup W [--workload intra|ld|hts|htl] [--resolution 3840x2160]
  N > 1: either under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (RANK /
  LOCAL_RANK / WORLD_SIZE in the environment), or plainly `python bench.py --gpus N ...`: without WORLD_SIZE the
  process re-launches itself under torch.distributed.run, one rank per GPU over RCCL (launch_ranks()).

Headline workload (BASELINE.json configs[1]): DCVC-UF-Intra (DMCI, 42.2 M parameters, seeded synthetic
weights of the reference architecture), 1920x1080 YUV420 synthetic pictures, q_index cycling over
{0, 16, 32, 48, 63}, skip_thres 0.15 (the reference's runtime setting, test_compress_time.py:41).
One step = compress one picture to a real rANS bit stream + decompress it again (DMCI.compress +
DMCI.decompress of the reference surface, host entropy coding included), pictures resident in HBM
as fp16 NHWC tensors. Every rank codes its own pictures (independent units, no data-path collective);
`value` = pictures/s of the whole job over the K timed steps (barrier + synchronize on both sides,
max over ranks). --workload ld|hts|htl: configs[2] with the inter models (one step = one compress +
one decompress call on separate encoder / decoder objects: 1 picture for LD, a chunk of 8 for HT).

The JSON line (rank 0) carries, besides the driver contract:
  encode_fps / decode_fps   SURVEY 8d's metric, measured the reference's way (test_video.py:261-265,
                            321-325, 380-388; test_compress_time.py:60-69): device synchronised, events
                            around every compress / decompress call, the first 4 calls dropped,
                            pictures per call / mean call time. A separate pass after the timed region.
  other_workloads           the same three numbers for the other three models (short runs; N = 1 only)
  roofline                  the DOMINANT contraction kernel (most time per step): its algorithmic FLOPs / the
                            HIP-event time of its launches, stamped live on the codec's stream by
                            hipExtLaunchKernelGGL in an extra eager pass; `all_contractions` = every contraction
                            launch together (> 99 % of the FLOPs), `kernels` = the per-kernel split; HBM
                            traffic from the committed PMC pass named in `traffic_source` (null when the
                            kernel sources changed since that pass)
  sustained                 the same loop run for >= --min-seconds after the K timed steps (the K-step region
                            of a short driver run is a fraction of a second)
  uhd                       short 3840x2160 runs of all four workloads (BASELINE configs[4]'s resolution)
  cpu_baseline              the reference's CPU-runnable path (fp32 graph forward_one_frame, restated in
                            oracle/torch_graph.py) on this host: all cores = `value`, one thread beside it
                            (the reference's set_torch_env pins 1, common.py:270), and the bit-exact
                            oracle's compress + decompress; N = 1 only, bounded samples
"""
import argparse
import contextlib
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

HEIGHT, WIDTH = 1080, 1920          # default resolution (BASELINE configs[1]); --resolution overrides
QPS = (0, 16, 32, 48, 63)
SKIP_THRES = 0.15
MFMA_PEAK_TFLOPS = 2500.0        # dense fp16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
DROP_CALLS = 4                   # test_video.py:380: the first 4 calls are warm-up
NAMES = {"intra": "DCVC-UF-Intra (DMCI)", "ld": "DCVC-UF inter LD (DMC low-delay)",
         "hts": "DCVC-UF inter HT-S (8-picture chunks)", "htl": "DCVC-UF inter HT-L (8-picture chunks)"}


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=100)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--workload", default="intra", choices=tuple(NAMES),
                   help="intra = the headline configuration (BASELINE.json configs[1]); ld / hts / htl = configs[2]")
    p.add_argument("--frames", type=int, default=5, help="distinct synthetic pictures per rank")
    p.add_argument("--fanout", action="store_true",
                   help="hts / htl with --gpus N > 1: ONE stream, the 8 reconstruction heads of a chunk spread over the ranks "
                        "(feature_p broadcast over RCCL; strong scaling) instead of N independent streams")
    p.add_argument("--one-codec", action="store_true",
                   help="intra: ONE codec object codes and decodes, one call after the other (as the reference harness uses its "
                        "i_frame_net). Default since round 4: separate encoder / decoder objects run as a two-stage pipeline")
    p.add_argument("--two-codecs", action="store_true", help="(the default; kept for old command lines)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--no-extras", action="store_true", help="skip the short runs of the other three workloads")
    p.add_argument("--resolution", default="%dx%d" % (WIDTH, HEIGHT), help="WxH of the synthetic pictures (3840x2160 = configs[4])")
    p.add_argument("--no-uhd", action="store_true", help="skip the short 3840x2160 runs of the default line")
    p.add_argument("--min-seconds", type=float, default=2.0,
                   help="length of the `sustained` region behind the K timed steps (0 = none)")
    a = p.parse_args()
    try:
        w, h = (int(v) for v in a.resolution.lower().split("x"))
        assert w > 0 and h > 0 and w % 2 == 0 and h % 2 == 0
    except (ValueError, AssertionError):
        raise SystemExit("--resolution wants WxH with even W and H, e.g. 1920x1080")
    a.width, a.height = w, h
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    return a


def launch_ranks(args):
    """`python bench.py --gpus N` with no launcher around it: re-run this very command line under
    torch.distributed.run, one rank per GPU (the reference's scaling mode is one worker process per GPU,
    test_video.py:407-420,496-500). Rank 0's JSON line goes straight to our stdout."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(sys.argv[0])] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


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


def make_pictures(n, rank, device, height=HEIGHT, width=WIDTH):
    from dcvc_amd import synthetic
    pics = []
    for i in range(n):
        y, uv = synthetic.synthetic_frame_yuv420(height, width, index=i, seed=rank)
        x = synthetic.yuv420_to_x(y, uv).half().to(device)
        pics.append(x.contiguous(memory_format=torch.channels_last))
    return pics


class IntraWorkload:
    """configs[1]: every picture is an I picture. Default (round 4): a decoder object of its own (`dec_net`), driven as the
    second stage of run_steps_overlapped() with its compute stream at high priority: the decoder's four host round trips per
    picture (0.6 ms of idle GPU) are filled with the next picture's encoder kernels - 143.6 -> 148.1 pictures/s (sustained
    150.0) on one box. The same two objects called one after the other from one thread are SLOWER than one object (125;
    round 1 measured the same: 105.8 vs 119.6) - the decoder's short kernels queue behind the encoder's reconstruction tail at
    equal priority. `--one-codec` (`dec_net=None`): one object codes and decodes, as the reference harness uses its i_frame_net."""
    frames, kind = 1, "intra"

    def __init__(self, gpu_net, pics, pad_b, pad_r, dec_net=None):
        self.net, self.pics, self.pad_b, self.pad_r = gpu_net, pics, pad_b, pad_r
        self.dec = dec_net if dec_net is not None else gpu_net
        self.height, self.width = int(pics[0].shape[2]), int(pics[0].shape[3])
        self.sps = {"height": self.height, "width": self.width}
        # --two-codecs: a decoder object of its own can run as the second stage of run_steps_overlapped()
        self.overlapped = dec_net is not None
        if self.overlapped and not os.environ.get("DCVC_BENCH_SEQUENTIAL") and os.environ.get("DCVC_BENCH_PRIORITIES", "1") != "0":
            os.environ["DCVC_COMPUTE_PRIORITY"] = "high"
            self.dec._ensure_proxy()
            del os.environ["DCVC_COMPUTE_PRIORITY"]

    def prepare(self, i):
        pass

    prepare_enc = prepare_dec = prepare

    def compress(self, i, qp):
        return self.net.compress(self.pics[i % len(self.pics)], qp, self.pad_b, self.pad_r)

    def decompress(self, i, qp, enc):
        return self.dec.decompress(enc["bit_stream"], self.sps, qp, enc["ec_parallel"])

    def set_use_graphs(self, on):
        for g in {id(self.net): self.net, id(self.dec): self.dec}.values():
            g._ensure_proxy().set_use_graphs(on)

    def closure(self, i, qp):
        """decoder reconstruction == encoder reconstruction, bit for bit (what the reference asserts nowhere but relies on)"""
        enc = self.compress(i, qp)
        want = enc["x_hat"].clone()             # proxy-owned buffer: the decode below may overwrite it
        got = self.decompress(i, qp, enc)["x_hat"]
        torch.cuda.synchronize()
        return bool(torch.equal(got, want)) and bool(torch.isfinite(got.float()).all()) and len(enc["bit_stream"]) > 0

    default_graphs = True


class InterWorkload:
    """configs[2]: P pictures with the inter models, separate encoder / decoder objects (the decoder
    sees only the bytes). Every `gop` steps both sides are re-seeded from an intra reconstruction
    (add_ref_feature_from_frame; the I picture itself is coded outside the timed calls); the feature memory is
    reset at the reference's cadence (test_video.py:148,232-235: reset_interval 32, a call resets when
    (frame_idx + g_frame_delay) % reset_interval == 1 - every 32nd picture for LD, every 4th chunk for HT)."""
    RESET_INTERVAL = 32

    def __init__(self, kind, device, pics, gpu_intra, pad_b, pad_r):
        from dcvc_amd import arch, models, synthetic
        self.kind, self.pad_b, self.pad_r = kind, pad_b, pad_r
        self.height, self.width = int(pics[0].shape[2]), int(pics[0].shape[3])
        self.sps = {"height": self.height, "width": self.width}
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
        if self.overlapped and not os.environ.get("DCVC_BENCH_SEQUENTIAL") and os.environ.get("DCVC_BENCH_PRIORITIES", "1") != "0":
            # the decoder's chain of short kernels and host round trips goes first, the encoder fills the gaps (the
            # priority of a codec's compute stream is read when the native object is created)
            for obj, prio in ((self.enc, "low"), (self.dec, "high")):
                os.environ["DCVC_COMPUTE_PRIORITY"] = prio
                obj._ensure_proxy()
            del os.environ["DCVC_COMPUTE_PRIORITY"]
        self.ref = gpu_intra.compress(pics[0], 32, pad_b, pad_r)["x_hat"].clone()
        if self.frames == 1:
            self.inputs = pics
        else:
            self.inputs = [torch.cat([pics[(i + j) % len(pics)] for j in range(8)], dim=1).contiguous(
                memory_format=torch.channels_last) for i in range(len(pics))]
        # launch mode: the codec's own default (LD launches eagerly, dmc_ld.hip:24-26)
        self.default_graphs = kind != "ld"

    def _reset(self, i):
        # step i codes the pictures frame_idx .. frame_idx + frames - 1 of its GOP, frame 0 being the I picture
        frame_idx = 1 + self.frames * (i % self.gop)
        return 1 if (frame_idx + self.frames) % self.RESET_INTERVAL == 1 else 0

    # encoder and decoder are separate objects that share nothing but the bytes: run_steps_overlapped() drives them as a
    # two-stage pipeline from two host threads (a real deployment runs them on different machines)
    overlapped = True

    def prepare_enc(self, i):
        if i % self.gop == 0:
            self.enc.add_ref_feature_from_frame(self.ref)

    def prepare_dec(self, i):
        if i % self.gop == 0:
            self.dec.add_ref_feature_from_frame(self.ref, apply_feature_adaptor=False)

    def prepare(self, i):
        self.prepare_enc(i)
        self.prepare_dec(i)

    def compress(self, i, qp):
        return self.enc.compress(self.inputs[i % len(self.inputs)], qp, self._reset(i), self.pad_b, self.pad_r)

    def decompress(self, i, qp, enc):
        return self.dec.decompress(enc["bit_stream"], self.sps, qp, enc["ec_parallel"], self._reset(i))

    def set_use_graphs(self, on):
        for g in (self.enc, self.dec):
            g._ensure_proxy().set_use_graphs(on)

    def closure(self, i, qp):
        """encoder / decoder lock-step: the decoder (which saw only the bytes) holds the very feature_p the encoder holds -
        every reconstruction head reads nothing else (video_model_ht.py:252-275) - and finite in-range pictures"""
        self.prepare(i)
        enc = self.compress(i, qp)
        xd = self.decompress(i, qp, enc)["x_hat"]
        torch.cuda.synchronize()
        xd = torch.cat(list(xd), 0) if isinstance(xd, (list, tuple)) else xd
        fe = self.enc._ensure_proxy().debug_read("feature_p", np.float16)
        fd = self.dec._ensure_proxy().debug_read("feature_p", np.float16)
        return bool(np.array_equal(fe, fd)) and bool(torch.isfinite(xd.float()).all()) and float(xd.abs().max()) <= 0.5 \
            and len(enc["bit_stream"]) > 0


class FanoutWorkload(InterWorkload):
    """ONE hierarchical stream decoded over all ranks. Rank 0 owns the stream (encoder and the
    decoder's entropy / prior / decoder stages, temporal state); it broadcasts feature_p over RCCL while its own
    reconstruction heads run, and every rank reconstructs its share of the 8 pictures
    (dcvc_amd/sharding.py decompress_fanout). Strong scaling: the work of a step does not grow with the number of GPUs."""
    overlapped = False          # every rank takes part in every decompress call: one thread

    def __init__(self, kind, device, pics, gpu_intra, pad_b, pad_r, dist):
        super().__init__(kind, device, pics, gpu_intra, pad_b, pad_r)
        self.dist, self.rank = dist, dist.get_rank()
        if self.rank != 0:
            self.enc = None

    def prepare(self, i):
        if self.rank == 0:
            super().prepare(i)

    def compress(self, i, qp):
        if self.rank != 0:
            return {"bit_stream": b"", "ec_parallel": 0}
        return super().compress(i, qp)

    def decompress(self, i, qp, enc):
        from dcvc_amd import sharding
        return sharding.decompress_fanout(self.dec._ensure_proxy(), np.frombuffer(enc["bit_stream"], dtype=np.uint8), qp,
                                          self.height, self.width, enc["ec_parallel"], bool(self._reset(i)), self.dist)

    def set_use_graphs(self, on):
        for g in (self.enc, self.dec):
            if g is not None:
                g._ensure_proxy().set_use_graphs(on)


def run_steps(work, first, n):
    """n pipelined steps (no host synchronisation between them) -> coded bytes"""
    nbytes = 0
    for i in range(first, first + n):
        qp = QPS[i % len(QPS)]
        work.prepare(i)
        enc = work.compress(i, qp)
        work.decompress(i, qp, enc)
        nbytes += len(enc["bit_stream"])
    return nbytes


def run_steps_overlapped(work, first, n, depth=int(os.environ.get("DCVC_BENCH_DEPTH", "2"))):
    """The same n steps as a two-stage pipeline: an encoder thread codes step i + 1 (its own stream) while this thread
    decodes step i. What it buys: the decoder's host round trips - entropy decoding of a whole P picture is 0.7 ms with
    the GPU idle (profiles/r04_ld_timeline.txt) - are filled with the encoder's kernels of the next picture, and the
    encoder's host entropy coding with the decoder's kernels. Encoder and decoder objects share nothing but the bytes (a
    fresh numpy copy per call); both see the steps in order, so GOP re-seeding and memory resets stay in lock-step. The
    codec calls are ctypes calls: the GIL is released inside them."""
    import queue
    import threading
    q = queue.Queue(maxsize=depth)
    device = torch.cuda.current_device()
    stream = getattr(work, "_enc_stream", None)
    if stream is None:
        stream = work._enc_stream = torch.cuda.Stream(device)
    failed = []

    def encoder():
        try:
            torch.cuda.set_device(device)
            with torch.cuda.stream(stream):
                for i in range(first, first + n):
                    qp = QPS[i % len(QPS)]
                    work.prepare_enc(i)
                    q.put((i, qp, work.compress(i, qp)))
        except BaseException as e:           # noqa: BLE001 - handed to the caller's thread
            failed.append(e)
        finally:
            q.put(None)

    t = threading.Thread(target=encoder, name="bench-encoder")
    t.start()
    nbytes = 0
    try:
        while True:
            item = q.get()
            if item is None:
                break
            i, qp, enc = item
            work.prepare_dec(i)
            work.decompress(i, qp, enc)
            nbytes += len(enc["bit_stream"])
    finally:
        while t.is_alive():                  # a decoder failure must not leave the encoder blocked on a full queue
            try:
                q.get(timeout=0.1)
            except queue.Empty:
                pass
        t.join()
    if failed:
        raise failed[0]
    return nbytes


def step_loop(work):
    """run_steps for one codec object / the fan-out, the two-stage pipeline for separate encoder / decoder objects
    (DCVC_BENCH_SEQUENTIAL=1: always the plain loop - the A/B partner)"""
    if getattr(work, "overlapped", False) and not os.environ.get("DCVC_BENCH_SEQUENTIAL"):
        return run_steps_overlapped
    return run_steps


def call_times(work, first, n):
    """The reference's timing loop (test_video.py:224-265, 300-325): synchronise, event, call, event,
    synchronise - per call. Returns mean seconds per compress / per decompress call, first 4 dropped."""
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    enc_t, dec_t = [], []
    for i in range(first, first + n):
        qp = QPS[i % len(QPS)]
        work.prepare(i)
        torch.cuda.synchronize()
        ev[0].record()
        enc = work.compress(i, qp)
        ev[1].record()
        torch.cuda.synchronize()
        ev[2].record()
        work.decompress(i, qp, enc)
        ev[3].record()
        torch.cuda.synchronize()
        enc_t.append(ev[0].elapsed_time(ev[1]) * 1e-3)
        dec_t.append(ev[2].elapsed_time(ev[3]) * 1e-3)
    keep = slice(DROP_CALLS, None) if n > DROP_CALLS else slice(0, None)
    return float(np.mean(enc_t[keep])), float(np.mean(dec_t[keep]))


def box_identity(device):
    """which machine produced the line (boxes of one pool differ by +- 8 % on this workload: power / clock policy)"""
    out = {}
    try:
        pr = torch.cuda.get_device_properties(device)
        out["gpu"] = getattr(pr, "name", None)
        uuid = getattr(pr, "uuid", None)
        out["gpu_uuid"] = str(uuid) if uuid is not None else None
        out["gcn_arch"] = getattr(pr, "gcnArchName", None)
    except Exception as e:          # noqa: BLE001 - identification only
        out["gpu"] = "unknown (%s)" % type(e).__name__
    try:
        with open("/proc/cpuinfo") as f:
            names = [l.split(":", 1)[1].strip() for l in f if l.startswith("model name")]
        out["host_cpu"] = names[0] if names else None
        out["host_threads"] = len(names)
    except OSError:
        pass
    try:
        with open(os.path.join(ROOT, ".git_head")) as f:
            out["commit"] = f.read().strip()
    except OSError:
        pass
    return out


def closure_ok(work, first):
    """After a timed region: one compress / decompress per rate point of QPS, checked (a throughput number from
    a desynchronised codec would otherwise look like any other). Continues the workload's stream at step `first`."""
    return all([work.closure(first + k, qp) for k, qp in enumerate(QPS)])


def fps_block(work, first, steps, warmup, with_roofline=False):
    """throughput (pipelined loop) + the reference-style encode / decode rates of one workload"""
    run_steps(work, first, warmup)           # (plain loop: graph capture of both objects from one thread)
    loop = step_loop(work)
    if loop is not run_steps:
        loop(work, first + warmup, 2)        # the pipeline's own set-up outside the timed region
        first += 2
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nbytes = loop(work, first + warmup, steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    first += steps
    ncalls = min(steps, 24) + DROP_CALLS
    te, td = call_times(work, first + warmup, ncalls)
    out = {"value": steps * work.frames / dt, "unit": "frames/s", "steps": steps, "ms_per_step": 1e3 * dt / steps,
           # (the plain loop is not measured on these objects: with the encoder's stream at low priority it runs slower than
           # it does on unprioritised objects - DCVC_BENCH_SEQUENTIAL=1 python bench.py --workload W is the A/B partner)
           "loop": "two-stage pipeline (encoder thread | decoder thread)" if loop is not run_steps else "one call after the other",
           "encode_fps": work.frames / te, "decode_fps": work.frames / td,
           "bpp": 8.0 * nbytes / steps / work.frames / (work.height * work.width),
           "closure_ok": closure_ok(work, first + warmup + ncalls)}
    if with_roofline:
        r = roofline(work, n=2)
        out["roofline"] = {k: r[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "all_contractions")}
    return out


def cpu_baseline(cpu_net):
    """The reference's CPU path (fp32 graph) and the bit-exact oracle on this host, bounded samples."""
    from oracle import codec, torch_graph
    from dcvc_amd import synthetic
    cores = os.cpu_count() or 1
    sd = cpu_net.state_dict()
    full = HEIGHT * WIDTH
    # thread count: the fastest of {64, 32, 16} on a 512x512 crop, then one padded 1080p picture (1088x1920) with it.
    # (All 256 hardware threads of the GPU box were measured once: 114 s for the CROP - one small convolution per op
    # oversubscribes a 2-socket host hopelessly - so that trial is not repeated in every run.) One thread: a 256x256
    # crop, scaled by area
    tried = {t: torch_graph.time_forward(sd, 512, 512, 32, t) for t in sorted({min(cores, c) for c in (64, 32, 16)})}
    best = min(tried, key=tried.get)
    t_all = torch_graph.time_forward(sd, 1088, 1920, 32, best)
    t_one = torch_graph.time_forward(sd, 256, 256, 32, 1) * (1088 * 1920) / (256 * 256)
    h = w = 160
    y, uv = synthetic.synthetic_frame_yuv420(h, w, 0, 0)
    x = synthetic.yuv420_to_x(y, uv)[0].permute(1, 2, 0).contiguous().numpy().astype(np.float16)
    o = codec.DMCIOracle(sd, SKIP_THRES, cpu_net.get_cdf_info())
    t0 = time.time()
    r = o.compress(x, 32)
    o.decompress(r["bit_stream"], 32, h, w, r["ec_parallel"])
    t_orc = (time.time() - t0) * full / (h * w)
    return {
        "value": 1.0 / t_all, "unit": "frames/s", "cores": best, "kind": "port",
        "sample": "fp32 PyTorch graph of the reference's CPU-runnable path (DMCI.forward_one_frame, image_model.py:150-171, "
                  "restated in oracle/torch_graph.py: encoder + priors + decoder of one 1088x1920 picture, no entropy coding) "
                  "on %d of %d hardware threads (fastest of %s on a 512x512 crop%s): %.2f s per picture"
                  % (best, cores, {t: round(v, 2) for t, v in tried.items()},
                     "; all 256 threads of the GPU box: 114 s for the crop, measured once" if cores > 64 else "", t_all),
        "one_thread": {"value": 1.0 / t_one, "unit": "frames/s", "cores": 1,
                       "sample": "the same graph on 1 thread (the reference harness pins 1, common.py:270): one 256x256 crop, "
                                 "scaled by area to 1080p (%.1f s per picture)" % t_one},
        "oracle": {"value": 1.0 / t_orc, "unit": "frames/s", "cores": cores, "kind": "port",
                   "sample": "bit-exact oracle (oracle/codec.py, matrix-core arithmetic emulated in integer SIMD) compress + decompress "
                             "of one %dx%d crop scaled by area to 1080p (%.0f s per picture)" % (h, w, t_orc)},
    }


TRAFFIC_FILE = os.path.join("profiles", "r04_hbm_traffic.json")
KERNEL_SOURCES = ("dcb_nsplit8_kernel.h", "dcb_nsplit_kernel.h", "dcb_nsplit.hip", "conv_gemm.hip", "dcb_tail.hip", "ffn_fused.hip", "dwconv.hip",
                  "arith.h")


def kernel_source_digest():
    """sha256 over the contraction kernels' sources: ties a committed PMC pass to the code it measured"""
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "dcvc_amd", "csrc", "kernels", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (tools/hbm_traffic.py), or
    (None, reason) when there is no pass for the kernel sources as they are now"""
    try:
        with open(os.path.join(ROOT, TRAFFIC_FILE)) as f:
            t = json.load(f)
        if t.get("kernel_source_digest") != kernel_source_digest():
            return None, "%s was collected for other kernel sources (digest %s, now %s): not reported" % (
                TRAFFIC_FILE, t.get("kernel_source_digest"), kernel_source_digest())
        return t["kernels"][kernel]["hbm_bytes_per_launch"], "%s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes of " \
            "`bench.py --steps 5`, kernel sources %s)" % (TRAFFIC_FILE, t["kernel_source_digest"])
    except (OSError, KeyError, ValueError) as e:
        return None, "no PMC pass on file (%s)" % type(e).__name__


KERNEL_NAMES = {0: "conv_gemm_kernel", 1: "dcb_core_kernel", 2: "dcb_tail_kernel", 3: "ffn_fused_kernel", 4: "dcb_nsplit_kernel",
                5: "dcb_nsplit8_kernel"}


def roofline(work, n=len(QPS)):
    """contraction launches of n steps bracketed by HIP events (eager pass on the codec's stream)"""
    from dcvc_amd import _lib
    en = _lib.fn("dcvc_gemm_profile_enable", ctypes.c_int, [ctypes.c_int])
    rs = _lib.fn("dcvc_gemm_profile_reset", ctypes.c_int, [])
    ln = _lib.fn("dcvc_gemm_profile_launches", ctypes.c_longlong, [ctypes.c_void_p, ctypes.c_longlong])
    work.set_use_graphs(False)
    run_steps(work, 1, 1)                                # eager warm-up
    torch.cuda.synchronize()
    _lib.check(en(1))
    _lib.check(rs())
    run_steps(work, 2, n)
    torch.cuda.synchronize()
    rec = np.dtype([("M", np.int32), ("N", np.int32), ("K", np.int32), ("variant", np.int32), ("ms", np.float32)])
    buf = np.zeros(131072, dtype=rec)
    used = int(ln(buf.ctypes.data, len(buf)))
    buf = buf[:used]
    _lib.check(en(0))
    work.set_use_graphs(work.default_graphs)
    flops = 2.0 * buf["M"].astype(np.float64) * buf["N"] * buf["K"]
    # variant bits 28..31: the kernel family of the launch (ops.h GemmLaunchInfo); 8 = dcb_core (the sign bit)
    fam = (buf["variant"].astype(np.int64) >> 28) & 0xF
    family = np.where(fam >= 8, 1, fam)
    if os.environ.get("DCVC_BENCH_SHAPES"):
        agg = {}
        for r, f in zip(buf, flops):
            k = (int(r["M"]), int(r["N"]), int(r["K"]), int(r["variant"]))
            a = agg.setdefault(k, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += float(r["ms"])
            a[2] += f
        with open(os.environ["DCVC_BENCH_SHAPES"], "w") as fh:
            fh.write("M,N,K,variant,calls_per_step,avg_us,tflops,ms_per_step\n")
            for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                fh.write("%d,%d,%d,%d,%.1f,%.2f,%.1f,%.3f\n" % (k + (a[0] / n, 1e3 * a[1] / a[0], a[2] / (a[1] * 1e-3) / 1e12, a[1] / n)))

    def part(sel, name, alg_bytes):
        ms = float(buf["ms"][sel].sum())
        fl = float(flops[sel].sum())
        cnt = int(sel.sum())
        if cnt == 0:
            return None
        return {"kernel": name, "launches_per_step": cnt / n, "avg_launch_us": 1e3 * ms / cnt, "ms_per_step": ms / n,
                "gflop_per_step": fl / n / 1e9, "achieved": fl / (ms * 1e-3) / 1e12, "frac": fl / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS,
                "algorithmic_bytes_per_launch": alg_bytes}
    # algorithmic bytes per launch (every operand once). dcb_core: the depthwise output and the block input
    # in, the block output and the next block's dc.0 output out ([P8][384] fp16 each) + 2.06 MB of weights
    P8 = ((work.height + 15) // 16 * 2) * ((work.width + 15) // 16 * 2)
    core_bytes = 4 * P8 * 384 * 2 + 7 * 384 * 384 * 2
    kernels = []
    for f, name in KERNEL_NAMES.items():
        sel = family == f
        if f in (4, 5):
            # one entry per shape of the N-split block kernel: <C, CI, pixels per workgroup> follows from (N, K, M): K = 7 CI
            # with the next block's dc.0 inside the launch, 6 CI without (dcb_nsplit_kernel.h launch())
            inner = {k * ci: ci for ci in (128, 256, 384, 512, 768) for k in (6, 7)}
            shapes = sorted({(int(a), int(b), inner[int(c)]) for a, b, c in zip(buf["M"][sel], buf["N"][sel], buf["K"][sel])})
            for (m, c, ci) in shapes:
                one = sel & (buf["M"] == m) & (buf["N"] == c) & ((buf["K"] == 6 * ci) | (buf["K"] == 7 * ci))
                nxt = float((buf["K"][one] == 7 * ci).mean())            # share of the launches with the next dc.0
                wide = m >= 64 * 200 and c < 768
                # every operand once: t2 [M][CI], x [M][C] in, y [M][C] (and t1' [M][CI]) out, the block's weights
                alg = 2 * m * (2 * c + ci + nxt * ci) + 2 * c * ci * (6 + nxt)
                k = part(one, "%s<%d, %d, %d px>" % (name, c, ci, 64 if wide else 32), alg)
                if k:
                    k["pixels"] = m
                    k["with_next_dc0"] = nxt
                    kernels.append(k)
            continue
        k = part(sel, name, core_bytes if f == 1 else None)
        if k:
            kernels.append(k)
    total_ms, total_fl = float(buf["ms"].sum()), float(flops.sum())
    dom = max(kernels, key=lambda k: k["ms_per_step"])
    traffic, source = pmc_traffic(dom["kernel"])
    return {"bound": "mfma", "kernel": dom["kernel"], "achieved": dom["achieved"], "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": dom["frac"], "traffic": traffic, "traffic_source": source,
            "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"],
            "note": "achieved / frac / traffic = the dominant kernel's own launches (`kernel`: the contraction kernel with the most "
                    "time per step); all_contractions = every contraction launch of the step together (> 99 % of the FLOPs)",
            "all_contractions": {"achieved": total_fl / (total_ms * 1e-3) / 1e12, "frac": total_fl / (total_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS,
                                 "launches_per_step": used / n, "gflop_per_step": total_fl / n / 1e9, "ms_per_step": total_ms / n},
            "kernels": kernels}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d: the launcher's world size is what runs" % (args.gpus, world), file=sys.stderr)
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    if os.environ.get("DCVC_BENCH_ONE_DEVICE"):
        local_rank = 0           # tests/test_bench_gpu.py: the whole N-rank launch path on a 1-GPU box (with the gloo backend)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # as the reference harness (test_video.py:423-425): the process works on a non-default stream
    torch.cuda.set_stream(torch.cuda.Stream(device))
    dist, comm_device = None, device
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("DCVC_BENCH_BACKEND", "nccl")       # "gloo": the CPU test of the launch path
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend)
            comm_device = torch.device("cpu")
    import __graft_entry__
    if rank == 0:
        with contextlib.redirect_stdout(sys.stderr):     # stdout carries the ONE JSON line and nothing else
            __graft_entry__.build()
    if dist is not None:
        dist.barrier()

    height, width = args.height, args.width
    cpu_net, gpu_net = build_model(device)

    def make_work(kind, h, w, frames=args.frames):
        pics = make_pictures(frames, rank, device, h, w)
        pad_r, pad_b = gpu_net.get_padding_size(h, w, 16)
        if kind == "intra":
            return IntraWorkload(gpu_net, pics, pad_b, pad_r, None if args.one_codec else _to_gpu(cpu_net, device))
        return InterWorkload(kind, device, pics, gpu_net, pad_b, pad_r)

    fanout = args.fanout and world > 1
    if args.fanout and args.workload not in ("hts", "htl"):
        raise SystemExit("--fanout needs --workload hts or htl (the models with 8 reconstruction heads per call)")
    if fanout:
        pics = make_pictures(args.frames, rank, device, height, width)
        pad_r, pad_b = gpu_net.get_padding_size(height, width, 16)
        work = FanoutWorkload(args.workload, device, pics, gpu_net, pad_b, pad_r, dist)
    else:
        work = make_work(args.workload, height, width)
    if os.environ.get("DCVC_BENCH_GRAPHS") in ("0", "1"):     # A/B of the codecs' launch mode (hipGraph replay / eager)
        work.default_graphs = os.environ["DCVC_BENCH_GRAPHS"] == "1"
        work.set_use_graphs(work.default_graphs)
    # independent streams: rank r codes the steps shard_range() gives it out of world * steps (weak scaling,
    # no data-path collective); fan-out: every rank takes part in every step
    from dcvc_amd import sharding
    mine = range(args.steps) if fanout else sharding.shard_range(world * args.steps, rank, world)
    assert len(mine) == args.steps

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds):
        if dist is None:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=comm_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    run_steps(work, 0, args.warmup)          # (plain loop: the codecs capture their graphs from one thread)
    loop = step_loop(work)
    if loop is not run_steps:
        # two more untimed steps through the pipeline itself: the encoder thread's stream and the first hand-overs between
        # the two threads are set up outside the timed region (they cost ~ 7 ms of a 20-step region otherwise)
        loop(work, args.warmup, 2)
        mine = range(mine.start + 2, mine.stop + 2)
    sync()
    t0 = time.perf_counter()
    nbytes = loop(work, args.warmup + mine.start, args.steps)
    sync()
    elapsed = max_over_ranks(time.perf_counter() - t0)

    # the same loop for >= --min-seconds: the K-step region of a short driver run lasts a fraction of a second
    sustained = None
    if args.min_seconds > 0:
        more = max(args.steps, int(np.ceil(1.15 * args.min_seconds * args.steps / max(elapsed, 1e-6))))      # (a margin: warm steps run faster)
        more = min(more, 100 * args.steps)
        sync()
        t0 = time.perf_counter()
        loop(work, args.warmup + mine.start + args.steps, more)
        sync()
        t_more = max_over_ranks(time.perf_counter() - t0)
        sustained = {"steps": more, "seconds": t_more, "value": (1 if fanout else world) * more * work.frames / t_more,
                     "unit": "frames/s"}

    # step index behind everything this rank has coded so far (the inter workloads' GOP / reset cadence follows the index)
    cursor = args.warmup + mine.start + args.steps + (sustained or {}).get("steps", 0)
    # beside the two-stage pipeline (rank 0, short, intra): ONE codec object coding and decoding, as the reference harness does.
    # (Inter models: DCVC_BENCH_SEQUENTIAL=1 is the A/B partner - on objects whose streams carry priorities the plain loop
    # runs slower than it does without them.)
    plain = None
    if rank == 0 and loop is not run_steps and args.workload == "intra":
        n_plain = min(args.steps, 30)
        w_plain = IntraWorkload(work.net, work.pics, work.pad_b, work.pad_r, None)
        run_steps(w_plain, cursor, 4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(w_plain, cursor + 4, n_plain)
        torch.cuda.synchronize()
        plain = {"value": n_plain * work.frames / (time.perf_counter() - t0), "unit": "frames/s (this rank)", "steps": n_plain,
                 "loop": "one codec object, compress then decompress (the reference harness's way)"}
    ncalls = min(args.steps, 32) + DROP_CALLS
    if fanout:       # every rank takes part in the per-call timing loop (the broadcast is a collective)
        te, td = call_times(work, cursor, ncalls)
    # every rank checks ITS codec objects after the timed regions (fan-out: the shared stream is checked by the tests)
    closure = None if fanout else closure_ok(work, cursor + ncalls)
    if dist is not None and closure is not None:
        flag = torch.tensor([1.0 if closure else 0.0], dtype=torch.float64, device=comm_device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        closure = bool(flag.item() > 0.5)
    if rank == 0:
        if not fanout:
            te, td = call_times(work, cursor, ncalls)
        fps = (1 if fanout else world) * args.steps * work.frames / elapsed
        res = "%dx%d" % (width, height)
        out = {
            "metric": "%s YUV420 %s encode+decode pictures per second (%s, real rANS bit streams, q_index in {0,16,32,48,63})"
                      % ("1080p" if (height, width) == (1080, 1920) else res, "intra" if args.workload == "intra" else "inter",
                         NAMES[args.workload]),
            "value": fps, "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong" if fanout else "weak", "vs_baseline": None, "dtype": "f16",
            "data": "synthetic (seeded low-pass noise + pan, 8-bit YUV420; seeded random weights of the reference architecture)",
            "config": {"workload": ("%s " + res + " YUV420, ONE stream over all ranks (rank 0 codes, feature_p broadcast, reconstruction "
                                    "heads fanned out), " if fanout else "%s " + res + " YUV420 on 1xMI355X per rank, ") % NAMES[args.workload]
                                   + "q_index cycling {0,16,32,48,63}, skip_thres 0.15, one step = compress + decompress of %d "
                                     "picture(s)" % work.frames,
                       "sharding": "recon-head fan-out" if fanout else "independent streams (sharding.shard_range)",
                       "codec_objects": "one" if (args.workload == "intra" and args.one_codec) else "separate encoder / decoder",
                       "pictures_per_step": work.frames, "resolution": res},
            "encode_fps": work.frames / te, "decode_fps": work.frames / td,
            "avg_frame_encoding_time_ms": 1e3 * te / work.frames, "avg_frame_decoding_time_ms": 1e3 * td / work.frames,
            "fps_method": "value: K pipelined steps between device synchronisations; encode_fps / decode_fps: the reference's loop "
                          "(events around each call on a synchronised device, first %d calls dropped, rank 0)" % DROP_CALLS,
            "loop": "two-stage pipeline: an encoder thread codes step i + 1 while the decoder thread decodes step i (separate "
                    "encoder / decoder objects; DCVC_BENCH_SEQUENTIAL=1 = one call after the other)" if loop is not run_steps
                    else "one call after the other (one codec object codes and decodes)",
            "bytes_per_picture": nbytes / args.steps / work.frames,
            "bpp": 8.0 * nbytes / args.steps / work.frames / (height * width),
            "closure_ok": closure,
            "box": box_identity(device),
        }
        if sustained is not None:
            out["sustained"] = sustained
        if plain is not None:
            out["plain_loop"] = plain
        if not args.no_roofline and not fanout:
            out["roofline"] = roofline(work)
        if world == 1 and not args.no_extras:
            del work
            torch.cuda.empty_cache()
            others = {}
            for kind in NAMES:
                if kind == args.workload:
                    continue
                w = make_work(kind, height, width)
                others[kind] = fps_block(w, 0, 48 if kind in ("hts", "htl") else 96, 12, with_roofline=not args.no_roofline)
                del w
                torch.cuda.empty_cache()
            out["other_workloads"] = others
            if not args.no_uhd and (height, width) == (HEIGHT, WIDTH):
                uhd = {"resolution": "3840x2160"}
                for kind in NAMES:
                    w = make_work(kind, 2160, 3840, frames=2)
                    uhd[kind] = fps_block(w, 0, 6 if kind in ("hts", "htl") else 12, 3, with_roofline=not args.no_roofline)
                    del w
                    torch.cuda.empty_cache()
                out["uhd"] = uhd
        if world == 1 and not args.no_cpu_baseline and args.workload == "intra":
            out["cpu_baseline"] = cpu_baseline(cpu_net)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
