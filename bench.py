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
max over ranks) of the SEQUENTIAL loop: compress, then decompress, one call after the other from one thread, as the
reference harness orders its calls (one codec object for the intra model). --workload ld|hts|htl: configs[2] with the
inter models (one step = one compress + one decompress call on separate encoder / decoder objects: 1 picture for LD, a
chunk of 8 for HT).

Other modes:
  --sweep64     BASELINE configs[4]: the full 64-point rate sweep at 3840x2160, rate points sharded over the ranks
  --handoff K   ONE running GOP of an inter model whose temporal state moves to the next rank every K coded units
                (RCCL point-to-point: north_star's context exchange); time / bytes per hand-off, bit-exact continuation
  --fanout      ONE hierarchical stream, the 8 reconstruction heads of a chunk spread over the ranks

The JSON line (rank 0) carries, besides the driver contract:
  encode_fps / decode_fps   SURVEY 8d's metric, measured the reference's way (test_video.py:261-265,
                            321-325, 380-388; test_compress_time.py:60-69): device synchronised, events
                            around every compress / decompress call, the first 4 calls dropped,
                            pictures per call / mean call time. A separate pass after the timed region.
                            Also inside `config`.
  pipelined                 an additional throughput mode, never `value`: encoder and decoder objects (their own, created with
                            stream priorities) as a two-stage pipeline - an encoder thread codes picture i + 1 while the decoder
                            thread decodes picture i (N = 1 only)
  other_workloads           value / encode_fps / decode_fps / pipelined for the other three models (short runs; N = 1 only)
  roofline                  the DOMINANT contraction kernel (most time per step): its algorithmic FLOPs / the
                            HIP-event time of its launches, stamped live on the codec's stream by
                            hipExtLaunchKernelGGL in an extra eager pass; `all_contractions` = every contraction
                            launch together (> 99 % of the FLOPs), `kernels` = the per-kernel split; HBM
                            traffic from the committed PMC pass named in `traffic_source` (null when the
                            kernel sources changed since that pass)
  sustained                 the same loop run for >= --min-seconds after the K timed steps (the K-step region
                            of a short driver run is a fraction of a second)
  uhd                       short 3840x2160 runs of all four workloads (BASELINE configs[4]'s resolution) and `sweep64`: the
                            64-point rate sweep as a short run (LD, 1 I + 2 P pictures per rate point, closure per rate point)
  cpu_baseline              the reference's CPU-runnable path (fp32 graph forward_one_frame, restated in
                            oracle/torch_graph.py) on this host: all cores = `value`, one thread beside it
                            (the reference's set_torch_env pins 1, common.py:270), the bit-exact
                            oracle's compress + decompress, and `psnr_vs_source`: the codec's PSNR on the picture the graph
                            just reconstructed beside the graph's (north_star's 0.02 dB); N = 1 only, bounded samples
"""
import argparse
import contextlib
import ctypes
import json
import os
import sys
import time

# ONE hardware queue per stream-priority level (ROCclr's GPU_MAX_HW_QUEUES, default 4), set before the HIP runtime starts:
# every codec object brings two streams, HIP deals the streams of a priority level out over up to 4 hardware queues, and with
# the default what a loop reaches depended on which objects had been created in the process before - LD's sequential loop 182
# or 325 pictures/s, HT-S's 479 / 544 / 733, the intra pipeline 101 / 109 / 155 / 180 in the round-5 sessions, each value
# reproducible for its creation order. With one queue per level every order gave the best of those
# (profiles/r05_hw_queues.txt). INTEGRATION.md recommends the setting to every host of the plug-in; an explicit value in the
# environment wins. ONLY for a one-rank run: with RCCL in the process (--gpus N, --fanout, --handoff) its kernels would share the
# normal-priority hardware queue with the codecs' compute streams and the fan-out's broadcast / the hand-off's sends would lose
# the overlap they are timed on - never measured, so the N-rank children keep the runtime's default (advisor, round 5). Decided
# here, in front of the HIP runtime's start-up, from what the launcher exported; the setting in force is in the line's `runtime_env`.
if int(os.environ.get("WORLD_SIZE", "1") or "1") <= 1:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "1")

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
    p.add_argument("--sweep64", action="store_true",
                   help="BASELINE configs[4]: the full 64-point rate sweep (q_index = linspace(0, 63, 64), test_video.py:512-514) at "
                        "--resolution (default 3840x2160), rate points sharded over the ranks; --workload picks the model "
                        "(intra: I pictures only; ld / hts / htl: 1 I picture + --sweep-units coded units per rate point)")
    p.add_argument("--sweep-units", type=int, default=4, help="coded inter units per rate point of --sweep64")
    p.add_argument("--handoff", type=int, default=0, metavar="K",
                   help="ld / hts / htl: ONE running GOP whose temporal state moves to the next rank every K coded units "
                        "(RCCL point-to-point; north_star's context exchange). Reports pictures/s, time and bytes per hand-off "
                        "and whether the stream continued bit-exactly")
    p.add_argument("--no-pipeline", action="store_true",
                   help="skip the `pipelined` blocks (encoder and decoder objects as a two-stage pipeline on prioritised streams)")
    p.add_argument("--one-codec", action="store_true", help="(the default since round 5; kept for old command lines)")
    p.add_argument("--two-codecs", action="store_true", help="(kept for old command lines: the two-object loop is `pipelined`)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--no-extras", action="store_true", help="skip the short runs of the other three workloads")
    p.add_argument("--resolution", default=None, help="WxH of the synthetic pictures (default 1920x1080; 3840x2160 = configs[4], "
                                                      "the default of --sweep64)")
    p.add_argument("--no-uhd", action="store_true", help="skip the short 3840x2160 runs of the default line")
    p.add_argument("--no-resolutions", action="store_true",
                   help="skip the short 1280x720 / 832x480 runs of the default line (the reference's other tuned sizes)")
    p.add_argument("--min-seconds", type=float, default=2.0,
                   help="length of the `sustained` region behind the K timed steps (0 = none)")
    a = p.parse_args()
    if a.resolution is None:
        a.resolution = "3840x2160" if a.sweep64 else "%dx%d" % (WIDTH, HEIGHT)
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
    """configs[1]: every picture is an I picture. Default (`dec_net=None`): ONE codec object codes and decodes, one call
    after the other, as the reference harness uses its i_frame_net (test_video.py:187,226,338) - this is the loop `value` is
    measured on. `dec_net` + `prioritised`: a decoder object of its own with its compute stream at high priority, the second
    stage of run_steps_overlapped() - the decoder's four host round trips per picture (0.6 ms of idle GPU) are filled with the
    next picture's encoder kernels (reported as `pipelined`, never as `value`). The same two objects called one after the other
    from one thread are SLOWER than one object (125 against 143 pictures/s: the decoder's short kernels queue behind the
    encoder's reconstruction tail at equal priority)."""
    frames, kind = 1, "intra"

    def __init__(self, gpu_net, pics, pad_b, pad_r, dec_net=None, prioritised=False):
        self.net, self.pics, self.pad_b, self.pad_r = gpu_net, pics, pad_b, pad_r
        self.dec = dec_net if dec_net is not None else gpu_net
        self.height, self.width = int(pics[0].shape[2]), int(pics[0].shape[3])
        self.sps = {"height": self.height, "width": self.width}
        self.overlapped = dec_net is not None and prioritised
        if self.overlapped and os.environ.get("DCVC_BENCH_PRIORITIES", "1") != "0":
            # (the priority of a codec's streams is read when the native object is created: `gpu_net` must be an object that
            # has not coded anything yet - make_work() passes a fresh one)
            for obj, prio in ((self.net, "low"), (self.dec, "high")):
                os.environ["DCVC_COMPUTE_PRIORITY"] = prio
                obj._ensure_proxy()
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

    def __init__(self, kind, device, pics, gpu_intra, pad_b, pad_r, prioritised=False):
        from dcvc_amd import arch, models, synthetic
        self.kind, self.pad_b, self.pad_r = kind, pad_b, pad_r
        # prioritised: the objects of the two-stage pipeline (`pipelined` in the JSON line); `value` is measured on plain ones
        self.overlapped = prioritised
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
        if self.overlapped and os.environ.get("DCVC_BENCH_PRIORITIES", "1") != "0":
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

    # encoder and decoder are separate objects that share nothing but the bytes: with `prioritised` run_steps_overlapped()
    # drives them as a two-stage pipeline from two host threads (a real deployment runs them on different machines)

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
    def __init__(self, kind, device, pics, gpu_intra, pad_b, pad_r, dist):
        super().__init__(kind, device, pics, gpu_intra, pad_b, pad_r)      # (every rank takes part in every call: one thread)
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
    """run_steps (one call after the other: the reference harness's loop) unless the workload's objects were created for
    the two-stage pipeline (`prioritised`)"""
    return run_steps_overlapped if getattr(work, "overlapped", False) else run_steps


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
    # which sources: HEAD when this is a git checkout (the build container); on the GPU box the snapshot carries no .git,
    # so __graft_entry__.build() leaves HEAD in .git_head whenever it runs inside a checkout - and, independent of either,
    # a digest of the tracked sources that define the numbers (kernels, codecs, this file)
    try:
        import subprocess
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, timeout=10)
        if head.returncode == 0 and head.stdout.strip():
            out["commit"] = head.stdout.strip()
            dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--untracked-files=no"], capture_output=True,
                                   text=True, timeout=10)
            out["commit_dirty"] = bool(dirty.stdout.strip())
    except (OSError, subprocess.SubprocessError):
        pass
    if "commit" not in out:
        try:
            with open(os.path.join(ROOT, ".git_head")) as f:
                out["commit"] = f.read().strip()
                out["commit_from"] = ".git_head (written by __graft_entry__.build() in the last git checkout it ran in)"
        except OSError:
            pass
    out["source_digest"] = source_digest()
    return out


def source_digest():
    """sha256 over this file and every source under dcvc_amd/csrc + include (sorted by path): identifies the code that
    produced a line whether or not a git checkout is around"""
    import hashlib
    h = hashlib.sha256()
    paths = [os.path.join(ROOT, "bench.py")]
    for top in ("dcvc_amd/csrc", "include"):
        for d, _, files in os.walk(os.path.join(ROOT, top)):
            paths += [os.path.join(d, f) for f in files if f.endswith((".hip", ".h", ".cpp"))]
    for path in sorted(paths):
        h.update(os.path.relpath(path, ROOT).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def closure_ok(work, first):
    """After a timed region: one compress / decompress per rate point of QPS, checked (a throughput number from
    a desynchronised codec would otherwise look like any other). Continues the workload's stream at step `first`."""
    return all([work.closure(first + k, qp) for k, qp in enumerate(QPS)])


def timed_region(work, loop, first, steps):
    """`steps` steps of `loop` between two device synchronisations -> (seconds, coded bytes)"""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nbytes = loop(work, first, steps)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, nbytes


def pipelined_block(work, first, steps, warmup, min_seconds=0.0):
    """The two-stage pipeline (run_steps_overlapped) on objects created for it (`prioritised`): an additional throughput
    mode, reported beside `value`, never as `value`."""
    run_steps(work, first, warmup)           # (plain loop: graph capture of both objects from one thread)
    run_steps_overlapped(work, first + warmup, 2)        # the pipeline's own set-up outside the timed region
    first += warmup + 2
    dt, _ = timed_region(work, run_steps_overlapped, first, steps)
    first += steps
    out = {"value": steps * work.frames / dt, "unit": "frames/s", "steps": steps, "ms_per_step": 1e3 * dt / steps,
           "loop": "two-stage pipeline: an encoder thread codes step i + 1 while the decoder thread decodes step i (separate encoder / "
                   "decoder objects, the decoder's compute stream at high priority, the encoder's at low)"}
    if min_seconds > 0:
        more = min(100 * steps, max(steps, int(np.ceil(1.15 * min_seconds * steps / max(dt, 1e-6)))))
        t_more, _ = timed_region(work, run_steps_overlapped, first, more)
        first += more
        out["sustained"] = {"steps": more, "seconds": t_more, "value": more * work.frames / t_more, "unit": "frames/s"}
    out["closure_ok"] = closure_ok(work, first)
    return out


# Two phases (round 5). A = everything that is a THROUGHPUT number, with no timing event anywhere in the process yet; B = the
# passes that record timing events (the reference-style per-call rates: torch events around every call; the roofline pass:
# hipExtLaunchKernelGGL start / stop events per contraction launch). Measured in the round's sessions: once such a pass has
# run, later loops of the same process run slower - the intra pipeline 112 instead of 156 pictures/s, HT-L's plain loop 392
# instead of 529, HT-S's pipeline 531 instead of 649 (profiles/r05_phase_order.txt) - and the same loops as processes of their
# own do not. The HIP streams of all codec objects share a few hardware queues, and a queue that has carried timestamped
# dispatches keeps collecting timestamps (a completion signal per dispatch); nothing in the codec changes. So: all loops
# first, all event-stamped passes last, on the objects of phase A kept alive.
def fps_block_a(make, steps, warmup, with_pipeline=True):
    """Phase A of one workload as the default line's `other_workloads` / `uhd` report it: `value` = the plain loop
    (compress then decompress, one call after the other, unprioritised objects - what the reference harness does);
    `pipelined` = the two-stage loop on a second, prioritised set of objects. -> (report, plain objects, next step index)"""
    work = make(False)
    run_steps(work, 0, warmup)
    dt, nbytes = timed_region(work, run_steps, warmup, steps)
    out = {"value": steps * work.frames / dt, "unit": "frames/s", "steps": steps, "ms_per_step": 1e3 * dt / steps,
           "loop": "one call after the other (compress, then decompress)",
           "bpp": 8.0 * nbytes / steps / work.frames / (work.height * work.width)}
    if with_pipeline:
        wp = make(True)
        out["pipelined"] = pipelined_block(wp, 0, steps, warmup)
        del wp
        torch.cuda.empty_cache()
    return out, work, warmup + steps


def fps_block_b(out, work, first, with_roofline=False):
    """Phase B: the reference-style encode / decode rates, closure, roofline of the objects phase A measured"""
    ncalls = min(out["steps"], 24) + DROP_CALLS
    te, td = call_times(work, first, ncalls)
    out.update({"encode_fps": work.frames / te, "decode_fps": work.frames / td, "closure_ok": closure_ok(work, first + ncalls)})
    if with_roofline:
        r = roofline(work, n=2)
        out["roofline"] = {k: r[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "all_contractions")}
    return out


def cpu_baseline(cpu_net, device):
    """The reference's CPU path (fp32 graph) and the bit-exact oracle on this host, bounded samples - and, since the graph
    reconstructs a whole 1088x1920 picture anyway, the product's PSNR on that very picture beside the graph's."""
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
    # the picture: synthetic picture 32 at q 32 (the q-32 picture of tests/golden/graph_psnr_fullsize.json), replicate-padded
    # 1080 -> 1088 as the inference path pads it
    yy, uv = synthetic.synthetic_frame_yuv420(HEIGHT, WIDTH, 32, 0)
    x16 = synthetic.yuv420_to_x(yy, uv).half()
    xp = torch.nn.functional.pad(x16.float(), (0, 0, 0, -HEIGHT % 64), mode="replicate")
    kept = []
    t_all = torch_graph.time_forward(sd, xp.shape[2], xp.shape[3], 32, best, x=xp, keep=kept)
    t_one = torch_graph.time_forward(sd, 256, 256, 32, 1) * (1088 * 1920) / (256 * 256)

    def psnr(a, b):
        return float(10 * torch.log10(1.0 / torch.mean((a.double() - b.double()) ** 2)))
    src = x16[:, :, :HEIGHT, :WIDTH].float()
    graph_psnr = psnr(kept[0].clamp(-0.5, 0.5)[:, :, :HEIGHT, :WIDTH], src)
    # the product on the same picture, skip mode off (the graph has none; tests/golden/make_graph_psnr_golden.py says what
    # the skip mode does to RANDOM weights even at skip_thres 0)
    import copy
    m = copy.deepcopy(cpu_net)
    m.skip_thres = -60000.0
    g = _to_gpu(m, device)
    pad_r, pad_b = g.get_padding_size(HEIGHT, WIDTH, 16)
    r = g.compress(x16.to(device).contiguous(memory_format=torch.channels_last), 32, pad_b, pad_r)
    torch.cuda.synchronize()
    ours_psnr = psnr(r["x_hat"][:, :, :HEIGHT, :WIDTH].float().cpu(), src)
    del g
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
        "psnr_vs_source": {"fp32_graph": graph_psnr, "this_codec": ours_psnr, "delta_psnr_vs_fp32_graph": ours_psnr - graph_psnr,
                           "tolerance": 0.02, "within_tolerance": abs(ours_psnr - graph_psnr) <= 0.02,
                           "picture": "synthetic 1920x1080 picture 32 at q 32, skip mode off on the codec (the graph has none), "
                                      "PSNR of the 4:4:4 working planes over the visible area; the reference's own graph on the "
                                      "same picture: tests/golden/graph_psnr_fullsize.json"},
        "one_thread": {"value": 1.0 / t_one, "unit": "frames/s", "cores": 1,
                       "sample": "the same graph on 1 thread (the reference harness pins 1, common.py:270): one 256x256 crop, "
                                 "scaled by area to 1080p (%.1f s per picture)" % t_one},
        "oracle": {"value": 1.0 / t_orc, "unit": "frames/s", "cores": cores, "kind": "port",
                   "sample": "bit-exact oracle (oracle/codec.py, matrix-core arithmetic emulated in integer SIMD) compress + decompress "
                             "of one %dx%d crop scaled by area to 1080p (%.0f s per picture)" % (h, w, t_orc)},
    }


TRAFFIC_FILE = os.path.join("profiles", "r06_hbm_traffic.json")
KERNEL_SOURCES = ("dcb_nsplit8_kernel.h", "dcb_nsplit_common.h", "dcb_nsplit.hip", "dcb_pair8_kernel.h", "dcb_pair.hip", "conv_gemm.hip", "dcb_tail.hip", "ffn_fused.hip", "dwconv.hip",
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


KERNEL_NAMES = {0: "conv_gemm_kernel", 2: "dcb_tail_kernel", 3: "ffn_fused_kernel", 4: "dcb_nsplit_kernel",
                5: "dcb_nsplit8_kernel", 6: "dcb_pair8_kernel"}


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
    # variant bits 28..31: the kernel family of the launch (ops.h GemmLaunchInfo)
    fam = (buf["variant"].astype(np.int64) >> 28) & 0xF
    family = fam
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
    kernels = []
    for f, name in KERNEL_NAMES.items():
        sel = family == f
        if f in (4, 5):
            # one entry per shape of the N-split block kernel: <C, CI, pixels per workgroup> follows from (N, K, M): K = 7 CI
            # with the next block's dc.0 inside the launch, 6 CI without (dcb_nsplit_kernel.h launch())
            var = buf["variant"].astype(np.int64)
            if f == 5:       # the 8-wave kernel records its inner width, its NEXT slot (0, 1 = next dc.0, NN = closing conv) and, in bit 24, whether the depthwise conv ran inside
                ci_of, slot = var & 0xFFF, (var >> 12) & 0xFFF
            else:
                inner = {k * ci: ci for ci in (128, 256, 384, 512, 768) for k in (6, 7)}
                ci_of = np.array([inner.get(int(c), 0) for c in buf["K"]], dtype=np.int64)
                slot = (buf["K"] == 7 * ci_of).astype(np.int64)
            width = np.where(slot == 1, ci_of, slot)                     # channels the NEXT slot writes
            shapes = sorted({(int(a), int(b), int(c)) for a, b, c in zip(buf["M"][sel], buf["N"][sel], ci_of[sel])})
            for (m, c, ci) in shapes:
                one = sel & (buf["M"] == m) & (buf["N"] == c) & (ci_of == ci)
                nxt = float((slot[one] == 1).mean())                     # share of the launches with the next dc.0
                fin = float((slot[one] > 1).mean())                      # ... with the chain's closing conv
                wn = float(width[one].mean())                            # mean width of the NEXT slot's output
                wide = m >= 64 * 200 and c < 768
                # every operand once: t2 [M][CI], x [M][C] in, y [M][C] (and the NEXT slot's output) out, the weights
                alg = 2 * m * (2 * c + ci + wn) + 2 * c * (6 * ci + wn)
                k = part(one, "%s<%d, %d, %d px>" % (name, c, ci, 64 if wide else 32), alg)
                if k:
                    k["pixels"] = m
                    k["with_next_dc0"] = nxt
                    k["with_closing_conv"] = fin
                    if f == 5:
                        k["with_depthwise_inside"] = float(((var[one] >> 24) & 1).mean())      # round 6: the block's depthwise conv inside the launch
                    kernels.append(k)
            continue
        k = part(sel, name, None)
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


class BenchEnv:
    """what every mode of main() needs"""

    def __init__(self, args, rank, world, device, dist, comm_device, cpu_net, gpu_net, sync, max_over_ranks):
        self.args, self.rank, self.world, self.device, self.dist, self.comm_device = args, rank, world, device, dist, comm_device
        self.cpu_net, self.gpu_net, self.sync, self.max_over_ranks = cpu_net, gpu_net, sync, max_over_ranks


def common_fields(env, fps, elapsed, steps, scaling, metric, config):
    args = env.args
    return {"metric": metric, "value": fps, "unit": "frames/s", "n_gpus": env.world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / max(steps, 1), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f16",
            "data": "synthetic (seeded low-pass noise + pan, 8-bit YUV420; seeded random weights of the reference architecture)",
            "config": config, "box": box_identity(env.device),
            "runtime_env": {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"),
                            "plugin_policy": getattr(sys.modules.get("inference_extensions_cuda"), "hw_queue_policy", None)}}


def run_default(env, make_work):
    """configs[1] / configs[2]: independent streams, one per rank (or --fanout: one hierarchical stream over all ranks)."""
    args, rank, world, dist, device = env.args, env.rank, env.world, env.dist, env.device
    height, width = args.height, args.width
    fanout = args.fanout and world > 1
    if args.fanout and args.workload not in ("hts", "htl"):
        raise SystemExit("--fanout needs --workload hts or htl (the models with 8 reconstruction heads per call)")
    if fanout:
        pics = make_pictures(args.frames, rank, device, height, width)
        pad_r, pad_b = env.gpu_net.get_padding_size(height, width, 16)
        work = FanoutWorkload(args.workload, device, pics, env.gpu_net, pad_b, pad_r, dist)
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

    # `value`: compress then decompress, one call after the other from one thread, no host synchronisation between steps -
    # the reference harness's order of calls (test_video.py:224-265, 300-325) without its per-call synchronisation
    run_steps(work, 0, args.warmup)
    env.sync()
    t0 = time.perf_counter()
    nbytes = run_steps(work, args.warmup + mine.start, args.steps)
    env.sync()
    elapsed = env.max_over_ranks(time.perf_counter() - t0)

    # the same loop for >= --min-seconds: the K-step region of a short driver run lasts a fraction of a second
    sustained = None
    if args.min_seconds > 0:
        more = max(args.steps, int(np.ceil(1.15 * args.min_seconds * args.steps / max(elapsed, 1e-6))))      # (a margin: warm steps run faster)
        more = min(more, 100 * args.steps)
        env.sync()
        t0 = time.perf_counter()
        run_steps(work, args.warmup + mine.start + args.steps, more)
        env.sync()
        t_more = env.max_over_ranks(time.perf_counter() - t0)
        sustained = {"steps": more, "seconds": t_more, "value": (1 if fanout else world) * more * work.frames / t_more,
                     "unit": "frames/s"}

    # step index behind everything this rank has coded so far (the inter workloads' GOP / reset cadence follows the index)
    cursor = args.warmup + mine.start + args.steps + (sustained or {}).get("steps", 0)
    # every rank checks ITS codec objects after the timed regions (fan-out: the shared stream is checked by the tests)
    closure = None if fanout else closure_ok(work, cursor)
    cursor += len(QPS)
    if dist is not None and closure is not None:
        flag = torch.tensor([1.0 if closure else 0.0], dtype=torch.float64, device=env.comm_device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        closure = bool(flag.item() > 0.5)
    ncalls = min(args.steps, 32) + DROP_CALLS
    if fanout:       # every rank takes part in the per-call timing loop (the broadcast is a collective)
        te, td = call_times(work, cursor, ncalls)
    if rank != 0:
        return None
    fps = (1 if fanout else world) * args.steps * work.frames / elapsed
    res = "%dx%d" % (width, height)
    one_object = args.workload == "intra"
    loop_text = ("one call after the other from one thread: compress, then decompress (%s), no host synchronisation between steps"
                 % ("ONE codec object codes and decodes, as the reference harness uses its i_frame_net" if one_object
                    else "separate encoder / decoder objects, the decoder sees only the bytes"))
    out = common_fields(
        env, fps, elapsed, args.steps, "strong" if fanout else "weak",
        "%s YUV420 %s encode+decode pictures per second (%s, real rANS bit streams, q_index in {0,16,32,48,63}; sequential loop: "
        "compress then decompress per picture)" % ("1080p" if (height, width) == (1080, 1920) else res,
                                                   "intra" if args.workload == "intra" else "inter", NAMES[args.workload]),
        {"workload": ("%s " + res + " YUV420, ONE stream over all ranks (rank 0 codes, feature_p broadcast, reconstruction "
                      "heads fanned out), " if fanout else "%s " + res + " YUV420 on 1xMI355X per rank, ") % NAMES[args.workload]
                     + "q_index cycling {0,16,32,48,63}, skip_thres 0.15, one step = compress + decompress of %d "
                       "picture(s)" % work.frames,
         "sharding": "recon-head fan-out" if fanout else "independent streams (sharding.shard_range)",
         "codec_objects": "one" if one_object else "separate encoder / decoder",
         "loop": "sequential",
         "pictures_per_step": work.frames, "resolution": res})
    out.update({
        "fps_method": "value: K steps of the sequential loop between device synchronisations; encode_fps / decode_fps: the "
                      "reference's loop (events around each call on a synchronised device, first %d calls dropped, rank 0), "
                      "measured behind every throughput loop of the process (see `measurement_order`)" % DROP_CALLS,
        "measurement_order": "phase A1: the plain loops of every workload (value, sustained, other_workloads, uhd, sweep64); phase A2: "
                             "the two-stage pipelines on objects of their own (priority streams); phase B: the per-call event loops "
                             "(encode_fps / decode_fps) and the event-stamped roofline passes on the objects of A1 - no timing event "
                             "and no priority stream exists in the process while a plain loop is measured",
        "loop": loop_text,
        "bytes_per_picture": nbytes / args.steps / work.frames,
        "bpp": 8.0 * nbytes / args.steps / work.frames / (height * width),
        "closure_ok": closure,
    })
    if sustained is not None:
        out["sustained"] = sustained
    extras = world == 1 and not args.no_extras
    # ---------------------------------------------------------------- phase A1, continued: the plain loops of every workload
    kept = []          # (report, plain objects, next step index) of phase A1, for phase B
    pipelines = []     # (report, factory, steps, warmup) of the two-stage pipelines, measured in phase A2
    if extras:
        others = {}
        for kind in NAMES:
            if kind == args.workload:
                continue
            steps = 48 if kind in ("hts", "htl") else 96
            make = (lambda pr, kind=kind: make_work(kind, height, width, prioritised=pr))
            o, w, nxt = fps_block_a(make, steps, 12, with_pipeline=False)
            others[kind] = o
            kept.append((o, w, nxt))
            pipelines.append((o, make, steps, 12))
        out["other_workloads"] = others
        if not args.no_uhd and (height, width) == (HEIGHT, WIDTH):
            uhd = {"resolution": "3840x2160"}
            for kind in NAMES:
                o, w, nxt = fps_block_a(lambda pr, kind=kind: make_work(kind, 2160, 3840, frames=2, prioritised=pr),
                                        6 if kind in ("hts", "htl") else 12, 3, with_pipeline=False)
                uhd[kind] = o
                kept.append((o, w, nxt))
            # BASELINE configs[4]: the 64-point rate sweep at 3840x2160, as a short run (1 I + 2 P pictures per rate point)
            uhd["sweep64"] = sweep64_block(env, "ld", 2160, 3840, 2)
            out["uhd"] = uhd
        if not args.no_resolutions and (height, width) == (HEIGHT, WIDTH):
            # the reference's other tuned picture sizes (README.md:202, cutlass/cutlass_kernel.h:229-249: per-shape tile tables for
            # 1280x720, 832x480, 416x240 too); BASELINE.md holds an A100 row for 720p (encode / decode fps, reference-style)
            res_block = {"reference_a100_720p": {"ld": [525.0, 501.2], "hts": [1098.4, 786.1], "htl": [648.2, 459.4],
                                                 "source": "assets/complexity.png (b), BASELINE.md section 1: encode / decode fps"}}
            for rh, rw in ((720, 1280), (480, 832)):
                blk = {}
                for kind in NAMES:
                    o, w, nxt = fps_block_a(lambda pr, kind=kind, rh=rh, rw=rw: make_work(kind, rh, rw, frames=3, prioritised=pr),
                                            24 if kind in ("hts", "htl") else 48, 6, with_pipeline=False)
                    blk[kind] = o
                    kept.append((o, w, nxt))
                res_block["%dx%d" % (rw, rh)] = blk
            out["resolutions"] = res_block
    # ---------------------------------------------------------------- phase A2: the two-stage pipelines (objects of their own,
    # streams at low / high priority) - behind every plain loop: what a plain loop on separate encoder / decoder objects reaches
    # depends on which streams have been created in the process before (profiles/r05_pipeline_order.txt: HIP multiplexes the
    # streams of a priority level onto a few hardware queues), and streams of a NEW priority level in front of the plain LD /
    # HT-S loops halved them in one session (r05 closing session, first attempt: LD 182 instead of 325 pictures/s)
    if world == 1 and not args.no_pipeline:
        wp = make_work(args.workload, height, width, prioritised=True)
        out["pipelined"] = pipelined_block(wp, 0, args.steps, args.warmup, args.min_seconds)
        del wp
        torch.cuda.empty_cache()
        for o, make, steps, warm in pipelines:
            wp = make(True)
            o["pipelined"] = pipelined_block(wp, 0, steps, warm)
            del wp
            torch.cuda.empty_cache()
    # ---------------------------------------------------------------- phase B: the event-stamped passes
    if not fanout:
        te, td = call_times(work, cursor, ncalls)
    out.update({"encode_fps": work.frames / te, "decode_fps": work.frames / td,
                "avg_frame_encoding_time_ms": 1e3 * te / work.frames, "avg_frame_decoding_time_ms": 1e3 * td / work.frames})
    # the reference's own metric (BASELINE.json; test_video.py:261-265, 380-388) inside `config` too, so that it survives any
    # filtering of top-level keys
    out["config"].update({"encode_fps": out["encode_fps"], "decode_fps": out["decode_fps"]})
    if not args.no_roofline and not fanout:
        out["roofline"] = roofline(work)
    del work
    torch.cuda.empty_cache()
    while kept:
        o, w, nxt = kept.pop(0)
        fps_block_b(o, w, nxt, with_roofline=not args.no_roofline)
        del w
        torch.cuda.empty_cache()
    if world == 1 and not args.no_cpu_baseline and args.workload == "intra":
        out["cpu_baseline"] = cpu_baseline(env.cpu_net, device)
    other = compact_other(out)
    if other:
        out["config"]["other"] = other
        out["config"]["other_fields"] = "[pictures/s of the sequential loop, encode_fps, decode_fps] per workload"
    return out


def compact_other(out):
    """`config.other`: [pictures/s of the sequential loop, encode_fps, decode_fps] of every workload measured beside the headline
    one, keyed "ld" / "hts" / "htl" / "intra" at the default resolution and "uhd_<kind>" at 3840x2160 - a record of the other
    models' numbers that survives a reader that keeps only the contract keys of the line."""
    def row(o):
        return [round(float(o[k]), 1) if o.get(k) is not None else None for k in ("value", "encode_fps", "decode_fps")]
    other = {}
    for kind, o in (out.get("other_workloads") or {}).items():
        other[kind] = row(o)
    for kind, o in (out.get("uhd") or {}).items():
        if isinstance(o, dict) and "value" in o and kind in NAMES:
            other["uhd_" + kind] = row(o)
    for res, block in (out.get("resolutions") or {}).items():
        if not res[0].isdigit():
            continue
        for kind, o in block.items():
            if isinstance(o, dict) and "value" in o:
                other[res + "_" + kind] = row(o)
    return other


def ordered_line(out):
    """The ONE JSON line with its long report blocks FIRST and the driver contract's keys LAST: a reader that keeps only the tail
    of stdout (round 5: an 8 KB tail that began inside the `uhd` block) still sees metric / value / config / roofline."""
    long_blocks = ("measurement_order", "fps_method", "loop", "data", "box", "uhd", "other_workloads", "resolutions", "pipelined",
                   "sustained")
    first = {k: out[k] for k in long_blocks if k in out}
    rest = {k: v for k, v in out.items() if k not in first}
    for k in ("cpu_baseline", "roofline", "config"):      # the structured contract blocks at the very end
        if k in rest:
            rest[k] = rest.pop(k)
    first.update(rest)
    return first


def sweep64_block(env, kind, height, width, units):
    """BASELINE configs[4] / test_video.py:512-514: q_index = linspace(0, 63, 64) - every one of the 64 rate points - at
    `height` x `width`, sharded over the ranks by rate point (sharding.shard_range: the reference's own scaling mode is one
    worker per (sequence, rate point), test_video.py:527-564). Per rate point: one I picture (intra codec, compress +
    decompress) and, for an inter `kind`, `units` coded units of the inter model seeded from it (LD: pictures, HT: chunks of
    8), each compress + decompress on separate encoder / decoder objects. closure is checked PER RATE POINT outside the
    timed sections: the decoder's I picture equals the encoder's bit for bit, the inter decoder (bytes only) holds the
    encoder's feature_p. Returns rank 0's report (None elsewhere)."""
    from dcvc_amd import sharding
    rank, world, dist, device = env.rank, env.world, env.dist, env.device
    pics = make_pictures(2, 0, device, height, width)
    pad_r, pad_b = env.gpu_net.get_padding_size(height, width, 16)
    i_enc, i_dec = env.gpu_net, _to_gpu(env.cpu_net, device)
    sps = {"height": height, "width": width}
    inter = None if kind == "intra" else InterWorkload(kind, device, pics, env.gpu_net, pad_b, pad_r)
    frames = 1 if inter is None else inter.frames
    qps = [int(q) for q in np.linspace(0, 63, 64)]
    assert qps == list(range(64))
    mine = [qps[i] for i in sharding.shard_range(len(qps), rank, world)]

    def rate_point(q):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = i_enc.compress(pics[0], q, pad_b, pad_r)
        xi = r["x_hat"]
        xd = i_dec.decompress(r["bit_stream"], sps, q, r["ec_parallel"])["x_hat"]
        nbytes = [len(r["bit_stream"])]
        if inter is not None:
            inter.enc.add_ref_feature_from_frame(xi)
            inter.dec.add_ref_feature_from_frame(xd, apply_feature_adaptor=False)
            for u in range(units):
                e = inter.enc.compress(inter.inputs[u % len(inter.inputs)], q, 0, pad_b, pad_r)
                inter.dec.decompress(e["bit_stream"], sps, q, e["ec_parallel"], 0)
                nbytes.append(len(e["bit_stream"]))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ok = bool(torch.equal(xi, xd)) and bool(torch.isfinite(xd.float()).all())
        if inter is not None:
            ok = ok and bool(np.array_equal(inter.enc._ensure_proxy().debug_read("feature_p", np.float16),
                                            inter.dec._ensure_proxy().debug_read("feature_p", np.float16)))
        return dt, nbytes, ok

    for q in mine[:2]:            # untimed: graph capture of every stage
        rate_point(q)
    env.sync()
    per_q, total = [], 0.0
    for q in mine:
        dt, nbytes, ok = rate_point(q)
        total += dt
        per_q.append({"q": q, "bytes_i": nbytes[0], "bytes_p": sum(nbytes[1:]), "closure_ok": ok})
    elapsed = env.max_over_ranks(total)
    if dist is not None:
        gathered = [None] * world
        dist.all_gather_object(gathered, per_q)
        per_q = [e for part in gathered for e in part]
    del inter, i_dec
    torch.cuda.empty_cache()
    if rank != 0:
        return None
    per_q.sort(key=lambda e: e["q"])
    pictures = len(qps) * (1 + (0 if kind == "intra" else units * frames))
    bpp = [8.0 * (e["bytes_i"] + e["bytes_p"]) / (1 + (0 if kind == "intra" else units * frames)) / (height * width) for e in per_q]
    return {"value": pictures / elapsed, "unit": "frames/s", "seconds": elapsed, "rate_points": len(per_q),
            "pictures_per_rate_point": pictures // len(qps), "resolution": "%dx%d" % (width, height), "workload": kind,
            "closure_ok": all(e["closure_ok"] for e in per_q) and [e["q"] for e in per_q] == qps,
            "closure_ok_per_q": [e["closure_ok"] for e in per_q],
            "bpp_per_q": [round(b, 4) for b in bpp],
            "note": "q_index = linspace(0, 63, 64) (test_video.py:512-514); per rate point 1 I picture + %s, compress + "
                    "decompress; rate points sharded over the ranks (sharding.shard_range); timed per rate point between device "
                    "synchronisations, closure checked outside the timed sections"
                    % ("nothing else" if kind == "intra" else "%d coded unit(s) of %d picture(s) of the %s model" % (units, frames, kind))}


def run_sweep64(env, kind, units):
    args = env.args
    blk = sweep64_block(env, kind, args.height, args.width, units)
    if env.rank != 0:
        return None
    res = "%dx%d" % (args.width, args.height)
    out = common_fields(
        env, blk["value"], blk["seconds"], blk["rate_points"], "strong",
        "%s YUV420 full 64-point rate sweep, encode+decode pictures per second (%s, real rANS bit streams)" % (res, NAMES[kind]),
        {"workload": "BASELINE configs[4]: %s %s YUV420, q_index = linspace(0, 63, 64), skip_thres 0.15, %d picture(s) per rate point, "
                     "rate points sharded over %d rank(s)" % (NAMES[kind], res, blk["pictures_per_rate_point"], env.world),
         "sharding": "rate points (sharding.shard_range)", "resolution": res, "loop": "sequential"})
    out["steps_are"] = "rate points"
    out["sweep64"] = blk
    out["closure_ok"] = blk["closure_ok"]
    return out


def run_handoff(env, kind, every):
    """north_star's "temporal context exchanged by RCCL point-to-point": ONE running GOP of an inter model whose coding moves
    to the next rank every `every` coded units. The owner of unit i is rank (i // every) % world; at a boundary the owner
    exports the temporal state of its encoder AND decoder objects (reference feature, memory, feature_p, context, temporal
    prior: one flat tensor each) and sends both point-to-point (sharding.send_state / recv_state: RCCL over xGMI; host-staged
    over gloo). Inside a GOP the recurrence is strictly sequential, so this buys placement (load balancing,
    pre-emption), not speed: the line reports pictures/s of the ONE stream, the time and size of a hand-off, and - the point -
    that the bytes of every coded unit equal those of the same stream coded on one rank without any hand-off."""
    from dcvc_amd import sharding
    args, rank, world, dist, device = env.args, env.rank, env.world, env.dist, env.device
    if kind == "intra":
        raise SystemExit("--handoff needs --workload ld, hts or htl (the intra codec has no temporal state)")
    height, width = args.height, args.width
    pics = make_pictures(args.frames, 0, device, height, width)          # every rank: the SAME pictures (one stream)
    pad_r, pad_b = env.gpu_net.get_padding_size(height, width, 16)
    work = InterWorkload(kind, device, pics, env.gpu_net, pad_b, pad_r)
    total = args.warmup + args.steps
    owner = lambda i: (i // every) % world
    units, t_send, n_bytes = [], [], 0

    def code(i):
        qp = QPS[i % len(QPS)]
        if i % work.gop == 0:
            work.prepare(i)
        enc = work.compress(i, qp)
        work.decompress(i, qp, enc)
        units.append((i, enc["bit_stream"]))

    def span(first, last):
        nonlocal n_bytes
        for i in range(first, last):
            if owner(i) == rank:
                code(i)
            if i + 1 < total and owner(i + 1) != owner(i) and world > 1:
                if owner(i) == rank:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    n = sharding.send_state(work.enc._ensure_proxy(), owner(i + 1), dist)
                    n += sharding.send_state(work.dec._ensure_proxy(), owner(i + 1), dist)
                    torch.cuda.synchronize()
                    if i >= args.warmup:
                        t_send.append(time.perf_counter() - t0)
                    n_bytes = n
                elif owner(i + 1) == rank:
                    sharding.recv_state(work.enc._ensure_proxy(), owner(i), height, width, dist)
                    sharding.recv_state(work.dec._ensure_proxy(), owner(i), height, width, dist)

    span(0, args.warmup)
    env.sync()
    t0 = time.perf_counter()
    span(args.warmup, total)
    env.sync()
    elapsed = env.max_over_ranks(time.perf_counter() - t0)
    coded = sharding.gather_units(units, dist)
    stats = torch.tensor([sum(t_send), len(t_send), n_bytes], dtype=torch.float64, device=env.comm_device)
    if dist is not None:
        parts = [torch.zeros_like(stats) for _ in range(world)]
        dist.all_gather(parts, stats)
        stats = torch.stack(parts).cpu()
    else:
        stats = stats.reshape(1, 3).cpu()
    if rank != 0:
        return None
    # the same stream on ONE rank, fresh objects, no hand-off
    ref = InterWorkload(kind, device, pics, env.gpu_net, pad_b, pad_r)
    want = []
    for i in range(total):
        qp = QPS[i % len(QPS)]
        ref.prepare(i)
        enc = ref.compress(i, qp)
        ref.decompress(i, qp, enc)
        want.append(enc["bit_stream"])
    torch.cuda.synchronize()
    exact = len(coded) == total and all(bytes(a) == bytes(b) for a, b in zip(coded, want))
    n_handoffs = int(stats[:, 1].sum().item())
    res = "%dx%d" % (width, height)
    out = common_fields(
        env, args.steps * work.frames / elapsed, elapsed, args.steps, "strong",
        "%s YUV420 inter encode+decode pictures per second of ONE running GOP handed from GPU to GPU every %d coded unit(s) (%s, "
        "real rANS bit streams)" % ("1080p" if (height, width) == (1080, 1920) else res, every, NAMES[kind]),
        {"workload": "%s %s YUV420, ONE stream over %d rank(s): the temporal state of encoder and decoder moves to the next rank every "
                     "%d coded unit(s) of %d picture(s) (point-to-point, sharding.send_state / recv_state), q_index cycling "
                     "{0,16,32,48,63}, skip_thres 0.15" % (NAMES[kind], res, world, every, work.frames),
         "sharding": "GOP hand-off (temporal state point-to-point)", "resolution": res, "loop": "sequential",
         "pictures_per_step": work.frames})
    out["handoff"] = {
        "every_units": every, "count": n_handoffs,
        "us_per_handoff": 1e6 * float(stats[:, 0].sum().item()) / n_handoffs if n_handoffs else None,
        "bytes_per_handoff": int(stats[:, 2].max().item()),
        "gb_per_s": (float(stats[:, 2].max().item()) * n_handoffs / float(stats[:, 0].sum().item()) / 1e9) if n_handoffs else None,
        "timing": "sender side: synchronise, export + send encoder state, export + send decoder state, synchronise",
        "backend": dist.get_backend() if dist is not None else None,
        "bit_exact_continuation": exact,
        "checked": "the bytes of all %d coded units (warm-up included) against the same stream coded on rank 0 alone" % total}
    out["closure_ok"] = exact
    return out


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

    def make_work(kind, h, w, frames=args.frames, prioritised=False):
        pics = make_pictures(frames, rank, device, h, w)
        pad_r, pad_b = gpu_net.get_padding_size(h, w, 16)
        if kind == "intra":
            if prioritised:      # encoder and decoder objects of their own (low / high priority streams)
                return IntraWorkload(_to_gpu(cpu_net, device), pics, pad_b, pad_r, _to_gpu(cpu_net, device), True)
            return IntraWorkload(gpu_net, pics, pad_b, pad_r)
        return InterWorkload(kind, device, pics, gpu_net, pad_b, pad_r, prioritised)

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

    env = BenchEnv(args, rank, world, device, dist, comm_device, cpu_net, gpu_net, sync, max_over_ranks)
    if args.sweep64:
        out = run_sweep64(env, args.workload, args.sweep_units)
    elif args.handoff > 0:
        out = run_handoff(env, args.workload, args.handoff)
    else:
        out = run_default(env, make_work)
    if rank == 0 and out is not None:
        print(json.dumps(ordered_line(out)), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
