"""Multi-GPU sharding of a coding job: one process per GPU, independent units per rank, no
data-path collective.

What is independent in DCVC-UF:
  * intra coding: every picture;
  * inter coding: every intra period (a GOP starts from an I picture and re-seeds the temporal
    state with add_ref_feature_from_frame - test_video.py:206-231), and inside a GOP nothing: picture
    t needs the feature memory of picture t-1, so a GOP stays on one GPU;
  * rate sweeps: every qp (the reference's own scaling mode, test_video.py:530-560 worker pool).
The only communication is the control plane: rank 0 collects the coded units (bytes) in display
order. With backend "nccl" (= RCCL) the byte payloads travel as uint8 device tensors over xGMI,
with "gloo" (CPU tests) as host tensors.
"""
import numpy as np
import torch


def shard_range(n_units, rank, world):
    """Contiguous balanced split: the first (n_units % world) ranks get one extra unit."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    base, extra = divmod(n_units, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def plan_gops(frame_num, intra_period, frame_delay=1):
    """Independent units of an inter-coded sequence as (first_frame, n_frames), following the
    I-picture placement of test_video.py:204-213: frame 0 is intra; with intra_period > 1 every
    frame with index % intra_period == 1 (other than frame 1) starts a new GOP; intra_period
    <= 0 means one GOP; intra_period == 1 means all-intra."""
    if frame_num <= 0:
        return []
    if intra_period == 1:
        return [(i, 1) for i in range(frame_num)]
    if intra_period > 1 and intra_period % frame_delay != 0:
        raise ValueError("intra_period must be a multiple of the model's frame delay")
    starts = [0]
    if intra_period > 1:
        starts += [i for i in range(2, frame_num) if i % intra_period == 1]
    return [(s, e - s) for s, e in zip(starts, starts[1:] + [frame_num])]


def _device_for(dist):
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def gather_units(local_units, dist=None, dst=0):
    """local_units: list of (unit_index, bytes) coded by this rank. Returns on rank `dst` the list
    of bytes of ALL ranks ordered by unit_index (None elsewhere). Two collectives: one all_gather
    of the (index, length) table, one all_gather of the padded payloads."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [b for _, b in sorted(local_units, key=lambda u: u[0])]
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = _device_for(dist)
    count = torch.tensor([len(local_units)], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(count) for _ in range(world)]
    dist.all_gather(counts, count)
    max_units = max(int(c.item()) for c in counts)
    table = torch.full((max(max_units, 1), 2), -1, dtype=torch.int64, device=dev)
    for i, (idx, b) in enumerate(local_units):
        table[i, 0], table[i, 1] = idx, len(b)
    tables = [torch.zeros_like(table) for _ in range(world)]
    dist.all_gather(tables, table)
    totals = [int(t[:, 1].clamp(min=0).sum().item()) for t in tables]
    cap = max(max(totals), 1)
    payload = torch.zeros(cap, dtype=torch.uint8, device=dev)
    mine = b"".join(b for _, b in local_units)
    if mine:
        payload[:len(mine)] = torch.from_numpy(np.frombuffer(mine, dtype=np.uint8).copy()).to(dev)
    payloads = [torch.zeros_like(payload) for _ in range(world)]
    dist.all_gather(payloads, payload)
    if rank != dst:
        return None
    units = []
    for t, p in zip(tables, payloads):
        host, off = p.cpu().numpy(), 0
        for idx, n in t.cpu().tolist():
            if idx < 0:
                continue
            units.append((idx, host[off:off + n].tobytes()))
            off += n
    return [b for _, b in sorted(units, key=lambda u: u[0])]


def code_sharded(n_units, code_unit, dist=None):
    """Runs code_unit(unit_index) -> bytes for this rank's share and gathers everything on rank 0."""
    rank = dist.get_rank() if dist is not None and dist.is_initialized() else 0
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    local = [(i, code_unit(i)) for i in shard_range(n_units, rank, world)]
    return gather_units(local, dist)


# ------------------------------------------------------------------ reconstruction-head fan-out (HT models)
# The one place where a chunk of 8 pictures splits one-picture-per-GPU (SURVEY 8e iii): the 8 heads
# of the hierarchical models depend only on feature_p (video_model_ht.py:252-275; 36 % of HT-S's
# decode MACs, 22 % of HT-L's). Rank `src` owns the stream: it decodes (entropy decoding, priors,
# decoder - everything that defines the bytes stays on one GPU), broadcasts feature_p (33.4 MB at
# 1080p: one RCCL broadcast, on xGMI 7 point-to-point links), every rank runs its heads.
def head_owner(picture, world):
    """rank that reconstructs `picture` (0..7) of a chunk. Picture 7 stays with rank 0 (its head
    output is the reset feature of the temporal state); with 2 or 4 ranks whole picture pairs stay
    together (HT-S pairs share a trunk block)."""
    if world <= 1:
        return 0
    if world in (2, 4):
        return (3 - picture // 2) % world
    return (7 - picture) % world


def head_mask(rank, world):
    return sum(1 << i for i in range(8) if head_owner(i, world) == rank)


def _has_gpu():
    return torch.cuda.is_available() and torch.cuda.device_count() > 0


def _to_wire(t, dist):
    """a tensor as the backend can carry it: device tensors over RCCL, host tensors over gloo (which has no
    point-to-point for device memory) - the latter also on a GPU box (bench.py's gloo mode, tests)"""
    return t if dist.get_backend() == "nccl" or not t.is_cuda else t.cpu()


def decompress_fanout(proxy, bit_stream, qp, height, width, ec_parallel, reset, dist, src=0):
    """One chunk of an HT stream decoded over all ranks of `dist`. `bit_stream` etc. are needed on
    rank `src` only. Returns this rank's {picture index: x_hat tensor}.

    Rank `src` decodes WITHOUT its reconstruction heads (recon mask 0: entropy decoding, priors, decoder),
    exports feature_p, starts the broadcast asynchronously (RCCL runs it on its own stream) and only then
    runs its own heads - so the transfer and the other ranks' heads overlap the owner's heads instead of
    waiting for them. The recon mask of the owner's proxy STAYS 0 afterwards (switching it re-captures the
    decode graphs, so a stream of fan-out calls must not flip it per chunk): a plain decompress() of that proxy
    then returns None for every picture until restore_heads(proxy) - or use the fanout_heads() context manager."""
    rank, world = dist.get_rank(), dist.get_world_size()
    if world > 8:
        raise ValueError("recon-head fan-out: a chunk has 8 pictures, at most 8 ranks can take part")
    if not 0 <= src < world:
        raise ValueError("bad src rank")
    mask = head_mask((rank - src) % world, world)
    if rank == src:
        if getattr(proxy, "recon_mask", lambda: None)() != 0:
            proxy.set_recon_mask(0)
        proxy.decompress(bit_stream, qp, height, width, ec_parallel, reset)
        feature = _to_wire(proxy.export_feature(), dist)
        work = dist.broadcast(feature, src, async_op=True) if world > 1 else None
        out = proxy.run_recon_heads(mask, height, width)
        if work is not None:
            work.wait()                  # `feature` must outlive the transfer; the heads above did not wait for it
    else:
        h8, w8 = (height + 15) // 16 * 2, (width + 15) // 16 * 2
        # one receive buffer per proxy and picture size (round 3 allocated 33 MB per chunk)
        key = (h8 * w8 * 512, str(_device_for(dist)))
        cache = getattr(proxy, "_fanout_rx", None)
        if cache is None or cache[0] != key:
            cache = (key, torch.empty(key[0], dtype=torch.float16, device=_device_for(dist)))
            try:
                proxy._fanout_rx = cache
            except AttributeError:
                pass
        feature = cache[1]
        dist.broadcast(feature, src)
        if not feature.is_cuda and _has_gpu():
            feature = feature.cuda()             # gloo on a GPU box: the transfer went through host memory
        proxy.import_feature(feature, height, width)
        out = proxy.run_recon_heads(mask, height, width)
    return {i: out[i] for i in range(8) if mask >> i & 1}


def restore_heads(proxy):
    """After decompress_fanout: decompress() of this proxy runs all 8 reconstruction heads again."""
    proxy.set_recon_mask(0xFF)


class fanout_heads:
    """`with fanout_heads(proxy): ... decompress_fanout(proxy, ...) ...` - the owner's recon mask is back at all
    8 heads when the block ends (one graph re-capture at each end, none per chunk)."""

    def __init__(self, proxy):
        self.proxy = proxy

    def __enter__(self):
        return self.proxy

    def __exit__(self, *exc):
        restore_heads(self.proxy)
        return False


def gather_pictures(mine, dist, dst=0, src=0, shape=None, dtype=torch.float16, device=None):
    """{picture: [1, 3, H, W] tensor} of every rank -> on rank `dst` the 8 pictures in display order (None
    elsewhere). Point-to-point: every picture travels once, from the rank that reconstructed it
    (head_owner, relative to the fan-out's `src` - pass the SAME src as to decompress_fanout: checked) to `dst`.
    `shape` / `dtype` / `device` describe a picture for a `dst` that owns none itself; by default they are taken from
    one of its own."""
    world, rank = dist.get_world_size(), dist.get_rank()
    if world == 1:
        return [mine[i] for i in range(8)]
    if rank == dst and shape is None and not mine:
        raise ValueError("gather_pictures: rank %d owns no picture - pass shape / dtype / device" % rank)
    want = {i for i in range(8) if (head_owner(i, world) + src) % world == rank}
    if set(mine) != want:
        raise ValueError("gather_pictures: rank %d holds pictures %s but owns %s for src=%d - the fan-out ran with another src"
                         % (rank, sorted(mine), sorted(want), src))
    if rank != dst:
        reqs = [dist.isend(_to_wire(mine[i].contiguous(), dist), dst) for i in sorted(mine)]
        for r in reqs:
            r.wait()
        return None
    if shape is None:
        any_t = next(iter(mine.values()))
        shape, dtype, device = tuple(any_t.shape), any_t.dtype, any_t.device
    out, reqs = [None] * 8, []
    wire_dev = _device_for(dist)
    for i in range(8):
        owner = (head_owner(i, world) + src) % world
        if owner == dst:
            out[i] = mine[i]
        else:
            out[i] = torch.empty(shape, dtype=dtype, device=wire_dev if dist.get_backend() != "nccl" else
                                 (device if device is not None else wire_dev))
            reqs.append(dist.irecv(out[i], owner))
    for r in reqs:
        r.wait()
    if device is not None:
        out = [t if t.device == torch.device(device) else t.to(device) for t in out]
    return out


# ------------------------------------------------------------------ GOP hand-off between GPUs
# Inside a GOP the coding order is strictly sequential, but WHERE the next picture is coded is free
# as long as the temporal state travels with it: DMCLDProxy.export_state() packs it into one flat
# uint8 device tensor (reference feature, memory, last decoded feature, context, temporal prior:
# 84 MB at 1080p), which moves point-to-point over xGMI with backend "nccl" (= RCCL).
def send_state(proxy, dst, dist):
    """-> bytes sent. Device tensors over RCCL; staged through host memory where the backend has no device
    point-to-point (gloo on a GPU box: bench.py's one-device rehearsal, tests) - `_to_wire`, like the fan-out."""
    state = _to_wire(proxy.export_state(), dist)
    dist.send(torch.tensor([state.numel()], dtype=torch.int64, device=state.device), dst)
    dist.send(state, dst)
    return state.numel()


def recv_state(proxy, src, height, width, dist, device=None):
    dev = device if device is not None else _device_for(dist)
    n = torch.zeros(1, dtype=torch.int64, device=dev)
    dist.recv(n, src)
    # one receive buffer per proxy and size (a hand-off every few pictures would otherwise allocate 84 MB each time)
    key = (int(n.item()), str(dev))
    cache = getattr(proxy, "_state_rx", None)
    if cache is None or cache[0] != key:
        cache = (key, torch.empty(key[0], dtype=torch.uint8, device=dev))
        try:
            proxy._state_rx = cache
        except AttributeError:
            pass
    state = cache[1]
    dist.recv(state, src)
    if not state.is_cuda and _has_gpu():
        # gloo on a GPU box: the transfer went through host memory; the device staging copy is cached with the receive buffer
        dev_buf = cache[2] if len(cache) > 2 else None
        if dev_buf is None or dev_buf.numel() != state.numel():
            dev_buf = torch.empty(state.numel(), dtype=torch.uint8, device="cuda")
            try:
                proxy._state_rx = (cache[0], cache[1], dev_buf)
            except AttributeError:
                pass
        dev_buf.copy_(state)
        state = dev_buf
    proxy.import_state(state, height, width)
    return state.numel()


def release_state_buffers(proxy):
    """Drops the receive / staging buffers recv_state() keeps on a proxy (84 MB per encoder or decoder object at 1080p, about
    4x that at 3840x2160) - for a rank that will not take over a stream again. The next recv_state() allocates them anew."""
    try:
        del proxy._state_rx
    except AttributeError:
        pass
