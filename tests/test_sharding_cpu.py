"""N > 1 path on CPU: world_size-2 gloo processes shard a coding job and rank 0 collects the coded
units in order (dcvc_amd/sharding.py). The per-unit coder is a stand-in (the codec itself has no
CPU path); what is under test is the unit assignment and the collection collectives."""
import hashlib
import os
import pickle
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from dcvc_amd import sharding


def test_shard_range_covers_everything_once():
    for n in (0, 1, 7, 8, 33):
        for world in (1, 2, 3, 8):
            got = [i for r in range(world) for i in sharding.shard_range(n, r, world)]
            assert got == list(range(n))
            sizes = [len(sharding.shard_range(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def test_plan_gops_follows_reference_intra_placement():
    assert sharding.plan_gops(5, 1) == [(i, 1) for i in range(5)]
    assert sharding.plan_gops(97, -1, 8) == [(0, 97)]
    # test_video.py:204-213: I pictures at 0 and at every index % 32 == 1 except index 1
    assert sharding.plan_gops(100, 32) == [(0, 33), (33, 32), (65, 32), (97, 3)]
    with pytest.raises(ValueError):
        sharding.plan_gops(100, 12, 8)


def _fake_code(i):
    return hashlib.sha256(b"unit %d" % i).digest() * (1 + i % 5) if i % 4 else b""


def _worker(rank, world, port, n_units, out_path):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = sharding.code_sharded(n_units, _fake_code, dist)
        if rank == 0:
            with open(out_path, 'wb') as f:
                pickle.dump(res, f)
        else:
            assert res is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_units", [7, 1])
def test_two_ranks_collect_units_in_order(tmp_path, n_units):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "units.pt")
    mp.spawn(_worker, args=(2, port, n_units, out), nprocs=2, join=True)
    with open(out, 'rb') as f:
        got = pickle.load(f)
    assert got == [_fake_code(i) for i in range(n_units)]


def test_single_process_path():
    assert sharding.code_sharded(5, _fake_code) == [_fake_code(i) for i in range(5)]


class _FakeProxy:
    """Stands in for DMCLDProxy on the CPU: the hand-off helpers only move its opaque state."""

    def __init__(self, fill=None):
        self.state = None if fill is None else torch.arange(fill, dtype=torch.int64).to(torch.uint8)
        self.size = None

    def export_state(self):
        return self.state

    def import_state(self, state, height, width):
        self.state, self.size = state.clone(), (height, width)


def _handoff_worker(rank, world, port, out_path):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if rank == 0:
            sharding.send_state(_FakeProxy(fill=100001), 1, dist)
        else:
            p = _FakeProxy()
            n = sharding.recv_state(p, 0, 1080, 1920, dist)
            with open(out_path, "wb") as f:
                pickle.dump((n, p.size, p.state.numpy().tobytes()), f)
    finally:
        dist.destroy_process_group()


def test_gop_state_hand_off_between_two_ranks(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "state.pkl")
    mp.spawn(_handoff_worker, args=(2, port, out), nprocs=2, join=True)
    with open(out, "rb") as f:
        n, size, data = pickle.load(f)
    want = torch.arange(100001, dtype=torch.int64).to(torch.uint8).numpy().tobytes()
    assert n == 100001 and size == (1080, 1920) and data == want


# ------------------------------------------------------------------ recon-head fan-out (HT models)
def test_head_masks_partition_the_chunk():
    for world in (1, 2, 3, 4, 5, 8):
        masks = [sharding.head_mask(r, world) for r in range(world)]
        assert sum(masks) == 0xFF and all(a & b == 0 for i, a in enumerate(masks) for b in masks[i + 1:])
        assert masks[0] & 0x80, "the last picture (reset feature) stays with the rank that owns the stream"
    # 2 and 4 ranks keep the picture pairs that share a trunk block together
    assert all(sharding.head_owner(2 * j, w) == sharding.head_owner(2 * j + 1, w) for w in (2, 4) for j in range(4))


class _FakeHT:
    """Stands in for DMCHTSProxy on the CPU: feature_p is a function of the bytes, picture i a function of
    feature_p - enough to check who computes what and that rank 0 ends up with all 8 pictures."""
    H8, W8 = 4, 6

    def __init__(self):
        self.mask, self.feature, self.decoded, self.head_calls = 0xFF, None, 0, []

    def set_recon_mask(self, mask):
        self.mask = mask

    def _heads(self, mask):
        f = self.feature.float().reshape(self.H8 * self.W8, 512)
        return [(f[:, :3].t().reshape(1, 3, self.H8, self.W8) * (i + 1)).half() if mask >> i & 1 else None for i in range(8)]

    def decompress(self, bit_stream, qp, height, width, ec, reset):
        self.decoded += 1
        g = torch.Generator().manual_seed(len(bit_stream) + qp)
        self.feature = torch.randn(self.H8 * self.W8 * 512, generator=g).half()
        self.head_calls.append((self.mask, "decompress"))
        return self._heads(self.mask)

    def export_feature(self):
        return self.feature.clone()

    def import_feature(self, feature, height, width):
        self.feature = feature.clone()

    def run_recon_heads(self, mask, height, width):
        self.head_calls.append((mask, "run_recon_heads"))
        return self._heads(mask)


def _fanout_worker(rank, world, port, out_path):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = _FakeHT()
        mine = sharding.decompress_fanout(p, b"x" * 77 if rank == 0 else None, 30, 32, 48, 1, False, dist)
        assert set(mine) == {i for i in range(8) if sharding.head_owner(i, world) == rank}
        assert p.decoded == (1 if rank == 0 else 0), "only the owner of the stream touches the bytes"
        if rank == 0:
            # the owner decodes WITHOUT heads (they run behind the start of the broadcast) and says so
            assert p.mask == 0 and p.head_calls == [(0, "decompress"), (sharding.head_mask(0, world), "run_recon_heads")]
            sharding.restore_heads(p)
            assert p.mask == 0xFF
        pics = sharding.gather_pictures(mine, dist)
        if rank == 0:
            with open(out_path, "wb") as f:
                pickle.dump([t.numpy().tobytes() for t in pics], f)
        else:
            assert pics is None
    finally:
        dist.destroy_process_group()


def test_recon_head_fan_out_over_two_ranks(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "pics.pkl")
    mp.spawn(_fanout_worker, args=(2, port, out), nprocs=2, join=True)
    with open(out, "rb") as f:
        got = pickle.load(f)
    ref = _FakeHT()
    want = [t.numpy().tobytes() for t in ref.decompress(b"x" * 77, 30, 32, 48, 1, False)]
    assert got == want


class _NineRanks:
    def get_rank(self):
        return 0

    def get_world_size(self):
        return 9


def test_fan_out_rejects_more_ranks_than_pictures():
    with pytest.raises(ValueError, match="at most 8 ranks"):
        sharding.decompress_fanout(_FakeHT(), b"x", 30, 32, 48, 1, False, _NineRanks())


class _Solo:
    """a rank that reconstructed nothing (cannot happen with <= 8 ranks; the error must still be a clean one, not a
    StopIteration with the other ranks hanging in a collective - ADVICE round 2)"""

    def get_world_size(self):
        return 3

    def get_rank(self):
        return 2


def test_gather_needs_a_shape_when_the_collector_owns_nothing():
    with pytest.raises(ValueError, match="owns no picture"):
        sharding.gather_pictures({}, _Solo(), dst=2)


class _TwoRanks:
    def __init__(self, rank):
        self.rank = rank

    def get_world_size(self):
        return 2

    def get_rank(self):
        return self.rank

    def get_backend(self):
        return "gloo"


def test_gather_checks_that_the_pictures_belong_to_this_rank_for_the_given_src():
    """advisor, round 3: gather_pictures needs the fan-out's `src`; a caller that forgets it (src 1, default 0) used to hang
    or collect the wrong pictures - now the mismatch between what the rank holds and what it owns is an error up front"""
    world = 2
    mine_of_rank0_for_src1 = {i: torch.zeros(1) for i in range(8) if (sharding.head_owner(i, world) + 1) % world == 0}
    with pytest.raises(ValueError, match="another src"):
        sharding.gather_pictures(mine_of_rank0_for_src1, _TwoRanks(0), dst=0, src=0)


def test_fanout_heads_context_restores_the_owner():
    p = _FakeHT()
    p.set_recon_mask(0)
    with sharding.fanout_heads(p) as q:
        assert q is p and p.mask == 0
    assert p.mask == 0xFF
