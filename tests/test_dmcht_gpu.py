"""DMC HT-S / HT-L (hierarchical inter, 8 pictures per call) parity on a real MI355X (-m gpu),
through the reference's plugin surface (inference_extensions_cuda.DMCHTSProxy / DMCHTLProxy over
the C ABI): bit-exact against the CPU oracle - rANS bytes, temporal state, all 8 reconstructions."""
import copy
import os

import numpy as np
import pytest
import torch

from codec_util import (chunk, dmc_ht_model, dmci_model, from_device_output, oracle_for, picture, psnr,
                        to_device_input)

pytestmark = pytest.mark.gpu


def _gpu_net(model):
    g = copy.deepcopy(model).half().cuda()      # finalize_model, test_video.py:27-29
    g.proxy = None
    return g


def _pads(g, h, w):
    pr, pb = g.get_padding_size(h, w, 16)
    return pb, pr


def _padded(x_hwc):
    h, w, _ = x_hwc.shape
    return np.pad(x_hwc, ((0, -h % 16), (0, -w % 16), (0, 0)), mode="edge")


def _decoded(xs):
    return np.concatenate([from_device_output(t) for t in xs], axis=-1)


@pytest.mark.parametrize("structure,hw,plan,thres", [
    ("hts", (64, 64), [(32, 0), (40, 1), (12, 0)], 0.15),
    ("hts", (40, 72), [(7, 0)], 0.0),
    ("htl", (64, 64), [(32, 0), (50, 1), (20, 0)], 0.15),
    ("htl", (40, 72), [(60, 0)], 0.0),
])
def test_sequence_matches_oracle(structure, hw, plan, thres):
    m = dmc_ht_model(structure, skip_thres=thres)
    enc_o, dec_o = oracle_for(m), oracle_for(m)
    enc_g, dec_g = _gpu_net(m), _gpu_net(m)
    ref = _padded(picture(*hw, index=0))
    enc_o.add_ref_feature_from_frame(ref, True)
    dec_o.add_ref_feature_from_frame(ref, False)
    enc_g.add_ref_feature_from_frame(to_device_input(ref))
    dec_g.add_ref_feature_from_frame(to_device_input(ref), apply_feature_adaptor=False)
    pb, pr = _pads(enc_g, *hw)
    sps = {"height": hw[0], "width": hw[1]}
    for i, (qp, reset) in enumerate(plan):
        x = chunk(hw[0], hw[1], 1 + 8 * i)
        if i == 0:
            for name, want in (("memory", enc_o.memory), ("ctx", enc_o.ctx)):
                assert np.array_equal(enc_g.proxy.debug_read(name, np.float16).reshape(want.shape), want), name
        want = enc_o.compress(x, qp, bool(reset))
        got = enc_g.compress(to_device_input(x), qp, reset, pb, pr)
        torch.cuda.synchronize()
        y_gpu = enc_g.proxy.debug_read("y", np.float16).reshape(enc_o.debug["y"].shape)
        print(structure, "chunk", i, "y mismatches:", int((y_gpu != enc_o.debug["y"]).sum()), "of", y_gpu.size)
        assert np.array_equal(enc_g.proxy.debug_read("z_i8", np.int8), enc_o.debug["z_i8"].reshape(-1))
        assert np.array_equal(y_gpu, enc_o.debug["y"])
        assert np.array_equal(enc_g.proxy.debug_read("y_hat", np.float16).reshape(y_gpu.shape), enc_o.debug["y_hat"])
        assert got["ec_parallel"] == want["ec_parallel"]
        assert got["bit_stream"] == want["bit_stream"], "rANS bitstream differs from the oracle's"
        for name, w in (("feature_p", enc_o.feature_p), ("memory", enc_o.memory), ("ctx", enc_o.ctx)):
            assert np.array_equal(enc_g.proxy.debug_read(name, np.float16).reshape(w.shape), w), name
        xd_want = np.concatenate(
            dec_o.decompress(want["bit_stream"], qp, hw[0], hw[1], want["ec_parallel"], bool(reset)), axis=-1)
        xd = dec_g.decompress(got["bit_stream"], sps, qp, got["ec_parallel"], reset)["x_hat"]
        torch.cuda.synchronize()
        assert len(xd) == 8
        xd = _decoded(xd)
        assert np.array_equal(xd, xd_want), "reconstructions differ from the oracle's"
        print(structure, "chunk", i, "bytes", len(got["bit_stream"]), "PSNR(x_hat, x) = %.2f dB" %
              psnr(xd[:hw[0], :hw[1]], x))


@pytest.mark.parametrize("structure", ["hts", "htl"])
def test_graph_replay_equals_eager(structure):
    m = dmc_ht_model(structure, skip_thres=0.15)
    hw = (96, 160)
    plan = [(10, 0), (50, 1), (30, 0), (30, 0), (12, 1), (40, 0)]
    ref = to_device_input(_padded(picture(*hw, index=0)))
    xs = [to_device_input(chunk(hw[0], hw[1], 1 + 8 * i)) for i in range(len(plan))]
    sps = {"height": hw[0], "width": hw[1]}
    results = {}
    for graphs in (False, True):
        enc, dec = _gpu_net(m), _gpu_net(m)
        enc._ensure_proxy().set_use_graphs(graphs)
        dec._ensure_proxy().set_use_graphs(graphs)
        pb, pr = _pads(enc, *hw)
        out = []
        enc.add_ref_feature_from_frame(ref)
        dec.add_ref_feature_from_frame(ref, apply_feature_adaptor=False)
        for (qp, reset), x in zip(plan, xs):
            r = enc.compress(x, qp, reset, pb, pr)
            d = dec.decompress(r["bit_stream"], sps, qp, r["ec_parallel"], reset)["x_hat"]
            torch.cuda.synchronize()
            out.append((r["bit_stream"], torch.cat(d, 0).clone()))
        results[graphs] = out
    for (b0, x0), (b1, x1) in zip(results[False], results[True]):
        assert b0 == b1
        assert torch.equal(x0, x1)


@pytest.mark.parametrize("structure", ["hts", "htl"])
def test_full_hd_closure(structure):
    """BASELINE config 2 size: I picture by the intra codec, two chunks of 8 P pictures (the
    second resets the memory); the decoder stays in lock-step with the encoder."""
    mi, mp = dmci_model(skip_thres=0.15), dmc_ht_model(structure, skip_thres=0.15)
    i_enc, p_enc, p_dec = _gpu_net(mi), _gpu_net(mp), _gpu_net(mp)
    H, W = 1080, 1920
    pb, pr = _pads(i_enc, H, W)
    sps = {"height": H, "width": W}
    e = i_enc.compress(to_device_input(picture(H, W, index=0)), 30, pb, pr)
    p_enc.add_ref_feature_from_frame(e["x_hat"])
    p_dec.add_ref_feature_from_frame(e["x_hat"], apply_feature_adaptor=False)
    pics = [picture(H, W, index=1 + j) for j in range(8)]
    for i, (qp, reset) in enumerate([(34, 1), (30, 0)]):
        x = to_device_input(np.concatenate(pics[i:] + pics[:i], axis=-1))
        r = p_enc.compress(x, qp, reset, pb, pr)
        xd = p_dec.decompress(r["bit_stream"], sps, qp, r["ec_parallel"], reset)["x_hat"]
        torch.cuda.synchronize()
        assert len(xd) == 8 and xd[0].shape == (1, 3, 1088, 1920)
        allx = torch.cat(xd, 0)
        assert torch.isfinite(allx.float()).all() and allx.abs().max() <= 0.5
        fe = p_enc.proxy.debug_read("feature_p", np.float16)
        fd = p_dec.proxy.debug_read("feature_p", np.float16)
        assert np.array_equal(fe, fd), "decoder drifted from the encoder at chunk %d" % i
        print(structure, "chunk %d qp %d reset %d: %d bytes, ec_parallel %d" %
              (i, qp, reset, len(r["bit_stream"]), r["ec_parallel"]))
        assert len(r["bit_stream"]) > 1000


def test_wrong_structure_and_missing_reference_fail_loudly():
    import inference_extensions_cuda as ext
    m = dmc_ht_model("hts", skip_thres=0.15)
    sd = m.add_cdf_to_state_dict(m.state_dict())
    with pytest.raises(Exception, match="HT-S state_dict"):
        ext.DMCHTLProxy().set_param(sd, 0.15)
    g = _gpu_net(m)
    with pytest.raises(Exception, match="reference feature"):
        g.compress(to_device_input(chunk(64, 64, 1)), 20, 0, 0, 0)


def test_gop_hand_off_continues_bit_exactly():
    """HT-S encoder / decoder state exported after a chunk and imported into fresh codec objects
    continue with exactly the bytes / pixels of the uninterrupted codecs."""
    m = dmc_ht_model("hts", skip_thres=0.15)
    hw = (64, 96)
    ref = to_device_input(_padded(picture(*hw, index=0)))
    xs = [to_device_input(chunk(hw[0], hw[1], 1 + 8 * i)) for i in range(2)]
    sps = {"height": hw[0], "width": hw[1]}
    enc, dec, enc2, dec2 = _gpu_net(m), _gpu_net(m), _gpu_net(m), _gpu_net(m)
    pb, pr = _pads(enc, *hw)
    enc.add_ref_feature_from_frame(ref)
    dec.add_ref_feature_from_frame(ref, apply_feature_adaptor=False)
    r0 = enc.compress(xs[0], 30, 0, pb, pr)
    dec.decompress(r0["bit_stream"], sps, 30, r0["ec_parallel"], 0)
    enc2._ensure_proxy().import_state(enc.proxy.export_state(), hw[0], hw[1])
    dec2._ensure_proxy().import_state(dec.proxy.export_state(), hw[0], hw[1])
    r1 = enc.compress(xs[1], 34, 0, pb, pr)
    d1 = torch.cat(dec.decompress(r1["bit_stream"], sps, 34, r1["ec_parallel"], 0)["x_hat"], 0).clone()
    r2 = enc2.compress(xs[1], 34, 0, pb, pr)
    d2 = torch.cat(dec2.decompress(r2["bit_stream"], sps, 34, r2["ec_parallel"], 0)["x_hat"], 0)
    torch.cuda.synchronize()
    assert r2["bit_stream"] == r1["bit_stream"] and torch.equal(d2, d1)
    with pytest.raises(Exception, match="model structure"):
        _gpu_net(dmc_ht_model("htl", skip_thres=0.15))._ensure_proxy().import_state(enc.proxy.export_state(), hw[0], hw[1])


@pytest.mark.parametrize("structure", ["hts", "htl"])
def test_recon_head_fan_out_equals_one_gpu(structure):
    """the owner of the stream decodes with its own head mask and exports feature_p,
    a second codec object (standing in for another GPU) imports it and runs the remaining heads: the
    8 pictures equal a plain decompress() bit for bit, for every split sharding.head_mask produces,
    and the owner's temporal state carries on (the next chunk decodes identically)."""
    from dcvc_amd import sharding
    m = dmc_ht_model(structure, skip_thres=0.15)
    hw = (96, 160)
    sps = {"height": hw[0], "width": hw[1]}
    ref = to_device_input(_padded(picture(*hw, index=0)))
    enc, plain = _gpu_net(m), _gpu_net(m)
    enc.add_ref_feature_from_frame(ref)
    plain.add_ref_feature_from_frame(ref, apply_feature_adaptor=False)
    pb, pr = _pads(enc, *hw)
    streams, want = [], []
    for i, (qp, reset) in enumerate([(20, 0), (44, 1), (30, 0)]):
        r = enc.compress(to_device_input(chunk(hw[0], hw[1], 1 + 8 * i)), qp, reset, pb, pr)
        d = plain.decompress(r["bit_stream"], sps, qp, r["ec_parallel"], reset)["x_hat"]
        torch.cuda.synchronize()
        streams.append((r, qp, reset))
        want.append([t.clone() for t in d])
    for world in (2, 4, 8):
        owner = _gpu_net(m)
        owner.add_ref_feature_from_frame(ref, apply_feature_adaptor=False)
        helpers = [_gpu_net(m) for _ in range(world - 1)]
        po = owner._ensure_proxy()
        po.set_recon_mask(sharding.head_mask(0, world))
        for c, (r, qp, reset) in enumerate(streams):
            out = po.decompress(np.frombuffer(r["bit_stream"], dtype=np.uint8), qp, hw[0], hw[1], r["ec_parallel"], reset)
            feature = po.export_feature()
            got = {i: out[i].clone() for i in range(8) if sharding.head_owner(i, world) == 0}
            for k, h in enumerate(helpers):
                ph = h._ensure_proxy()
                ph.import_feature(feature, hw[0], hw[1])
                o = ph.run_recon_heads(sharding.head_mask(k + 1, world), hw[0], hw[1])
                torch.cuda.synchronize()
                got.update({i: o[i].clone() for i in range(8) if sharding.head_owner(i, world) == k + 1})
            torch.cuda.synchronize()
            assert sorted(got) == list(range(8))
            for i in range(8):
                assert torch.equal(got[i], want[c][i]), (world, c, i)


class _LoopbackDist:
    """torch.distributed of a 2-rank job played by ONE process: rank 0's broadcast parks the tensor, rank 1's picks
    it up - enough to drive sharding.decompress_fanout itself (owner without heads -> export -> async broadcast
    -> own heads; peer: import -> heads) on a 1-GPU box."""
    mailbox = {}

    class _Work:
        def wait(self):
            pass

    def __init__(self, rank, world=2):
        self.rank, self.world = rank, world

    def get_rank(self):
        return self.rank

    def get_world_size(self):
        return self.world

    def get_backend(self):
        return "nccl"

    def broadcast(self, t, src, async_op=False):
        if self.rank == src:
            type(self).mailbox["t"] = t.clone()
        else:
            t.copy_(type(self).mailbox["t"])
        return self._Work() if async_op else None


@pytest.mark.parametrize("structure", ["hts", "htl"])
def test_decompress_fanout_runs_the_owners_heads_behind_the_export(structure):
    """sharding.decompress_fanout as bench.py --fanout calls it: three chunks incl. a memory reset, the 8 pictures of
    every chunk equal a plain decompress() bit for bit, the owner's temporal state carries on, and after
    restore_heads() the owner's proxy decodes plainly again."""
    from dcvc_amd import sharding
    m = dmc_ht_model(structure, skip_thres=0.15)
    hw = (96, 160)
    sps = {"height": hw[0], "width": hw[1]}
    ref = to_device_input(_padded(picture(*hw, index=0)))
    enc, plain, owner, peer = _gpu_net(m), _gpu_net(m), _gpu_net(m), _gpu_net(m)
    enc.add_ref_feature_from_frame(ref)
    for d in (plain, owner):
        d.add_ref_feature_from_frame(ref, apply_feature_adaptor=False)
    pb, pr = _pads(enc, *hw)
    po, pp = owner._ensure_proxy(), peer._ensure_proxy()
    plan = [(20, 0), (44, 1), (30, 0), (36, 0)]
    for i, (qp, reset) in enumerate(plan):
        r = enc.compress(to_device_input(chunk(hw[0], hw[1], 1 + 8 * i)), qp, reset, pb, pr)
        want = [t.clone() for t in plain.decompress(r["bit_stream"], sps, qp, r["ec_parallel"], reset)["x_hat"]]
        torch.cuda.synchronize()
        bits = np.frombuffer(r["bit_stream"], dtype=np.uint8)
        if i < 3:
            got = dict(sharding.decompress_fanout(po, bits, qp, hw[0], hw[1], r["ec_parallel"], bool(reset), _LoopbackDist(0)))
            got = {k: v.clone() for k, v in got.items()}
            mine = sharding.decompress_fanout(pp, None, qp, hw[0], hw[1], 0, bool(reset), _LoopbackDist(1))
            torch.cuda.synchronize()
            assert not set(got) & set(mine)
            got.update(mine)
        else:
            sharding.restore_heads(po)
            out = po.decompress(bits, qp, hw[0], hw[1], r["ec_parallel"], bool(reset))
            torch.cuda.synchronize()
            got = dict(enumerate(out))
        assert sorted(got) == list(range(8))
        for k in range(8):
            assert torch.equal(got[k], want[k]), (i, k)


def _fanout_rank(rank, world, port, structure, out_path):
    import pickle
    import torch.distributed as dist
    from dcvc_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        m = dmc_ht_model(structure, skip_thres=0.15)
        hw = (96, 160)
        ref = to_device_input(_padded(picture(*hw, index=0)))
        dec = _gpu_net(m)
        r = None
        if rank == 0:
            enc = _gpu_net(m)
            enc.add_ref_feature_from_frame(ref)
            pb, pr = _pads(enc, *hw)
            r = enc.compress(to_device_input(chunk(hw[0], hw[1], 1)), 28, 0, pb, pr)
            dec.add_ref_feature_from_frame(ref, apply_feature_adaptor=False)
        p = dec._ensure_proxy()
        mine = sharding.decompress_fanout(p, None if r is None else np.frombuffer(r["bit_stream"], dtype=np.uint8), 28,
                                          hw[0], hw[1], 0 if r is None else r["ec_parallel"], False, dist)
        pics = sharding.gather_pictures(mine, dist)
        if rank == 0:
            plain = _gpu_net(m)
            plain.add_ref_feature_from_frame(ref, apply_feature_adaptor=False)
            want = plain.decompress(r["bit_stream"], {"height": hw[0], "width": hw[1]}, 28, r["ec_parallel"], 0)["x_hat"]
            torch.cuda.synchronize()
            ok = all(torch.equal(a, b) for a, b in zip(pics, want))
            with open(out_path, "wb") as f:
                pickle.dump(ok, f)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs (RCCL broadcast of feature_p over xGMI)")
def test_recon_head_fan_out_over_rccl(tmp_path):
    import pickle
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "ok.pkl")
    mp.spawn(_fanout_rank, args=(2, port, "hts", out), nprocs=2, join=True)
    with open(out, "rb") as f:
        assert pickle.load(f) is True
