"""All 64 q indexes of every codec against the live oracle on a real MI355X (-m gpu).

The reference captures one CUDA graph per (stage, qp) (dmc_common.cpp:85-134: 64 graph slots per
stage); this build keeps ONE graph per stage and copies the qp's rows of the q-scale tables into
fixed device slots in front of it (codec_base.hip::copy_qp_rows). A sweep over all 64 values
(test_video.py:512-514, BASELINE configs[4]: `--rate_num 64`) exercises exactly that mechanism, so
every row of every table is pinned here: rANS bytes, encoder-side reconstruction / temporal state
and the reconstruction of a decoder object that saw only the bytes, bit-exact against
oracle/codec.py at 64x64 with graph replay ON (and, for LD, with its default eager launches too).
The synthetic q-scale tables (dcvc_amd/synthetic.py) hold an independent random vector per row,
so a row read at the wrong index cannot pass."""
import copy

import numpy as np
import pytest
import torch

from codec_util import (chunk, dmc_ht_model, dmc_ld_model, dmci_model, from_device_output, oracle_for,
                        picture, to_device_input)

pytestmark = pytest.mark.gpu

HW = (64, 64)
SPS = {"height": HW[0], "width": HW[1]}
# every q index exactly once, neighbours far apart (37 is coprime to 64): a row offset by a constant, a
# stale slot from the previous call and a clamped index all show up
ORDER = [(37 * i + 5) % 64 for i in range(64)]


def _gpu_net(model, graphs):
    g = copy.deepcopy(model).half().cuda()      # finalize_model, test_video.py:27-29
    g.proxy = None
    g._ensure_proxy().set_use_graphs(graphs)
    return g


def test_order_covers_every_q_index():
    assert sorted(ORDER) == list(range(64))


def test_intra_all_64_q_indexes_match_oracle():
    m = dmci_model(skip_thres=0.15)
    o = oracle_for(m)
    enc, dec = _gpu_net(m, True), _gpu_net(m, True)
    sizes = set()
    for i, qp in enumerate(ORDER):
        x = picture(*HW, index=i % 5)
        want = o.compress(x, qp)
        got = enc.compress(to_device_input(x), qp, 0, 0)
        torch.cuda.synchronize()
        assert got["ec_parallel"] == want["ec_parallel"], qp
        assert got["bit_stream"] == want["bit_stream"], "q %d: rANS bytes differ from the oracle's" % qp
        assert np.array_equal(from_device_output(got["x_hat"]), want["x_hat"]), qp
        d = dec.decompress(got["bit_stream"], SPS, qp, got["ec_parallel"])["x_hat"]
        torch.cuda.synchronize()
        assert np.array_equal(from_device_output(d), want["x_hat"]), "q %d: decoder reconstruction" % qp
        sizes.add(len(got["bit_stream"]))
    assert len(sizes) > 16      # the rate really moves with q


@pytest.mark.parametrize("graphs", [True, False])
def test_ld_all_64_q_indexes_match_oracle(graphs):
    m = dmc_ld_model(skip_thres=0.15)
    enc_o, dec_o = oracle_for(m), oracle_for(m)
    enc_g, dec_g = _gpu_net(m, graphs), _gpu_net(m, graphs)
    ref = picture(*HW, index=0)
    enc_o.add_ref_feature_from_frame(ref, True)
    dec_o.add_ref_feature_from_frame(ref, False)
    enc_g.add_ref_feature_from_frame(to_device_input(ref))
    dec_g.add_ref_feature_from_frame(to_device_input(ref), apply_feature_adaptor=False)
    for i, qp in enumerate(ORDER):
        reset = i % 16 == 9          # index_map cadence of the reference is 8 pictures; any reset will do
        x = picture(*HW, index=1 + i % 7)
        want = enc_o.compress(x, qp, reset)
        got = enc_g.compress(to_device_input(x), qp, reset, 0, 0)
        torch.cuda.synchronize()
        assert got["ec_parallel"] == want["ec_parallel"], qp
        assert got["bit_stream"] == want["bit_stream"], "q %d: rANS bytes differ from the oracle's" % qp
        for name, w in (("feature_p", enc_o.feature_p), ("memory", enc_o.memory), ("ctx", enc_o.ctx),
                        ("temporal", enc_o.temporal)):
            assert np.array_equal(enc_g.proxy.debug_read(name, np.float16).reshape(w.shape), w), (qp, name)
        xd_want = dec_o.decompress(want["bit_stream"], qp, HW[0], HW[1], want["ec_parallel"], reset)
        xd = dec_g.decompress(got["bit_stream"], SPS, qp, got["ec_parallel"], reset)["x_hat"]
        torch.cuda.synchronize()
        assert np.array_equal(from_device_output(xd), xd_want), "q %d: decoder reconstruction" % qp
        assert np.array_equal(dec_g.proxy.debug_read("feature_p", np.float16).reshape(enc_o.feature_p.shape),
                              enc_o.feature_p), qp


@pytest.mark.parametrize("structure", ["hts", "htl"])
def test_ht_all_64_q_indexes_match_oracle(structure):
    m = dmc_ht_model(structure, skip_thres=0.15)
    enc_o, dec_o = oracle_for(m), oracle_for(m)
    enc_g, dec_g = _gpu_net(m, True), _gpu_net(m, True)
    ref = picture(*HW, index=0)
    enc_o.add_ref_feature_from_frame(ref, True)
    dec_o.add_ref_feature_from_frame(ref, False)
    enc_g.add_ref_feature_from_frame(to_device_input(ref))
    dec_g.add_ref_feature_from_frame(to_device_input(ref), apply_feature_adaptor=False)
    chunks = [chunk(HW[0], HW[1], 1 + 8 * j) for j in range(3)]
    for i, qp in enumerate(ORDER):
        reset = i % 16 == 9
        x = chunks[i % 3]
        want = enc_o.compress(x, qp, reset)
        got = enc_g.compress(to_device_input(x), qp, reset, 0, 0)
        torch.cuda.synchronize()
        assert got["ec_parallel"] == want["ec_parallel"], qp
        assert got["bit_stream"] == want["bit_stream"], "%s q %d: rANS bytes differ from the oracle's" % (structure, qp)
        for name, w in (("feature_p", enc_o.feature_p), ("memory", enc_o.memory), ("ctx", enc_o.ctx)):
            assert np.array_equal(enc_g.proxy.debug_read(name, np.float16).reshape(w.shape), w), (qp, name)
        xd_want = np.concatenate(
            dec_o.decompress(want["bit_stream"], qp, HW[0], HW[1], want["ec_parallel"], reset), axis=-1)
        xd = dec_g.decompress(got["bit_stream"], SPS, qp, got["ec_parallel"], reset)["x_hat"]
        torch.cuda.synchronize()
        xd = np.concatenate([from_device_output(t) for t in xd], axis=-1)
        assert np.array_equal(xd, xd_want), "%s q %d: decoder reconstructions" % (structure, qp)
