"""DMC-LD (low-delay inter) codec parity on a real MI355X (-m gpu), through the reference's plugin
surface (inference_extensions_cuda.DMCLDProxy over the C ABI): bit-exact against the CPU oracle
- identical rANS bytes, identical temporal state, identical reconstruction - over sequences with
qp changes and a feature-memory reset; graph replay; full-HD closure; error behaviour."""
import copy

import numpy as np
import pytest
import torch

from codec_util import (dmc_ld_model, dmci_model, from_device_output, oracle_for, picture, psnr,
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
    """edge-replicated to multiples of 16, like the intra codec's reconstruction"""
    h, w, _ = x_hwc.shape
    return np.pad(x_hwc, ((0, -h % 16), (0, -w % 16), (0, 0)), mode="edge")


@pytest.mark.parametrize("hw,plan,thres", [
    ((64, 64), [(32, 0), (40, 1), (40, 0)], 0.15),
    ((70, 100), [(7, 0), (63, 0)], 0.0),
    ((128, 96), [(50, 1), (20, 0)], 0.15),
])
def test_sequence_matches_oracle(hw, plan, thres):
    m = dmc_ld_model(skip_thres=thres)
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
        x = picture(*hw, index=i + 1)
        if i == 0:      # state derived from the reference frame
            for name, want in (("memory", enc_o.memory), ("ctx", enc_o.ctx), ("temporal", enc_o.temporal)):
                got = enc_g.proxy.debug_read(name, np.float16).reshape(want.shape)
                assert np.array_equal(got, want), name
        want = enc_o.compress(x, qp, bool(reset))
        got = enc_g.compress(to_device_input(x), qp, reset, pb, pr)
        torch.cuda.synchronize()
        y_gpu = enc_g.proxy.debug_read("y", np.float16).reshape(enc_o.debug["y"].shape)
        print("picture", i, "y mismatches:", int((y_gpu != enc_o.debug["y"]).sum()), "of", y_gpu.size)
        assert np.array_equal(enc_g.proxy.debug_read("z_i8", np.int8), enc_o.debug["z_i8"].reshape(-1))
        assert np.array_equal(y_gpu, enc_o.debug["y"])
        assert np.array_equal(enc_g.proxy.debug_read("y_hat", np.float16).reshape(y_gpu.shape), enc_o.debug["y_hat"])
        assert got["ec_parallel"] == want["ec_parallel"]
        assert got["bit_stream"] == want["bit_stream"], "rANS bitstream differs from the oracle's"
        for name, w in (("feature_p", enc_o.feature_p), ("memory", enc_o.memory), ("ctx", enc_o.ctx),
                        ("temporal", enc_o.temporal)):
            assert np.array_equal(enc_g.proxy.debug_read(name, np.float16).reshape(w.shape), w), name
        # decoder: its own object, fed only the bytes
        xd_want = dec_o.decompress(want["bit_stream"], qp, hw[0], hw[1], want["ec_parallel"], bool(reset))
        xd = dec_g.decompress(got["bit_stream"], sps, qp, got["ec_parallel"], reset)["x_hat"]
        torch.cuda.synchronize()
        xd = from_device_output(xd)
        assert np.array_equal(xd, xd_want), "reconstruction differs from the oracle's"
        assert np.array_equal(dec_g.proxy.debug_read("feature_p", np.float16).reshape(enc_o.feature_p.shape),
                              enc_o.feature_p)
        print("picture", i, "bytes", len(got["bit_stream"]), "PSNR(x_hat, x) = %.2f dB" %
              psnr(xd[:hw[0], :hw[1]], x))


def test_graph_replay_equals_eager():
    """Same bytes and pixels from hipGraph replay as from eager launches, picture after picture
    (every stage passes through warm-up, capture and replay; both reset variants; qp changes)."""
    m = dmc_ld_model(skip_thres=0.15)
    hw = (96, 160)
    plan = [(10, 0), (50, 0), (50, 1), (30, 0), (30, 0), (12, 1), (12, 0), (40, 0)]
    ref = to_device_input(_padded(picture(*hw, index=0)))
    xs = [to_device_input(picture(*hw, index=i + 1)) for i in range(len(plan))]
    sps = {"height": hw[0], "width": hw[1]}
    results = {}
    for graphs in (False, True):
        enc, dec = _gpu_net(m), _gpu_net(m)
        enc._ensure_proxy().set_use_graphs(graphs)
        dec._ensure_proxy().set_use_graphs(graphs)
        pb, pr = _pads(enc, *hw)
        out = []
        for rep in range(2):            # a second GOP re-enters the stages after add_ref
            enc.add_ref_feature_from_frame(ref)
            dec.add_ref_feature_from_frame(ref, apply_feature_adaptor=False)
            for (qp, reset), x in zip(plan, xs):
                r = enc.compress(x, qp, reset, pb, pr)
                d = dec.decompress(r["bit_stream"], sps, qp, r["ec_parallel"], reset)["x_hat"]
                torch.cuda.synchronize()
                out.append((r["bit_stream"], d.clone()))
        results[graphs] = out
    for (b0, x0), (b1, x1) in zip(results[False], results[True]):
        assert b0 == b1
        assert torch.equal(x0, x1)
    n = len(plan)
    for i in range(n):                  # both GOPs are identical
        assert results[True][i][0] == results[True][n + i][0]


def test_full_hd_gop_closure():
    """BASELINE config 1 size (1920x1080): I picture by the intra codec, P pictures by the LD codec
    with a reset; the decoder (separate objects, bytes only) stays in lock-step with the encoder:
    identical feature_p after every picture, finite in-range reconstructions."""
    mi, mp = dmci_model(skip_thres=0.15), dmc_ld_model(skip_thres=0.15)
    i_enc, i_dec, p_enc, p_dec = _gpu_net(mi), _gpu_net(mi), _gpu_net(mp), _gpu_net(mp)
    H, W = 1080, 1920
    pb, pr = _pads(i_enc, H, W)
    sps = {"height": H, "width": W}
    x0 = to_device_input(picture(H, W, index=0))
    e = i_enc.compress(x0, 30, pb, pr)
    p_enc.add_ref_feature_from_frame(e["x_hat"])
    d = i_dec.decompress(e["bit_stream"], sps, 30, e["ec_parallel"])
    p_dec.add_ref_feature_from_frame(d["x_hat"], apply_feature_adaptor=False)
    total = 0
    for i, (qp, reset) in enumerate([(34, 0), (30, 0), (34, 1), (30, 0)]):
        x = to_device_input(picture(H, W, index=i + 1))
        r = p_enc.compress(x, qp, reset, pb, pr)
        xd = p_dec.decompress(r["bit_stream"], sps, qp, r["ec_parallel"], reset)["x_hat"]
        torch.cuda.synchronize()
        assert xd.shape == (1, 3, 1088, 1920)
        assert torch.isfinite(xd.float()).all() and xd.abs().max() <= 0.5
        fe = p_enc.proxy.debug_read("feature_p", np.float16)
        fd = p_dec.proxy.debug_read("feature_p", np.float16)
        assert np.array_equal(fe, fd), "decoder drifted from the encoder at picture %d" % i
        total += len(r["bit_stream"])
        print("P picture %d qp %d reset %d: %d bytes, ec_parallel %d" % (i, qp, reset, len(r["bit_stream"]), r["ec_parallel"]))
    assert total > 4000


def test_compress_without_reference_fails_loudly():
    m = dmc_ld_model(skip_thres=0.15)
    g = _gpu_net(m)
    x = to_device_input(picture(64, 64))
    with pytest.raises(Exception, match="reference feature"):
        g.compress(x, 20, 0, 0, 0)
    with pytest.raises(Exception, match="reference feature"):
        g.decompress(b"\x00" * 32, {"height": 64, "width": 64}, 20, 1, 0)
    with pytest.raises(Exception, match="padding"):
        g.add_ref_feature_from_frame(x)
        g.compress(to_device_input(picture(70, 64)), 20, 0, 0, 0)


def test_corrupt_stream_does_not_crash():
    m = dmc_ld_model(skip_thres=0.15)
    enc, dec = _gpu_net(m), _gpu_net(m)
    ref = to_device_input(picture(64, 64))
    enc.add_ref_feature_from_frame(ref)
    dec.add_ref_feature_from_frame(ref, apply_feature_adaptor=False)
    r = enc.compress(to_device_input(picture(64, 64, index=1)), 20, 0, 0, 0)
    bad = bytearray(r["bit_stream"])
    for i in range(8, len(bad), 7):
        bad[i] ^= 0x5a
    # a damaged stream either decodes to garbage pictures or is rejected with a clean error (an escape code no
    # int8 symbol can have) - never a crash, a hang or an out-of-bounds read; the codec object stays usable
    from dcvc_amd._lib import DcvcError
    try:
        d = dec.decompress(bytes(bad[:len(bad) // 2]), {"height": 64, "width": 64}, 20, r["ec_parallel"], 0)
        torch.cuda.synchronize()
        assert torch.isfinite(d["x_hat"].float()).all()
    except DcvcError as e:
        assert "corrupt" in str(e)
    dec2 = _gpu_net(m)
    dec2.add_ref_feature_from_frame(ref, apply_feature_adaptor=False)
    dec.add_ref_feature_from_frame(ref, apply_feature_adaptor=False)
    good = [g.decompress(r["bit_stream"], {"height": 64, "width": 64}, 20, r["ec_parallel"], 0)["x_hat"].clone() for g in (dec, dec2)]
    torch.cuda.synchronize()
    assert torch.equal(good[0], good[1])


def test_gop_hand_off_continues_bit_exactly():
    """north_star (e): a GOP may move to another GPU between two pictures. Encoder and decoder
    state exported after picture 2 and imported into fresh codec objects continue with exactly the
    bytes / pixels of the uninterrupted codecs (same GPU here; the buffer is what
    dcvc_amd.sharding.send_state / recv_state move over RCCL)."""
    m = dmc_ld_model(skip_thres=0.15)
    hw = (96, 160)
    ref = to_device_input(_padded(picture(*hw, index=0)))
    xs = [to_device_input(picture(*hw, index=i + 1)) for i in range(4)]
    sps = {"height": hw[0], "width": hw[1]}
    enc, dec, enc2, dec2 = _gpu_net(m), _gpu_net(m), _gpu_net(m), _gpu_net(m)
    pb, pr = _pads(enc, *hw)
    enc.add_ref_feature_from_frame(ref)
    dec.add_ref_feature_from_frame(ref, apply_feature_adaptor=False)
    plan = [(30, 0), (34, 1), (30, 0), (34, 0)]
    want = []
    for i, ((qp, reset), x) in enumerate(zip(plan, xs)):
        if i == 2:      # hand the GOP over before picture 2
            s_enc, s_dec = enc.proxy.export_state(), dec.proxy.export_state()
            assert s_enc.dtype == torch.uint8 and s_enc.numel() == s_dec.numel() > 1000
            enc2._ensure_proxy().import_state(s_enc, hw[0], hw[1])
            dec2._ensure_proxy().import_state(s_dec, hw[0], hw[1])
        r = enc.compress(x, qp, reset, pb, pr)
        d = dec.decompress(r["bit_stream"], sps, qp, r["ec_parallel"], reset)["x_hat"].clone()
        want.append((r["bit_stream"], d))
    for i in (2, 3):
        qp, reset = plan[i]
        r = enc2.compress(xs[i], qp, reset, pb, pr)
        d = dec2.decompress(r["bit_stream"], sps, qp, r["ec_parallel"], reset)["x_hat"]
        torch.cuda.synchronize()
        assert r["bit_stream"] == want[i][0], "picture %d after the hand-off" % i
        assert torch.equal(d, want[i][1])
    with pytest.raises(Exception, match="picture size"):
        enc2.proxy.import_state(s_enc, 64, 64)


def test_null_stream_consumer_sees_finished_results():
    """A caller on torch's default (= legacy null) stream - plain torch code; the reference harness
    installs a non-default one, test_video.py:423-425. Results
    are joined to it through a blocking stream instead of hipStreamWaitEvent(null stream)
    (codec_base.hip, CodecBase::leave): a consumer queued on the null stream right behind
    decompress - no host synchronisation in between - must read the finished picture."""
    m = dmc_ld_model(skip_thres=0.15)
    hw = (720, 1280)
    ref = to_device_input(_padded(picture(*hw, index=0)))
    enc, dec = _gpu_net(m), _gpu_net(m)
    pb, pr = _pads(enc, *hw)
    enc.add_ref_feature_from_frame(ref)
    dec.add_ref_feature_from_frame(ref, apply_feature_adaptor=False)
    sps = {"height": hw[0], "width": hw[1]}
    assert torch.cuda.current_stream().cuda_stream == 0
    for i in range(6):
        x = to_device_input(picture(*hw, index=1 + i))
        r = enc.compress(x, 30 + i, 0, pb, pr)
        d = dec.decompress(r["bit_stream"], sps, 30 + i, r["ec_parallel"], 0)["x_hat"]
        seen = d.clone()                           # queued on the null stream, no host sync
        torch.cuda.synchronize()
        assert torch.equal(seen, d), "picture %d: the consumer read an unfinished reconstruction" % i
