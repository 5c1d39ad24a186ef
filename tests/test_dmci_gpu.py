"""DMCI codec parity on a real MI355X (-m gpu), through the reference's plugin surface
(inference_extensions_cuda.DMCIProxy over the C ABI).

The bar (integer/byte work and, thanks to the measured matrix-core arithmetic, the whole path):
bit-exact against the CPU oracle - identical rANS bytes, identical reconstruction."""
import numpy as np
import pytest
import torch

from codec_util import dmci_model, from_device_output, oracle_for, picture, psnr, to_device_input

pytestmark = pytest.mark.gpu


def _gpu_net(skip_thres):
    m = dmci_model(skip_thres=skip_thres)
    # finalize_model (test_video.py:27-29): half, on the device, channels_last
    import copy
    g = copy.deepcopy(m).half().cuda()
    g.proxy = None
    return m, g


@pytest.mark.parametrize("hw,qp,thres", [((64, 64), 32, 0.15), ((70, 100), 7, 0.0), ((128, 96), 60, 0.15)])
def test_bitstream_and_reconstruction_match_oracle(hw, qp, thres):
    m, g = _gpu_net(thres)
    o = oracle_for(m)
    x = picture(*hw)
    want = o.compress(x, qp)
    pr, pb = g.get_padding_size(hw[0], hw[1], 16)
    got = g.compress(to_device_input(x), qp, pb, pr)
    torch.cuda.synchronize()
    x_hat = from_device_output(got["x_hat"])
    # intermediate tensors first: they localise a failure
    y_gpu = g.proxy.debug_read("y", np.float16).reshape(o.debug["y"].shape)
    print("y mismatches:", int((y_gpu != o.debug["y"]).sum()), "of", y_gpu.size)
    assert np.array_equal(g.proxy.debug_read("z_i8", np.int8), o.debug["z_i8"].reshape(-1))
    assert np.array_equal(y_gpu, o.debug["y"])
    assert got["ec_parallel"] == want["ec_parallel"]
    assert got["bit_stream"] == want["bit_stream"], "rANS bitstream differs from the oracle's"
    assert np.array_equal(x_hat, want["x_hat"]), "reconstruction differs from the oracle's"
    # decode on the GPU: closure, and equal to the oracle decoding the same bytes
    dec = g.decompress(got["bit_stream"], {"height": hw[0], "width": hw[1]}, qp, got["ec_parallel"])
    torch.cuda.synchronize()
    assert np.array_equal(from_device_output(dec["x_hat"]), x_hat)
    print("bytes", len(got["bit_stream"]), "PSNR(x_hat, x) = %.2f dB" %
          psnr(x_hat[:hw[0], :hw[1]], x))


def test_graph_replay_equals_eager():
    """Stages replayed from hipGraphs produce the same bytes / pixels as eager launches, call after
    call (first call = eager warm-up, second = capture, third = replay) and across qp changes."""
    m, g = _gpu_net(0.15)
    x = to_device_input(picture(96, 160))
    pr, pb = g.get_padding_size(96, 160, 16)
    ref = {}
    g._ensure_proxy().set_use_graphs(False)
    for qp in (10, 50):
        r = g.compress(x, qp, pb, pr)
        torch.cuda.synchronize()
        ref[qp] = (r["bit_stream"], r["x_hat"].clone())
    g.proxy.set_use_graphs(True)
    for rep in range(3):
        for qp in (10, 50):
            r = g.compress(x, qp, pb, pr)
            torch.cuda.synchronize()
            assert r["bit_stream"] == ref[qp][0], (rep, qp)
            assert torch.equal(r["x_hat"], ref[qp][1]), (rep, qp)
            d = g.decompress(r["bit_stream"], {"height": 96, "width": 160}, qp, r["ec_parallel"])
            torch.cuda.synchronize()
            assert torch.equal(d["x_hat"], ref[qp][1]), (rep, qp)


@pytest.mark.parametrize("qp", [0, 32, 63])
def test_full_hd_closure(qp):
    """BASELINE config 2 size (1920x1080 YUV420): encode -> bytes -> decode reproduces the
    encoder's reconstruction exactly (size-independent property), several sub-streams in use."""
    m, g = _gpu_net(0.15)
    x = to_device_input(picture(1080, 1920, index=qp))
    pr, pb = g.get_padding_size(1080, 1920, 16)
    enc = g.compress(x, qp, pb, pr)
    torch.cuda.synchronize()
    x_hat = enc["x_hat"].clone()
    assert x_hat.shape == (1, 3, 1088, 1920)
    assert torch.isfinite(x_hat.float()).all() and x_hat.abs().max() <= 0.5
    dec = g.decompress(enc["bit_stream"], {"height": 1080, "width": 1920}, qp, enc["ec_parallel"])
    torch.cuda.synchronize()
    assert torch.equal(dec["x_hat"], x_hat)
    bpp = len(enc["bit_stream"]) * 8 / (1080 * 1920)
    print("qp %d: %d bytes (%.3f bpp), ec_parallel %d" % (qp, len(enc["bit_stream"]), bpp, enc["ec_parallel"]))
    assert enc["ec_parallel"] >= 1 and len(enc["bit_stream"]) > 1000


def test_corrupt_stream_does_not_crash():
    m, g = _gpu_net(0.15)
    x = to_device_input(picture(64, 64))
    enc = g.compress(x, 20, 0, 0)
    torch.cuda.synchronize()
    bad = bytearray(enc["bit_stream"])
    for i in range(8, len(bad), 7):
        bad[i] ^= 0x5a
    d = g.decompress(bytes(bad[:len(bad) // 2]), {"height": 64, "width": 64}, 20, enc["ec_parallel"])
    torch.cuda.synchronize()
    assert torch.isfinite(d["x_hat"].float()).all()

def test_null_stream_consumer_sees_finished_results():
    """compress() returns as soon as the bit stream is complete - the reconstruction tail is still
    running on the codec's stream. A consumer queued on torch's default (legacy null) stream right
    behind it, no host synchronisation, must read the finished picture (CodecBase::leave joins the
    results through a blocking stream; tests/test_dmcld_gpu.py has the decoder-side twin)."""
    m, g = _gpu_net(0.15)
    hw = (720, 1280)
    pr, pb = g.get_padding_size(hw[0], hw[1], 16)
    assert torch.cuda.current_stream().cuda_stream == 0
    for i in range(4):
        x = to_device_input(picture(*hw, index=i))
        r = g.compress(x, 20 + 10 * i, pb, pr)
        early = r["x_hat"].clone()
        d = g.decompress(r["bit_stream"], {"height": hw[0], "width": hw[1]}, 20 + 10 * i, r["ec_parallel"])["x_hat"]
        seen = d.clone()
        torch.cuda.synchronize()
        assert torch.equal(early, r["x_hat"]), "picture %d: unfinished encoder-side reconstruction" % i
        assert torch.equal(seen, d), "picture %d: unfinished decoder-side reconstruction" % i
        assert torch.equal(early, seen)
