"""BASELINE config 5 size (3840x2160 YUV420) on a real MI355X: the size-independent properties of
the 1080p tests (encode -> bytes -> decode closure, encoder / decoder lock-step of the temporal
state, finite in-range reconstructions) at 4K, for the intra, LD and HT-S codecs.

Part of the default `-m gpu` run (first run on hardware in round 2: gpurun session 1, all three pass; ~15 s)."""
import copy

import numpy as np
import pytest
import torch

from codec_util import dmc_ht_model, dmc_ld_model, dmci_model, picture, to_device_input

pytestmark = pytest.mark.gpu

H, W = 2160, 3840
H16, W16 = 2160, 3840           # both multiples of 16: no padding at 4K


def _gpu_net(model):
    g = copy.deepcopy(model).half().cuda()
    g.proxy = None
    return g


def _tile(index):
    """a 4K picture tiled from 1080p synthetic pictures (the generator is slow at 8 M pixels)"""
    a, b = picture(1080, 1920, index=2 * index), picture(1080, 1920, index=2 * index + 1)
    return np.concatenate([np.concatenate([a, b], axis=1), np.concatenate([b, a], axis=1)], axis=0)


def test_intra_closure_uhd():
    g = _gpu_net(dmci_model(skip_thres=0.15))
    x = to_device_input(_tile(0))
    for qp in (8, 50):
        enc = g.compress(x, qp, 0, 0)
        torch.cuda.synchronize()
        x_hat = enc["x_hat"].clone()
        assert x_hat.shape == (1, 3, H16, W16)
        assert torch.isfinite(x_hat.float()).all() and x_hat.abs().max() <= 0.5
        dec = g.decompress(enc["bit_stream"], {"height": H, "width": W}, qp, enc["ec_parallel"])
        torch.cuda.synchronize()
        assert torch.equal(dec["x_hat"], x_hat)
        assert enc["ec_parallel"] == 8 and len(enc["bit_stream"]) > 4000


def _gop(p_model, frames, plan):
    i_enc, p_enc, p_dec = _gpu_net(dmci_model(skip_thres=0.15)), _gpu_net(p_model), _gpu_net(p_model)
    sps = {"height": H, "width": W}
    e = i_enc.compress(to_device_input(_tile(0)), 30, 0, 0)
    p_enc.add_ref_feature_from_frame(e["x_hat"])
    p_dec.add_ref_feature_from_frame(e["x_hat"], apply_feature_adaptor=False)
    for i, (qp, reset) in enumerate(plan):
        x = to_device_input(np.concatenate([_tile(1 + i * frames + j) for j in range(frames)], axis=-1))
        r = p_enc.compress(x, qp, reset, 0, 0)
        xd = p_dec.decompress(r["bit_stream"], sps, qp, r["ec_parallel"], reset)["x_hat"]
        torch.cuda.synchronize()
        allx = torch.cat(xd, 0) if isinstance(xd, (list, tuple)) else xd
        assert allx.shape == (frames, 3, H16, W16)
        assert torch.isfinite(allx.float()).all() and allx.abs().max() <= 0.5
        fe = p_enc.proxy.debug_read("feature_p", np.float16)
        fd = p_dec.proxy.debug_read("feature_p", np.float16)
        assert np.array_equal(fe, fd), "decoder drifted from the encoder at call %d" % i
        assert len(r["bit_stream"]) > 1000


def test_ld_gop_closure_uhd():
    _gop(dmc_ld_model(skip_thres=0.15), 1, [(34, 0), (30, 1), (40, 0)])


def test_hts_chunk_closure_uhd():
    _gop(dmc_ht_model("hts", skip_thres=0.15), 8, [(34, 0)])


def test_htl_chunk_closure_uhd():
    """HT-L at 3840x2160 (round 4; DESIGN claimed this test one round before it existed): two chunks, the second with a
    memory reset - the 4-step entropy scheme of dmc_htl_proxy.cpp:624-689 on the 135 x 240 grid."""
    _gop(dmc_ht_model("htl", skip_thres=0.15), 8, [(34, 0), (46, 1)])
