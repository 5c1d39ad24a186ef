"""smoke(): one small invocation of the hot path on cuda:0, checked against the oracle: an I picture
(DMCI) followed by a P picture (DMC low-delay) of a 64x64 sequence through the reference's plugin
surface; rANS bytes and reconstructions must equal the CPU oracle's bit for bit."""
import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run():
    from codec_util import dmci_model, from_device_output, oracle_for, picture, to_device_input
    m = dmci_model(skip_thres=0.15)
    g = copy.deepcopy(m).half().cuda()
    g.proxy = None
    x = picture(64, 64)
    got = g.compress(to_device_input(x), 32, 0, 0)
    torch.cuda.synchronize()
    want = oracle_for(m).compress(x, 32)
    assert got["bit_stream"] == want["bit_stream"], "bit stream differs from the oracle"
    assert np.array_equal(from_device_output(got["x_hat"]), want["x_hat"]), "x_hat differs from the oracle"
    dec = g.decompress(got["bit_stream"], {"height": 64, "width": 64}, 32, got["ec_parallel"])
    torch.cuda.synchronize()
    assert torch.equal(dec["x_hat"], got["x_hat"])
    print("smoke ok: DMCI 64x64 qp32 -> %d bytes, bit-exact vs oracle, decode closure ok"
          % len(got["bit_stream"]))
    # P picture on the intra reconstruction (test_video.py:226-238)
    from codec_util import dmc_ld_model
    mp = dmc_ld_model(skip_thres=0.15)
    enc, dec_p = copy.deepcopy(mp).half().cuda(), copy.deepcopy(mp).half().cuda()
    enc.proxy = dec_p.proxy = None
    enc.add_ref_feature_from_frame(got["x_hat"])
    dec_p.add_ref_feature_from_frame(dec["x_hat"], apply_feature_adaptor=False)
    eo, do = oracle_for(mp), oracle_for(mp)
    eo.add_ref_feature_from_frame(want["x_hat"], True)
    do.add_ref_feature_from_frame(want["x_hat"], False)
    x1 = picture(64, 64, index=1)
    gp = enc.compress(to_device_input(x1), 30, 0, 0, 0)
    wp = eo.compress(x1, 30, False)
    assert gp["bit_stream"] == wp["bit_stream"], "P-picture bit stream differs from the oracle"
    xd = dec_p.decompress(gp["bit_stream"], {"height": 64, "width": 64}, 30, gp["ec_parallel"], 0)["x_hat"]
    torch.cuda.synchronize()
    assert np.array_equal(from_device_output(xd), do.decompress(wp["bit_stream"], 30, 64, 64, wp["ec_parallel"], False))
    print("smoke ok: DMC-LD P picture qp30 -> %d bytes, bit-exact vs oracle" % len(gp["bit_stream"]))
