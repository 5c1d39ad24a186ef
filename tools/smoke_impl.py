"""smoke(): one small invocation of the hot path on cuda:0, checked against the oracle:
DMCI compress + decompress of a 64x64 picture through the reference's plugin surface; the rANS
bytes and the reconstruction must equal the CPU oracle's bit for bit."""
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
