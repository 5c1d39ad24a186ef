"""smoke(): one tiny invocation of the hot path on cuda:0, checked against the oracle."""
import ctypes

import numpy as np
import torch


def run():
    from dcvc_amd import _lib
    # fused 1x1 conv + bias + WSiLU on the matrix cores vs fp32 torch
    vp, ci = ctypes.c_void_p, ctypes.c_int
    conv = _lib.fn("dcvc_conv1x1", ci, [vp, ci, vp, vp, vp, ci, vp, ci, vp, vp, vp, ci, ci, ci, ci, ci, vp])
    g = torch.Generator().manual_seed(0)
    P, K, N = 300, 128, 128
    x = torch.randn((P, K), generator=g).half().cuda()
    w = (torch.randn((N, K), generator=g) / K ** 0.5).half().cuda()
    b = torch.randn((N,), generator=g).half().cuda()
    y = torch.zeros((P, N), dtype=torch.half, device="cuda")
    _lib.check(conv(x.data_ptr(), K, w.data_ptr(), b.data_ptr(), None, 0, None, 0, None, None,
                    y.data_ptr(), N, P, K, N, 1, None))
    torch.cuda.synchronize()
    acc = x.float() @ w.float().t() + b.float()
    want = (acc * torch.sigmoid(4 * acc)).half()
    err = (y.float() - want.float()).abs().max().item()
    assert err < 5e-3, err
    print("smoke ok: conv1x1+bias+wsilu max abs err %.2e" % err)
