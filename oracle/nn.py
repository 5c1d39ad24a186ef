"""ctypes front-end of oracle/nn_oracle.c plus the numpy glue (im2col, pixel shuffles) that turns
every dense operator of the path into the one bit-exact contraction routine.
TEST INFRASTRUCTURE ONLY.

Tensors are NHWC numpy float16 arrays [H, W, C] (or [P, C])."""
import ctypes
import os

import numpy as np

from oracle import rans as _r   # shares the liboracle loader

F16 = np.float16
WSILU, CHUNK_ADD = 1, 2
_vp = ctypes.c_void_p


def _lib():
    lib = _r.liboracle()
    lib.orc_mfma16.restype = ctypes.c_float
    lib.orc_mfma16.argtypes = [ctypes.c_float, _vp, _vp]
    return lib


def _p(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


def _h(a):
    return None if a is None else np.ascontiguousarray(a, dtype=F16)


def mfma16(c, a16, b16):
    a16, b16 = _h(a16), _h(b16)
    return np.float32(_lib().orc_mfma16(ctypes.c_float(float(c)), _p(a16), _p(b16)))


def conv1x1(x, w, bias=None, r1=None, r2=None, q=None, q2=None, wsilu=False, chunk_add=False):
    """x [..., K] fp16, w [N, K]; returns [..., N] (or N/4 with chunk_add) fp16."""
    lead = x.shape[:-1]
    K = x.shape[-1]
    x2 = _h(x.reshape(-1, K))
    w2 = _h(w.reshape(w.shape[0], -1))
    N = w2.shape[0]
    assert w2.shape[1] == K and K % 16 == 0
    P = x2.shape[0]
    nout = N // 4 if chunk_add else N
    y = np.empty((P, nout), dtype=F16)
    bias, q, q2 = _h(bias), _h(q), _h(q2)
    r1 = None if r1 is None else _h(r1.reshape(P, nout))
    r2 = None if r2 is None else _h(r2.reshape(P, nout))
    flags = (WSILU if wsilu else 0) | (CHUNK_ADD if chunk_add else 0)
    _lib().orc_conv1x1(_p(x2), K, _p(w2), _p(bias), _p(r1), nout, _p(r2), nout, _p(q), _p(q2),
                       _p(y), nout, P, K, N, flags)
    return y.reshape(lead + (nout,))


def conv_kxk(x, w, bias, ksize, stride, pad):
    """x [H, W, Cin]; w in PyTorch layout [Cout, Cin, k, k]; contraction index = (ky, kx, cin) -
    the order the HIP implicit GEMM walks (dcvc_amd/csrc/kernels/conv_gemm.hip)."""
    H, W, C = x.shape
    Ho = (H + 2 * pad - ksize) // stride + 1
    Wo = (W + 2 * pad - ksize) // stride + 1
    xp = np.zeros((H + 2 * pad, W + 2 * pad, C), dtype=F16)
    xp[pad:pad + H, pad:pad + W] = x
    cols = np.empty((Ho, Wo, ksize * ksize * C), dtype=F16)
    for ky in range(ksize):
        for kx in range(ksize):
            t = ky * ksize + kx
            cols[:, :, t * C:(t + 1) * C] = xp[ky:ky + stride * Ho:stride, kx:kx + stride * Wo:stride]
    wt = np.ascontiguousarray(np.transpose(w, (0, 2, 3, 1))).reshape(w.shape[0], -1)
    return conv1x1(cols, wt, bias)


def subpel_conv1x1(x, w):
    """SubpelConv2x with kernel 1 and no bias (layers.py:92-103) = 1x1 conv to 4*Cout channels +
    pixel_shuffle(2); the product runs it as a 2x2 stride-2 transposed conv, one contraction per
    output phase (dy, dx) - identical arithmetic."""
    H, W, _ = x.shape
    N4 = w.shape[0]
    cout = N4 // 4
    y = conv1x1(x, w.reshape(N4, -1))            # [H, W, 4*cout], channel = co*4 + dy*2 + dx
    y = y.reshape(H, W, cout, 2, 2)
    return np.ascontiguousarray(np.transpose(y, (0, 3, 1, 4, 2))).reshape(2 * H, 2 * W, cout)


def pixel_shuffle(x, r):
    H, W, C = x.shape
    c = C // (r * r)
    y = x.reshape(H, W, c, r, r)
    return np.ascontiguousarray(np.transpose(y, (0, 3, 1, 4, 2))).reshape(H * r, W * r, c)


def pixel_unshuffle(x, r):
    H, W, C = x.shape
    y = x.reshape(H // r, r, W // r, r, C)
    return np.ascontiguousarray(np.transpose(y, (0, 2, 4, 1, 3))).reshape(H // r, W // r, C * r * r)


def dwconv3x3(x, w):
    """x [H, W, C]; w PyTorch depthwise weight [C, 1, 3, 3]."""
    H, W, C = x.shape
    x = _h(x)
    wt = _h(np.transpose(w[:, 0], (1, 2, 0)).reshape(9, C))
    y = np.empty((H, W, C), dtype=F16)
    _lib().orc_dwconv3x3(_p(x), C, _p(wt), _p(y), C, H, W, C)
    return y


def fold_dw_bias(w3, b2, b3):
    """bias of dc.3 with the depthwise bias folded through it (layers_proxy.cpp:175-178)."""
    w3 = _h(w3.reshape(w3.shape[0], -1))
    out = np.empty(w3.shape[0], dtype=F16)
    _lib().orc_fold_dw_bias(_p(w3), _p(_h(b2)), _p(_h(b3)), _p(out), w3.shape[0], w3.shape[1])
    return out


def mul_channel(x, q):
    return (x * q.astype(F16)).astype(F16)
