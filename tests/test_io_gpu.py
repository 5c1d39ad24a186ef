"""Picture I/O kernels (frame_io.hip) on a real MI355X against the oracle / reference vectors."""
import ctypes
import os

import numpy as np
import pytest
import torch

from dcvc_amd import _lib

pytestmark = pytest.mark.gpu
vp, ci = ctypes.c_void_p, ctypes.c_int


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _load(y, uv, ldx=3, offset=0):
    fn = _lib.fn("dcvc_yuv420_to_x", ci, [vp, vp, ci, ci, vp, ci, vp])
    H, W = y.shape
    x = torch.zeros((H, W, ldx), dtype=torch.float16, device="cuda")
    yd, uvd = torch.from_numpy(y).cuda(), torch.from_numpy(uv).cuda()
    _lib.check(fn(_ptr(yd), _ptr(uvd), H, W, ctypes.c_void_p(x.data_ptr() + 2 * offset), ldx, _stream()))
    torch.cuda.synchronize()
    return x.cpu().numpy()


def _store(x_hat, H, W):
    fn = _lib.fn("dcvc_x_to_yuv420", ci, [vp, ci, ci, ci, vp, vp, vp, vp, vp])
    xd = torch.from_numpy(x_hat).cuda()
    y16 = torch.empty((H, W), dtype=torch.float16, device="cuda")
    uv16 = torch.empty((2, H // 2, W // 2), dtype=torch.float16, device="cuda")
    y8 = torch.empty((H, W), dtype=torch.uint8, device="cuda")
    uv8 = torch.empty((2, H // 2, W // 2), dtype=torch.uint8, device="cuda")
    _lib.check(fn(_ptr(xd), x_hat.shape[1], H, W, _ptr(y16), _ptr(uv16), _ptr(y8), _ptr(uv8), _stream()))
    torch.cuda.synchronize()
    return dict(y16=y16.cpu().numpy(), uv16=uv16.cpu().numpy(), y8=y8.cpu().numpy(), uv8=uv8.cpu().numpy())


def test_reference_vectors(golden_dir):
    g = np.load(os.path.join(golden_dir, "io_golden.npz"))
    assert np.array_equal(_load(g["y"], g["uv"]), g["x"])
    h, w = g["y"].shape
    r = _store(g["x_hat"], h, w)
    for k in ("y16", "uv16", "y8", "uv8"):
        assert np.array_equal(r[k], g[k]), k


def test_full_hd_against_oracle_and_chunk_layout():
    from oracle import frame_io
    rng = np.random.default_rng(1)
    H, W = 1080, 1920
    y = rng.integers(0, 256, (H, W), dtype=np.uint8)
    uv = rng.integers(0, 256, (2, H // 2, W // 2), dtype=np.uint8)
    want = frame_io.yuv420_to_x(y, uv)
    assert np.array_equal(_load(y, uv), want)
    # picture 5 of an 8-picture chunk: pixel stride 24, channel offset 15; the rest stays untouched
    got = _load(y, uv, ldx=24, offset=15)
    assert np.array_equal(got[..., 15:18], want) and not got[..., :15].any() and not got[..., 18:].any()
    x_hat = rng.uniform(-0.55, 0.55, (1088, 1920, 3)).astype(np.float16)
    r, w_ = _store(x_hat, H, W), frame_io.x_to_yuv420(x_hat, H, W)
    for k in ("y16", "uv16", "y8", "uv8"):
        assert np.array_equal(r[k], w_[k]), k
