"""numpy restatement of the harness's picture I/O (TEST INFRASTRUCTURE ONLY):

  yuv420_to_x   test_video.py:69-123 get_src_frame + transforms.py:69-80 ycbcr420_to_444_np
                (order 0 = nearest), then x.half() / 255.0 - 0.5 with one fp16 rounding per op
  x_to_yuv420   test_video.py:32-45 get_distortion (fp16 planes scaled to 0..255) and :356-363
                (the writer: Y .round() = half to even, U/V .byte() = truncation)
"""
import numpy as np

F16 = np.float16


def yuv420_to_x(y, uv):
    """y u8 [H, W], uv u8 [2, H/2, W/2] -> fp16 [H, W, 3]."""
    uv_up = np.repeat(np.repeat(uv, 2, axis=1), 2, axis=2)
    yuv = np.concatenate([y[None], uv_up], axis=0).astype(F16)
    x = (yuv.astype(np.float32) / np.float32(255.0)).astype(F16)
    x = (x.astype(np.float32) - np.float32(0.5)).astype(F16)
    return np.ascontiguousarray(x.transpose(1, 2, 0))


def x_to_yuv420(x_hat, height, width):
    """x_hat fp16 [Hp, Wp, 3] -> dict(y16, uv16 fp16 0..255; y8, uv8 uint8)."""
    t = (x_hat[:height, :width].astype(np.float32) + np.float32(0.5)).astype(F16)

    def scale(p):
        v = (p.astype(np.float32) * np.float32(255.0)).astype(F16)
        return np.clip(v, F16(0), F16(255)).astype(F16)

    y16 = scale(t[..., 0])
    c = t[..., 1:].astype(np.float32)
    pooled = (c[0::2, 0::2] + c[0::2, 1::2] + c[1::2, 0::2] + c[1::2, 1::2]) * np.float32(0.25)
    uv16 = scale(pooled.astype(F16).transpose(2, 0, 1))
    return dict(y16=y16, uv16=uv16, y8=np.rint(y16.astype(np.float32)).astype(np.uint8),
                uv8=uv16.astype(np.float32).astype(np.uint8))
