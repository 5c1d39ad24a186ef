"""Generates tests/golden/io_golden.npz from the REFERENCE's host-side helpers (run in the build
container, where /root/reference exists):

  * container bytes: src/utils/stream_helper.py write_sps / write_ip for a set of header values
    and payload lengths that cross every varuint size class;
  * picture loading: transforms.ycbcr420_to_444_np + the tensor ops of test_video.py:113-122
    (get_src_frame) on the CPU;
  * picture writing / distortion planes: the tensor ops of test_video.py:32-45 and :356-363.

Usage: python tests/golden/make_io_golden.py
"""
import io
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
from src.utils import stream_helper as ref_sh  # noqa: E402
from src.utils.transforms import ycbcr420_to_444_np, yuv_444_to_420  # noqa: E402

IP_CASES = [(True, 0, 0, 1, 0, 0), (False, 3, 63, 8, 1, 100), (False, 15, 255, 127, 0, 127),
            (True, 1, 32, 2, 0, 128), (False, 2, 17, 5, 1, 16383), (True, 0, 40, 8, 0, 16384),
            (False, 7, 9, 3, 1, 20000)]
SPS_CASES = [(0, 1080, 1920), (5, 64, 64), (15, 2160, 3840), (1, 100, 20000)]


def main():
    rng = np.random.default_rng(0)
    out = {}
    buf = io.BytesIO()
    for sid, h, w in SPS_CASES:
        ref_sh.write_sps(buf, {"sps_id": sid, "height": h, "width": w})
    payloads = []
    for is_i, sid, qp, ec, reset, n in IP_CASES:
        p = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        payloads.append(p)
        ref_sh.write_ip(buf, is_i, sid, qp, ec, reset, p)
    out["container"] = np.frombuffer(buf.getvalue(), dtype=np.uint8)
    out["ip_cases"] = np.array(IP_CASES, dtype=np.int64)
    out["sps_cases"] = np.array(SPS_CASES, dtype=np.int64)

    H, W = 36, 52
    y = rng.integers(0, 256, (1, H, W), dtype=np.uint8)
    y[0, :4, :64 if W >= 64 else W] = np.arange(4 * W, dtype=np.int64).reshape(4, W) % 256   # every value occurs
    uv = rng.integers(0, 256, (2, H // 2, W // 2), dtype=np.uint8)
    yuv = ycbcr420_to_444_np(y, uv)
    x = torch.from_numpy(yuv).unsqueeze(0).half()
    x = x / 255.0
    x = x - 0.5
    out["y"], out["uv"] = y[0], uv
    out["x"] = x[0].permute(1, 2, 0).numpy()

    x_hat = (torch.from_numpy(rng.uniform(-0.52, 0.52, (1, 3, 48, 64)).astype(np.float32))).half()
    crop = x_hat[:, :, :H, :W]
    y_rec, uv_rec = yuv_444_to_420(crop + 0.5)
    out["x_hat"] = x_hat[0].permute(1, 2, 0).numpy()
    out["y16"] = torch.clamp(y_rec * 255, 0, 255)[0, 0].numpy()
    out["uv16"] = torch.clamp(uv_rec * 255, 0, 255)[0].numpy()
    out["y8"] = torch.clamp(y_rec * 255, 0, 255).round().byte()[0, 0].numpy()
    out["uv8"] = torch.clamp(uv_rec * 255, 0, 255).byte()[0].numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "io_golden.npz"), **out)
    print("wrote io_golden.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
