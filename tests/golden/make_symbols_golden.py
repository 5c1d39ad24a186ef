"""Golden vectors for oracle/symbols_np.py from the reference's OWN Python restatement of its symbol kernels:
DCVC-RT's PyTorch fallbacks, /root/reference/DCVC-family/DCVC-RT/src/layers/cuda_inference.py
  process_with_mask   :58-74     build_index_dec :124-143     build_index_enc :146-171
  restore_y_4x        :113-119   round_and_to_int8 :26-33
(SURVEY section 8c names them as the previous generation's statement of the same dataflow.) They are run here on CPU
in fp16 - one rounding per op, as the DCVC-UF kernels have it - on inputs WITHOUT rounding ties (torch.round is half
to even, the DCVC-UF kernels round half away from zero: stream.cu:549-630) and with the DCVC-UF constants
(def_const.h:6-12) passed in as their fp16 values. Run in the build container (needs /root/reference):

    python tests/golden/make_symbols_golden.py        -> tests/golden/symbols_rt_golden.npz
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import symbols_np as sym  # noqa: E402  (constants and the mask layout only)

os.environ["SUPPRESS_CUSTOM_KERNEL_WARNING"] = "1"
spec = importlib.util.spec_from_file_location(
    "rt_cuda_inference", "/root/reference/DCVC-family/DCVC-RT/src/layers/cuda_inference.py")
rt = importlib.util.module_from_spec(spec)
spec.loader.exec_module(rt)
assert not rt.CUSTOMIZED_CUDA_INFERENCE


def nchw(a):          # [H, W, C] numpy -> [1, C, H, W] torch
    return torch.from_numpy(np.ascontiguousarray(a)).permute(2, 0, 1)[None].contiguous()


def nhwc(t):
    return t[0].permute(1, 2, 0).contiguous().numpy()


def main():
    rng = np.random.default_rng(20260924)
    H, W, C = 10, 12, 32
    F16 = np.float16
    out = {}
    for case, (spread, thres) in enumerate([(3.0, 0.15), (40.0, 0.15), (200.0, 0.0), (1.0, 0.3)]):
        y = (rng.standard_normal((H, W, C)) * spread).astype(F16)
        means = (rng.standard_normal((H, W, C)) * spread * 0.5).astype(F16)
        scales = np.exp(rng.uniform(-3.0, 3.2, (H, W, C))).astype(F16)
        scales[rng.random((H, W, C)) < 0.05] = F16(0.05)
        scales[rng.random((H, W, C)) < 0.02] = F16(40.0)
        while True:                                                           # redraw rounding ties
            res = (y - means).astype(F16).astype(np.float32)
            tie = np.abs(res - np.trunc(res)) == 0.5
            if not tie.any():
                break
            y = np.where(tie, (rng.standard_normal((H, W, C)) * spread).astype(F16), y)
        masks = sym.get_mask_4x(H, W, C)
        out["y%d" % case], out["means%d" % case], out["scales%d" % case] = y, means, scales
        out["thres%d" % case] = np.float32(thres)
        for k, mask in enumerate(masks):
            m = nchw(mask.astype(F16))
            thres_h = float(F16(np.float32(thres)))
            y_res, y_q, y_hat, s_hat = rt.process_with_mask(nchw(y), nchw(scales), nchw(means), m, thres_h)
            tag = "%d_%d" % (case, k)
            out["y_q" + tag], out["y_hat" + tag], out["s_hat" + tag] = nhwc(y_q), nhwc(y_hat), nhwc(s_hat)
            consts = (float(F16(sym.SCALE_MIN)), float(F16(sym.SCALE_MAX)), float(F16(sym.LOG_SCALE_MIN)),
                      float(F16(sym.LOG_SCALE_STEP_RECIP)))
            # the index kernels see the FOLDED tensors (one valid channel group per position, dmci_proxy.cpp:339-369):
            # x1 + x2 + x3 + x4 in fp16, NHWC order
            fold = lambda t: sum(t.permute(0, 2, 3, 1).contiguous().chunk(4, -1)[1:], t.permute(0, 2, 3, 1).contiguous().chunk(4, -1)[0])
            s_w, y_w = fold(s_hat), fold(y_q)
            idx, keep = rt.build_index_dec(s_w.clone(), *consts, skip_thres=thres_h)
            out["idx" + tag], out["keep" + tag] = idx[0].numpy(), keep[0].numpy()
            comb = rt.build_index_enc(y_w, s_w.clone(), *consts, skip_thres=thres_h)
            out["comb" + tag] = comb.numpy()                      # compacted in NHWC order (stream.cu:96-97)
            if k == 0:
                folded = sym.fold4(nhwc(y_q))
                out["restored" + tag] = nhwc(rt.restore_y_4x(nchw(folded), nchw(means), m))
        z = (rng.standard_normal((H, W, C)) * spread).astype(F16)
        zf = z.astype(np.float32)
        z = np.where(np.abs(zf - np.trunc(zf)) == 0.5, (zf + 0.25).astype(F16), z)
        z = np.clip(z, -63.4, 62.4).astype(F16)                   # DCVC-UF clamps z to [-64, 63], DCVC-RT to int8
        z_hat, z_i8 = rt.round_and_to_int8(nchw(z))
        out["z%d" % case], out["z_hat%d" % case], out["z_i8%d" % case] = z, nhwc(z_hat), nhwc(z_i8)
    path = os.path.join(HERE, "symbols_rt_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
