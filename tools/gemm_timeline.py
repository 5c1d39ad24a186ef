"""In-kernel timeline of the contraction kernel (shader-clock stamps of wave 0 per workgroup).
Usage: python tools/gemm_timeline.py [P,K,N,flags ...]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

NAMES = ["entry", "prologue"] + ["k%d" % i for i in range(8)] + ["loop_done", "epi_math", "stores"]


def main():
    from dcvc_amd import _lib
    from gpu_util import Ops, call, ptr, stream
    ops = Ops()
    setbuf = _lib.fn("dcvc_gemm_timeline_buffer", ctypes.c_int, [ctypes.c_void_p])
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [
        (32640, 384, 1536, 3), (32640, 384, 384, 4), (32640, 384, 384, 1), (8160, 512, 2048, 3), (8160, 512, 512, 4)]
    for P, K, N, fl in shapes:
        x = torch.randn((P, K), device="cuda").half()
        w = (torch.randn((N, K), device="cuda") / K ** 0.5).half()
        b = torch.randn((N,), device="cuda").half()
        nout = N // 4 if fl & 2 else N
        r = torch.randn((P, nout), device="cuda").half() if fl & 4 else None
        y = torch.zeros((P, nout), device="cuda", dtype=torch.half)
        tl = torch.zeros((4096, 16), dtype=torch.int64, device="cuda")
        run = lambda: call(ops.conv1x1, ptr(x), K, ptr(w), ptr(b), ptr(r), nout, None, 0, None, None, ptr(y), nout, P, K, N, fl & 3, stream())
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        _lib.check(setbuf(ctypes.c_void_p(tl.data_ptr())))
        run()
        torch.cuda.synchronize()
        _lib.check(setbuf(None))
        t = tl.cpu().numpy()
        t = t[t[:, 0] != 0]
        nk = K // 64
        cols = [0, 1] + list(range(2, 2 + min(nk, 8))) + [10, 11, 12]
        t0 = t[:, 0].min()
        print("P=%d K=%d N=%d flags=%d: %d workgroups, kernel span %.0f cycles" % (P, K, N, fl, len(t), t[:, 12].max() - t0))
        d = np.diff(t[:, cols], axis=1)
        labels = [NAMES[c] for c in cols]
        print("   segment (median / p90 cycles): " + "  ".join(
            "%s->%s %.0f/%.0f" % (labels[i], labels[i + 1], np.median(d[:, i]), np.percentile(d[:, i], 90)) for i in range(d.shape[1])))
        print("   workgroup start offsets (cycles after first): median %.0f  p90 %.0f  max %.0f; duration median %.0f" % (
            np.median(t[:, 0] - t0), np.percentile(t[:, 0] - t0, 90), (t[:, 0] - t0).max(), np.median(t[:, 12] - t[:, 0])))


if __name__ == "__main__":
    main()
