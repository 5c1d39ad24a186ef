"""Micro-benchmark of the contraction kernel on the shapes of the DMCI hot path (for rocprofv3).
Usage: python tools/gemm_bench.py [shape ...]   shape = P,K,N,flags  (flags: 1 wsilu, 2 chunk, 4 residual)"""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from gpu_util import Ops, call, ptr, stream
    ops = Ops()
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [
        (32640, 384, 1536, 3), (32640, 384, 384, 4), (32640, 384, 384, 1), (8160, 512, 2048, 3), (8160, 512, 512, 4)]
    reps = int(os.environ.get("REPS", "20"))
    for P, K, N, fl in shapes:
        x = torch.randn((P, K), device="cuda").half()
        w = (torch.randn((N, K), device="cuda") / K ** 0.5).half()
        b = torch.randn((N,), device="cuda").half()
        nout = N // 4 if fl & 2 else N
        r = torch.randn((P, nout), device="cuda").half() if fl & 4 else None
        y = torch.zeros((P, nout), device="cuda", dtype=torch.half)
        for _ in range(3):
            call(ops.conv1x1, ptr(x), K, ptr(w), ptr(b), ptr(r), nout, None, 0, None, None, ptr(y), nout, P, K, N, fl & 3, stream())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            call(ops.conv1x1, ptr(x), K, ptr(w), ptr(b), ptr(r), nout, None, 0, None, None, ptr(y), nout, P, K, N, fl & 3, stream())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print("P=%d K=%d N=%d flags=%d: %.2f us  %.1f TFLOP/s" % (P, K, N, fl, dt * 1e6, 2.0 * P * K * N / dt / 1e12), flush=True)


if __name__ == "__main__":
    main()
