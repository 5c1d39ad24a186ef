"""Fused FFN (one launch) against the two-launch sequence on the DCB shapes. Usage: python tools/ffn_bench.py"""
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
    reps = 20
    for P, C, CF in [(32640, 384, 384), (32640, 256, 128), (32640, 128, 64), (8160, 384, 384), (130560, 384, 384)]:
        x = torch.randn((P, C), device="cuda").half()
        w0 = (torch.randn((4 * CF, C), device="cuda") / C ** 0.5).half()
        b0 = torch.randn((4 * CF,), device="cuda").half()
        w2 = (torch.randn((C, CF), device="cuda") / CF ** 0.5).half()
        b2 = torch.randn((C,), device="cuda").half()
        t = torch.zeros((P, CF), device="cuda", dtype=torch.half)
        y = torch.zeros((P, C), device="cuda", dtype=torch.half)

        def two():
            call(ops.conv1x1, ptr(x), C, ptr(w0), ptr(b0), None, 0, None, 0, None, None, ptr(t), CF, P, C, 4 * CF, 3, stream())
            call(ops.conv1x1, ptr(t), CF, ptr(w2), ptr(b2), ptr(x), C, None, 0, None, None, ptr(y), C, P, CF, C, 0, stream())

        def one():
            call(ops.ffn_fused, ptr(x), C, ptr(w0), ptr(b0), ptr(w2), ptr(b2), None, 0, None, None, ptr(y), C, P, C, CF, stream())

        res = {}
        for name, fn in (("two launches", two), ("fused", one), ("two launches", two), ("fused", one)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            res.setdefault(name, []).append((time.perf_counter() - t0) / reps * 1e6)
        fl = 2.0 * P * C * 5 * CF
        print("P=%d C=%d CF=%d: two launches %.1f us, fused %.1f us (%.0f TFLOP/s)" % (
            P, C, CF, min(res["two launches"]), min(res["fused"]), fl / min(res["fused"]) / 1e6), flush=True)


if __name__ == "__main__":
    main()
