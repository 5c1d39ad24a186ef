"""In-kernel timeline of dcb_core (shader-clock stamps of wave 0 per workgroup: entry, then one per
weight slab after its barrier) + wall time of the launch and of the launch sequence it replaces.
Usage: python tools/core_timeline.py [pixels]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from dcvc_amd import _lib
    from gpu_util import Ops, call, ptr, stream
    ops = Ops()
    setbuf = _lib.fn("dcvc_dcb_core_timeline_buffer", ctypes.c_int, [ctypes.c_void_p])
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 32640
    C = 384
    g = lambda *s: (torch.randn(s, device="cuda") * 0.5).half()
    x, t2 = g(P, C), g(P, C)
    w3, w2, w1 = [(torch.randn((C, C), device="cuda") / C ** 0.5).half() for _ in range(3)]
    w0 = (torch.randn((4 * C, C), device="cuda") / C ** 0.5).half()
    b3, b2, b1, b0 = g(C), g(C), g(C), g(4 * C)
    y = torch.zeros((P, C), device="cuda", dtype=torch.half)
    t1 = torch.zeros((P, C), device="cuda", dtype=torch.half)
    y1 = torch.zeros((P, C), device="cuda", dtype=torch.half)
    t = torch.zeros((P, C), device="cuda", dtype=torch.half)

    def core(nxt=True):
        call(ops.dcb_core, ptr(t2), C, ptr(x), C, ptr(w3), ptr(b3), ptr(w0), ptr(b0), ptr(w2), ptr(b2), None, None,
             ptr(w1) if nxt else None, ptr(b1) if nxt else None, ptr(t1) if nxt else None, C, ptr(y), C, P, C, 0, stream())

    def seq():
        call(ops.conv1x1, ptr(t2), C, ptr(w3), ptr(b3), ptr(x), C, None, 0, None, None, ptr(y1), C, P, C, C, 0, stream())
        call(ops.conv1x1, ptr(y1), C, ptr(w0), ptr(b0), None, 0, None, 0, None, None, ptr(t), C, P, C, 4 * C, 3, stream())
        call(ops.conv1x1, ptr(t), C, ptr(w2), ptr(b2), ptr(y1), C, None, 0, None, None, ptr(y), C, P, C, C, 0, stream())
        call(ops.conv1x1, ptr(y), C, ptr(w1), ptr(b1), None, 0, None, 0, None, None, ptr(t1), C, P, C, C, 1, stream())

    def wall(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n * 1e3

    flop = 2.0 * P * 7 * C * C
    for name, fn, f in (("dcb_core + next dc.0", core, flop), ("dcb_core", lambda: core(False), flop * 6 / 7),
                        ("4 conv1x1 launches", seq, flop)):
        us = wall(fn)
        print("%-24s %7.1f us  %6.0f TFLOP/s" % (name, us, f / us / 1e6))
    tl = torch.zeros((1024, 64), dtype=torch.int64, device="cuda")
    _lib.check(setbuf(ctypes.c_void_p(tl.data_ptr())))
    core()
    torch.cuda.synchronize()
    _lib.check(setbuf(None))
    s = tl.cpu().numpy()
    s = s[s[:, 0] != 0]
    t0 = s[:, 0].min()
    d = np.diff(s, axis=1)
    print("%d workgroups; entry skew median %.0f max %.0f cycles" % (len(s), np.median(s[:, 0] - t0), (s[:, 0] - t0).max()))
    names = ["first slab landed", "dc.3 (18 slabs)", "y1 epilogue", "ffn super-chunk 0 (12 slabs)"] + \
            ["ffn super-chunk %d (15 slabs)" % i for i in range(1, 6)] + \
            ["last pair epilogue + ffn.2 (3 slabs)", "y epilogue + stores", "next dc.0 (18 slabs) + drain"]
    fine = s[:, 16:40]
    s = s[:, :12]
    d = np.diff(s, axis=1)
    for i in range(d.shape[1]):
        print("   %-40s %7.0f cycles  (p90 %7.0f)" % (names[i + 1] if i + 1 < len(names) else "?", np.median(d[:, i]), np.percentile(d[:, i], 90)))
    print("   total %.0f cycles" % np.median(s[:, 11] - s[:, 0]))
    if fine.any():
        labels = ["wait + barrier + plan", "slice 0", "slice 1", "slice 2", "slice 3"]
        f = np.diff(fine[:, 6:12], axis=1) % (1 << 32)
        print("   the flush of that step: 4 ds_write %4.0f | wait %4.0f | 4 ds_read issue %4.0f | waits + stores %4.0f | final wait %4.0f"
              % tuple(np.median(f[:, i]) for i in range(5)))
        for name, base in (("dc.3 slab 7", 0), ("next dc.0, pair 1, k-third 2 (flush; slices in twos)", 18)):
            f = np.diff(fine[:, base:base + 6], axis=1) % (1 << 32)
            print("   %-38s %s   = %d" % (name, "  ".join("%s %4.0f" % (l, np.median(f[:, i])) for i, l in enumerate(labels)),
                                           np.median((fine[:, base + 5] - fine[:, base]) % (1 << 32))))

if __name__ == "__main__":
    main()
