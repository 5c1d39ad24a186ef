"""Launches per coded picture from a rocprofv3 kernel trace, set-up excluded.

  python tools/trace_after_setup.py <dir or *_kernel_trace.csv> [--marker KERNEL_SUBSTRING] [--per N]

`rocprofv3 --kernel-trace --stats` sums over the whole process: a codec object's set-up (one hipMalloc + fill per device
buffer, one upload per weight tensor: ~ 1 300 fillBufferAligned and ~ 2 200 copyBuffer launches for the objects of an LD
bench run) sits in the same table as the per-picture work, and the round-4 review read those two rows as ~ 36 copies per
picture. This tool cuts the trace at the first launch of a kernel that only a coded picture runs (`--marker`, default the
symbol kernels `mask_step_enc` / `y_step_enc`) and reports, for everything behind that point, launches per picture
(`--per`: marker launches per coded picture; 2 for LD / HT-S - two checkerboard steps -, 4 for the intra model and HT-L)
and microseconds per picture, by kernel name."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.search(r"(dcb_nsplit8?_kernel<[^>]*>|conv_gemm_kernel|dwconv3x3\w*kernel|dcb_tail_kernel|ffn_fused_kernel|__amd_rocclr_\w+)", name)
    if m:
        return m.group(1)
    m = re.search(r"(\w+_kernel)", name)
    return m.group(1) if m else name[:60]


def main():
    args = sys.argv[1:]
    marker, per = None, None
    if "--marker" in args:
        i = args.index("--marker")
        marker = args[i + 1]
        del args[i:i + 2]
    if "--per" in args:
        i = args.index("--per")
        per = float(args[i + 1])
        del args[i:i + 2]
    path = args[0]
    if os.path.isdir(path):
        found = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))
        if not found:
            raise SystemExit("no *kernel_trace.csv under " + path)
        path = found[0]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    markers = (marker,) if marker else ("mask_step_enc", "y_step_enc")
    first = next((i for i, r in enumerate(rows) if any(m in r[2] for m in markers)), None)
    if first is None:
        raise SystemExit("no launch of %s in the trace" % (markers,))
    before, after = rows[:first], rows[first:]
    n_marker = sum(1 for r in after if any(m in r[2] for m in markers))
    if per is None:
        per = 4.0 if any("y_step_enc" in r[2] for r in after) else 2.0
    pictures = n_marker / per
    agg = defaultdict(lambda: [0, 0])
    for s, e, name in after:
        a = agg[short(name)]
        a[0] += 1
        a[1] += e - s
    setup = defaultdict(int)
    for _, _, name in before:
        setup[short(name)] += 1
    print("%s: %d launches in front of the first coded picture (set-up), %d behind it = %.1f coded pictures (%d marker launches / %g)"
          % (os.path.basename(path), len(before), len(after), pictures, n_marker, per))
    print("set-up, top rows: " + ", ".join("%s %d" % kv for kv in sorted(setup.items(), key=lambda kv: -kv[1])[:4]))
    print("%-50s %12s %12s" % ("kernel (behind the set-up)", "launches/pic", "us/pic"))
    tot_n = tot_us = 0.0
    for name, (n, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-50s %12.2f %12.1f" % (name, n / pictures, ns / 1e3 / pictures))
        tot_n += n / pictures
        tot_us += ns / 1e3 / pictures
    print("%-50s %12.2f %12.1f" % ("all", tot_n, tot_us))


if __name__ == "__main__":
    main()
