"""Aggregates rocprofv3 --pmc CSV output (counter_collection.csv) per kernel name x grid size:
mean counter value per dispatch. Usage: python tools/pmc_summary.py <dir>"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    if "conv_gemm_kernel" in name:
        i = name.index("conv_gemm_kernel")
        return "conv_gemm" + name[i + len("conv_gemm_kernel"):].split("EEvNS")[0][:60]
    return name[:60]


def main():
    root = sys.argv[1]
    for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
        with open(path) as f:
            for row in csv.DictReader(f):
                key = (short(row.get("Kernel_Name", "?")), row.get("Grid_Size", "?"))
                a = agg[key][row.get("Counter_Name", "?")]
                a[0] += float(row.get("Counter_Value", 0) or 0)
                a[1] += 1
        print("==", os.path.relpath(path, root))
        for key, counters in sorted(agg.items(), key=lambda kv: -max(v[0] for v in kv[1].values())):
            n = max(v[1] for v in counters.values())
            if n < 2 and "bench" not in path:
                continue
            vals = "  ".join("%s=%.4g" % (c, v[0] / max(v[1], 1)) for c, v in sorted(counters.items()))
            print("  %-70s grid=%-9s n=%-5d %s" % (key[0], key[1], n, vals))


if __name__ == "__main__":
    main()
