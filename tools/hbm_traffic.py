"""HBM bytes per launch of the codec's kernels from two rocprofv3 PMC passes of the bench command
(FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950: MI355X_MICROARCH.md, PMC slots).
FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B, same guide); units are KB.
Usage: python tools/hbm_traffic.py <fetch_dir> <write_dir> <out.json> [commit]"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

FAMILIES = ("conv_gemm_kernel", "dwconv3x3", "dcb_tail_kernel", "ffn_fused_kernel", "dcb_pair8_kernel")
# (the fourth template argument - what the NEXT slot holds - was a bool until round 5 and is an int since round 6; a fifth - the depthwise
# conv inside the launch - came with round 6's last session)
NSPLIT = re.compile(r"dcb_nsplit(8?)_kernel<(\d+), (\d+), (\d+), (true|false|\d+)(?:, \d+)?>")
NSPLIT_MANGLED = re.compile(r"dcb_nsplit(8?)_kernelILi(\d+)ELi(\d+)ELi(\d+)EL[bi](\d+)E(?:Li\d+E)?")


def family(name):
    """the key bench.py's roofline uses for the kernel: the N-split block kernel per shape <C, CI, pixels per workgroup>
    (launches with and without the next block's dc.0 together), the others by name"""
    m = NSPLIT.search(name) or NSPLIT_MANGLED.search(name)
    if m:
        return "dcb_nsplit%s_kernel<%s, %s, %d px>" % (m.group(1), m.group(2), m.group(3), 32 * int(m.group(4)))
    for f in FAMILIES:
        if f in name:
            return "dwconv3x3_kernel" if f == "dwconv3x3" else f        # (every instantiation of the depthwise walk)
    return None


def collect(root, counter):
    agg = defaultdict(lambda: [0.0, 0])
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                fam = family(row.get("Kernel_Name", ""))
                if fam is None:
                    continue
                a = agg[fam]
                a[0] += float(row.get("Counter_Value", 0) or 0)
                a[1] += 1
    return agg


def main():
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    import bench
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE --kernel-trace -- python bench.py --steps 5 --warmup 3 "
                     "--no-cpu-baseline --no-roofline --no-extras --no-uhd (separate passes); FETCH_SIZE doubled per "
                     "MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B), counter units KB",
           "commit": sys.argv[4] if len(sys.argv) > 4 else None,
           # bench.py reports these numbers only while the kernel sources are the ones measured here
           "kernel_source_digest": bench.kernel_source_digest(), "kernels": {}}
    for fam in sorted(set(fetch) | set(write)):
        f, w = fetch.get(fam, [0.0, 0]), write.get(fam, [0.0, 0])
        fb = 2.0 * 1024.0 * f[0] / max(f[1], 1)
        wb = 1024.0 * w[0] / max(w[1], 1)
        out["kernels"][fam] = {"launches": max(f[1], w[1]), "fetch_bytes_corrected_avg": fb, "write_bytes_avg": wb,
                               "hbm_bytes_per_launch": fb + wb}
    with open(sys.argv[3], "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
