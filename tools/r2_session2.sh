#!/bin/bash
# Round 2, GPU session 2: first run of dcb_core (fused full-width DepthConvBlock) + the oracle digests at 720p / 1080p
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "dcb_core" 2>&1 | tail -15 ) > gpurun_out/s2_test_core.log
tail -5 gpurun_out/s2_test_core.log
( timeout 600 python -m pytest tests/test_fullsize_gpu.py tests/test_dmci_gpu.py -m gpu -q 2>&1 | tail -25 ) > gpurun_out/s2_test_codec.log
tail -12 gpurun_out/s2_test_codec.log
timeout 200 python bench.py --no-cpu-baseline > gpurun_out/s2_bench_core.json 2> gpurun_out/s2_bench_core.err
DCVC_NO_DCB_CORE=1 timeout 200 python bench.py --no-cpu-baseline > gpurun_out/s2_bench_nocore.json 2> gpurun_out/s2_bench_nocore.err
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/s2_prof -o s2 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/s2_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import glob, json, csv
for f in sorted(glob.glob("gpurun_out/s2_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-40s %8.1f pictures/s  %.2f ms/step gemm %.0f TFLOP/s" % (f, d["value"], d["ms_per_step"], d.get("roofline", {}).get("achieved", 0)))
    except Exception as e:
        print(f, "unreadable:", e)
for f in glob.glob("gpurun_out/s2_prof/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print("%-90s calls %6s avg %10.1f us  %5.1f%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
