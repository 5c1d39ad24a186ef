#!/bin/bash
# Round 2, GPU session 1: what round 1 left unmeasured (full suite after the null-stream join,
# the 4K tests, all four workloads, gemm_pipe). Every step under its own timeout.
mkdir -p gpurun_out
export TMPDIR=/tmp
( DCVC_TEST_UHD=1 timeout 700 python -m pytest tests -m gpu -q -x --durations=15 2>&1 | tail -40 ) > gpurun_out/r2_test_gpu.log
for w in intra ld hts htl; do
    timeout 200 python bench.py --workload $w --no-cpu-baseline > gpurun_out/r2_bench_$w.json 2> gpurun_out/r2_bench_$w.err
done
PIPE=dcvc_amd/libdcvc_amd_pipe.so
if [ -f $PIPE ]; then
    ( DCVC_LIB=$PWD/$PIPE timeout 120 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k gemm_pipe 2>&1 | tail -15 ) > gpurun_out/r2_test_pipe.log
    if grep -q " passed" gpurun_out/r2_test_pipe.log && ! grep -q "failed" gpurun_out/r2_test_pipe.log; then
        for sched in 1 0; do
            DCVC_LIB=$PWD/$PIPE DCVC_GEMM_PIPE=1 DCVC_GEMM_PIPE_SCHED=$sched timeout 200 python bench.py --no-cpu-baseline \
                > gpurun_out/r2_bench_intra_pipe_sched$sched.json 2> gpurun_out/r2_bench_intra_pipe_sched$sched.err
        done
    fi
fi
tail -25 gpurun_out/r2_test_gpu.log; tail -5 gpurun_out/r2_test_pipe.log 2>/dev/null
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r2_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-48s %8.1f pictures/s  gemm %.0f TFLOP/s" % (f, d["value"], d.get("roofline", {}).get("achieved", 0)))
    except Exception as e:
        print(f, "unreadable:", e)
PY
