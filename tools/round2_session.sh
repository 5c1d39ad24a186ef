#!/bin/bash
# First GPU session of the next round: everything round 1 left unmeasured, one gpurun call
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/round2_session.sh'
# Before it (on the CPU box, ~17 min): DCVC_BUILD_VARIANT=pipe python -m dcvc_amd.build
# Every step runs under its own `timeout`: gemm_pipe.hip has never run on hardware, a wrong barrier
# there hangs the GPU - its tests come LAST so that everything else is on disk by then.
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > gpurun_out/r2_test_gpu.log
( DCVC_TEST_UHD=1 timeout 300 python -m pytest tests/test_uhd_gpu.py -m gpu -q 2>&1 | tail -15 ) > gpurun_out/r2_test_uhd.log
timeout 600 bash tools/stream_matrix.sh > /dev/null 2>&1
for w in intra ld hts htl; do
    timeout 200 python bench.py --workload $w > gpurun_out/r2_bench_$w.json 2> gpurun_out/r2_bench_$w.err
done
PIPE=dcvc_amd/libdcvc_amd_pipe.so
if [ -f $PIPE ]; then
    ( DCVC_LIB=$PWD/$PIPE timeout 120 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k gemm_pipe 2>&1 | tail -15 ) > gpurun_out/r2_test_pipe.log
    if grep -q " passed" gpurun_out/r2_test_pipe.log && ! grep -q "failed" gpurun_out/r2_test_pipe.log; then
        for sched in 1 0; do
            DCVC_LIB=$PWD/$PIPE DCVC_GEMM_PIPE=1 DCVC_GEMM_PIPE_SCHED=$sched timeout 200 python bench.py \
                > gpurun_out/r2_bench_intra_pipe_sched$sched.json 2> gpurun_out/r2_bench_intra_pipe_sched$sched.err
        done
        DCVC_LIB=$PWD/$PIPE timeout 200 python bench.py --no-cpu-baseline > gpurun_out/r2_bench_intra_pipe_off.json 2>/dev/null
    fi
fi
tail -3 gpurun_out/r2_test_gpu.log gpurun_out/r2_test_uhd.log gpurun_out/r2_test_pipe.log 2>/dev/null
cat gpurun_out/stream_matrix.txt
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r2_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-48s %8.1f pictures/s  gemm %.0f TFLOP/s" % (f, d["value"], d.get("roofline", {}).get("achieved", 0)))
    except Exception as e:
        print(f, "unreadable:", e)
PY
