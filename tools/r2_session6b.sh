#!/bin/bash
# Round 2, GPU session 6b: store-aware vmcnt in dcb_core's last phase, CLI test (fixed reference conversion),
# regenerated full-size digests, host share of decompress() with the spinning worker pool
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "dcb_core" 2>&1 | tail -3 ) | tee gpurun_out/s6b_core_test.log
timeout 120 python tools/core_timeline.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s6b_timeline.txt
echo "== cli + digests"; ( timeout 1200 python -m pytest tests/test_cli_gpu.py tests/test_fullsize_gpu.py -m gpu -q --durations=6 2>&1 | tail -25 ) | tee gpurun_out/s6b_tests.log
echo "== host time of decompress()"
( DCVC_TIMING=1 timeout 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-roofline --no-extras 2>&1 >/dev/null | grep "decompress host" | tail -5 ) | tee gpurun_out/s6b_timing.log
echo "== bench"
timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/s6b_bench.json 2> gpurun_out/s6b_bench.err; wc -l gpurun_out/s6b_bench.json; cut -c1-300 gpurun_out/s6b_bench.json; tail -2 gpurun_out/s6b_bench.err
