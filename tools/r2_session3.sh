#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "dcb_core" 2>&1 | tail -8 ) > gpurun_out/s3_test_core.log
tail -4 gpurun_out/s3_test_core.log
timeout 120 python tools/core_timeline.py > gpurun_out/s3_core_timeline.txt 2>&1
cat gpurun_out/s3_core_timeline.txt
timeout 300 python tools/host_repro.py 2>&1 | grep -v Warn > gpurun_out/s3_host_repro.txt
cat gpurun_out/s3_host_repro.txt; lscpu | grep -E "Model name|Flags" | cut -c1-400
