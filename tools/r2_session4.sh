#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "dcb_core" 2>&1 | tail -12 ) > gpurun_out/s4_test_core.log
tail -6 gpurun_out/s4_test_core.log
timeout 120 python tools/core_timeline.py > gpurun_out/s4_core_timeline.txt 2>&1
head -70 gpurun_out/s4_core_timeline.txt
