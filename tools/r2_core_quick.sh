#!/bin/bash
# quick loop for dcb_core tuning: bit-identity test of the 1080p chain configuration + wall time / coarse timeline
mkdir -p gpurun_out
( timeout 200 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "dcb_core and 32640" 2>&1 | tail -3 ) > gpurun_out/q_test.log
tail -2 gpurun_out/q_test.log
timeout 120 python tools/core_timeline.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/q_timeline.txt
