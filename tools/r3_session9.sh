#!/bin/bash
# round 3, session 9: the depthwise conv inside the N-split block launch
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
B=tools/_bin
timeout 240 $B/core_bench -r 3 -n 20 $B/lib_head.so > gpurun_out/core_bench9.txt 2>&1
grep -v "^  timeline" gpurun_out/core_bench9.txt
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "nsplit" 2>&1 | tail -6
timeout 900 python -m pytest tests/test_dmci_gpu.py -m gpu -q -x 2>&1 | tail -4
timeout 600 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-uhd > gpurun_out/bench9.log 2> gpurun_out/bench9.err
tail -1 gpurun_out/bench9.log | cut -c1-1200
tail -2 gpurun_out/bench9.err
DCVC_NSPLIT_DW=0 timeout 600 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-uhd --no-extras --no-roofline > gpurun_out/bench9b.log 2> gpurun_out/bench9b.err
tail -1 gpurun_out/bench9b.log | cut -c1-400
