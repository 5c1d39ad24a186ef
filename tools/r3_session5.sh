#!/bin/bash
# round 3, session 5: N-split kernel with the cheap prologue and the pipelined ffn.0 epilogue
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
B=tools/_bin
timeout 240 $B/core_bench -r 3 -n 20 $B/lib_head.so > gpurun_out/core_bench5.txt 2>&1
grep -v "^  timeline" gpurun_out/core_bench5.txt
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "nsplit" 2>&1 | tail -4
timeout 600 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-uhd > gpurun_out/bench5.log 2> gpurun_out/bench5.err
tail -1 gpurun_out/bench5.log | cut -c1-4500
tail -3 gpurun_out/bench5.err
