#!/bin/bash
# round 3, session 6: N-split tuning (prologue order, epilogue stage order, sched_group_barrier interleave)
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
B=tools/_bin
timeout 240 $B/core_bench -r 3 -n 20 $B/lib_head.so $B/lib_NS_SGB5.so > gpurun_out/core_bench6.txt 2>&1
grep -v "^  timeline\|dcb_core + next" gpurun_out/core_bench6.txt
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "nsplit" 2>&1 | tail -4
