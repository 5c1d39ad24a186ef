#!/bin/bash
# round 3, session 4: where the N-split block kernel's time goes (ablations, prefetch depth, in-kernel timeline)
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
B=tools/_bin
timeout 240 $B/core_bench -r 3 -n 20 $B/lib_head.so $B/lib_NS_NOLOAD.so $B/lib_NS_NOEPI.so $B/lib_NS_NOLOAD_NOEPI.so $B/lib_NS_RING32.so $B/lib_NS_RING8.so > gpurun_out/core_bench4.txt 2>&1
grep -v "^  timeline\|dcb_core + next" gpurun_out/core_bench4.txt
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "nsplit" 2>&1 | tail -4
