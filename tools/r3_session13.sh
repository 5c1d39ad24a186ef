#!/bin/bash
# round 3, session 13: next tile's transfers requested behind ffn.2's MFMAs, x in registers
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
B=tools/_bin
L=dcvc_amd/libdcvc_amd.so
{ timeout 200 $B/core_bench -r 3 -n 20 $L
  timeout 200 $B/core_bench -r 3 -n 20 -c 512 -i 256 $L
  timeout 200 $B/core_bench -r 3 -n 20 -c 512 -i 512 $L
  timeout 200 $B/core_bench -r 3 -n 20 -c 256 -i 256 $L
  timeout 200 $B/core_bench -r 3 -n 20 -c 256 -i 128 $L
  timeout 200 $B/core_bench -r 3 -n 20 -c 768 -i 768 -p 8160 $L ; } > gpurun_out/core_bench13.txt 2>&1
grep -v "^  timeline" gpurun_out/core_bench13.txt | cut -c1-900
timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "nsplit" 2>&1 | tail -4
for w in intra hts htl ld; do
  timeout 300 python bench.py --workload $w --steps 30 --warmup 6 --no-cpu-baseline --no-uhd --no-extras --min-seconds 0 > gpurun_out/bench13_$w.log 2> gpurun_out/bench13_$w.err
  tail -1 gpurun_out/bench13_$w.log | cut -c1-300
  tail -1 gpurun_out/bench13_$w.err
done
