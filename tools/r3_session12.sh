#!/bin/bash
# round 3, session 12: two 32-pixel workgroups per CU (DCVC_NSPLIT_DUAL=1) against one 64-pixel workgroup; new shapes
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
B=tools/_bin
L=dcvc_amd/libdcvc_amd.so
for dual in 0 1; do
  export DCVC_NSPLIT_DUAL=$dual
  { timeout 200 $B/core_bench -r 3 -n 20 $L
    timeout 200 $B/core_bench -r 3 -n 20 -c 512 -i 256 $L
    timeout 200 $B/core_bench -r 3 -n 20 -c 256 -i 256 $L
    timeout 200 $B/core_bench -r 3 -n 20 -c 256 -i 128 $L ; } > gpurun_out/core_bench12_dual$dual.txt 2>&1
  grep -v "^  timeline" gpurun_out/core_bench12_dual$dual.txt | cut -c1-900
  timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "nsplit" 2>&1 | tail -4
done
unset DCVC_NSPLIT_DUAL
timeout 200 $B/core_bench -r 3 -n 20 -c 768 -i 768 -p 8160 $L 2>&1 | grep -v "^  timeline" | cut -c1-600
for w in hts htl; do
  for dual in 0 1; do
    DCVC_NSPLIT_DUAL=$dual timeout 300 python bench.py --workload $w --steps 30 --warmup 6 --no-cpu-baseline --no-uhd --no-extras --min-seconds 0 > gpurun_out/bench12_${w}_dual$dual.log 2> gpurun_out/bench12_${w}_dual$dual.err
    tail -1 gpurun_out/bench12_${w}_dual$dual.log | cut -c1-300
    tail -1 gpurun_out/bench12_${w}_dual$dual.err
  done
done
