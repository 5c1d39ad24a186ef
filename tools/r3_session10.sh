#!/bin/bash
# round 3, session 10: the N-split block kernel for the half-width (dcb2) blocks of the inter models
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
B=tools/_bin
L=dcvc_amd/libdcvc_amd.so
{ timeout 200 $B/core_bench -r 3 -n 20 $L
  timeout 200 $B/core_bench -r 3 -n 20 -c 512 -i 256 $L
  timeout 200 $B/core_bench -r 3 -n 20 -c 256 -i 128 $L
  timeout 200 $B/core_bench -r 3 -n 20 -c 512 -i 512 -p 8160 $L ; } > gpurun_out/core_bench10.txt 2>&1
grep -v "^  timeline" gpurun_out/core_bench10.txt
timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "nsplit" 2>&1 | tail -6
timeout 900 python -m pytest tests/test_dmcht_gpu.py tests/test_dmcld_gpu.py -m gpu -q -x 2>&1 | tail -4
for w in ld hts htl; do
  timeout 300 python bench.py --workload $w --steps 40 --warmup 8 --no-cpu-baseline --no-uhd --no-extras > gpurun_out/bench10_$w.log 2> gpurun_out/bench10_$w.err
  tail -1 gpurun_out/bench10_$w.log | cut -c1-900
  tail -2 gpurun_out/bench10_$w.err
  DCVC_NSPLIT=1 timeout 300 python bench.py --workload $w --steps 40 --warmup 8 --no-cpu-baseline --no-uhd --no-extras --no-roofline > gpurun_out/bench10_${w}_off.log 2> gpurun_out/bench10_${w}_off.err
  tail -1 gpurun_out/bench10_${w}_off.log | cut -c1-300
done
