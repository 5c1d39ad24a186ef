#!/bin/bash
# round 3, session 18: 4 / 8 / 16 interleaved copies of the WSiLU table (as many as fit beside the activation tiles)
set -x
mkdir -p gpurun_out
B=tools/_bin
L=dcvc_amd/libdcvc_amd.so
{ timeout 300 $B/core_bench -r 5 -n 20 $B/tab4.so $B/tab8.so $L
  timeout 300 $B/core_bench -r 5 -n 20 -c 512 -i 256 $B/tab4.so $B/tab8.so $L
  timeout 300 $B/core_bench -r 5 -n 20 -c 512 -i 512 $B/tab4.so $B/tab8.so $L
  timeout 300 $B/core_bench -r 5 -n 20 -c 256 -i 256 $B/tab4.so $B/tab8.so $L
  timeout 300 $B/core_bench -r 5 -n 20 -c 256 -i 128 $B/tab4.so $B/tab8.so $L
  timeout 300 $B/core_bench -r 5 -n 20 -c 512 -i 512 -p 8160 $B/tab4.so $B/tab8.so $L
  timeout 300 $B/core_bench -r 5 -n 20 -c 768 -i 768 -p 8160 $B/tab4.so $B/tab8.so $L ; } > gpurun_out/core_bench18.txt 2>&1
grep -v "^  timeline\|dcb_core\|nsplit timeline" gpurun_out/core_bench18.txt | cut -c1-230
timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "nsplit" 2>&1 | tail -3
