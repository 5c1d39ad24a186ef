#!/bin/bash
# round 3, session 17: output rows stored straight from the epilogue's registers (NS_DIRECT: 1 = t1', 2 = y, 3 = both)
set -x
mkdir -p gpurun_out
B=tools/_bin
L=dcvc_amd/libdcvc_amd.so
{ timeout 300 $B/core_bench -r 5 -n 20 $L $B/direct1.so $B/direct2.so $B/direct3.so
  timeout 300 $B/core_bench -r 5 -n 20 -c 512 -i 256 $L $B/direct1.so $B/direct2.so $B/direct3.so
  timeout 300 $B/core_bench -r 5 -n 20 -c 512 -i 512 $L $B/direct1.so $B/direct2.so $B/direct3.so
  timeout 300 $B/core_bench -r 5 -n 20 -c 256 -i 256 $L $B/direct1.so $B/direct2.so $B/direct3.so
  timeout 300 $B/core_bench -r 5 -n 20 -c 512 -i 512 -p 8160 $L $B/direct1.so $B/direct2.so $B/direct3.so ; } > gpurun_out/core_bench17.txt 2>&1
grep -v "^  timeline\|dcb_core" gpurun_out/core_bench17.txt | cut -c1-1000
DCVC_LIB=$PWD/$B/direct3.so timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "nsplit" 2>&1 | tail -3
for w in intra hts; do
  for lib in $L $B/direct3.so; do
    DCVC_LIB=$PWD/$lib timeout 300 python bench.py --workload $w --steps 30 --warmup 6 --no-cpu-baseline --no-uhd --no-extras --no-roofline --min-seconds 0 2>/dev/null | tail -1 | cut -c1-260
  done
done
