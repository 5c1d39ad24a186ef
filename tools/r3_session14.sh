#!/bin/bash
# round 3, session 14: x of the next tile behind dc.0's MFMAs, prologue without the wait for x; HT-S kernel trace
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
B=tools/_bin
L=dcvc_amd/libdcvc_amd.so
{ timeout 200 $B/core_bench -r 3 -n 20 $L
  timeout 200 $B/core_bench -r 3 -n 20 -c 512 -i 256 $L
  timeout 200 $B/core_bench -r 3 -n 20 -c 512 -i 512 $L
  timeout 200 $B/core_bench -r 3 -n 20 -c 512 -i 512 -p 8160 $L ; } > gpurun_out/core_bench14.txt 2>&1
grep -v "^  timeline" gpurun_out/core_bench14.txt | cut -c1-900
timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "nsplit" 2>&1 | tail -4
for w in intra hts htl; do
  timeout 300 python bench.py --workload $w --steps 30 --warmup 6 --no-cpu-baseline --no-uhd --no-extras --min-seconds 0 > gpurun_out/bench14_$w.log 2> gpurun_out/bench14_$w.err
  tail -1 gpurun_out/bench14_$w.log | cut -c1-300
  tail -1 gpurun_out/bench14_$w.err
done
R=$PWD
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof14_hts -o t -- python $R/bench.py --workload hts --steps 10 --warmup 3 --no-cpu-baseline --no-uhd --no-extras --no-roofline --min-seconds 0 > $R/gpurun_out/prof14_hts.log 2>&1
find /tmp/prof14_hts -name "*.csv" | head
f=$(find /tmp/prof14_hts -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/prof14_hts_kernel_stats.csv
head -40 "$f" | cut -c1-250
