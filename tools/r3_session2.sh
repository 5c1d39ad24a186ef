#!/bin/bash
# round 3, session 2: A/B of the block kernel's builds (policy v2 = round 2, v3, ablations) + new tests + bench
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
B=tools/_bin
$B/core_bench -r 3 -n 20 $B/lib_r2.so $B/lib_v3.so $B/lib_NOEPI.so $B/lib_NOGATHER.so $B/lib_NODMA.so $B/lib_NOBAR.so $B/lib_NODMA_NOEPI.so $B/lib_ALL.so > gpurun_out/core_bench.txt 2>&1
cat gpurun_out/core_bench.txt
timeout 900 python -m pytest tests/test_dmcht_gpu.py tests/test_cli_gpu.py -m gpu -q -x -k "fanout or fan_out or encode_decode" 2>&1 | tail -8 > gpurun_out/test_new.log
cat gpurun_out/test_new.log
timeout 900 python bench.py --steps 60 --warmup 10 > gpurun_out/bench2.log 2> gpurun_out/bench2.err
tail -1 gpurun_out/bench2.log | cut -c1-6000
tail -5 gpurun_out/bench2.err
