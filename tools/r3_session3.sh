#!/bin/bash
# round 3, session 3: first run of the N-split block kernel (A/B against dcb_core builds), its unit tests, codec tests, bench
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
B=tools/_bin
timeout 180 $B/core_bench -r 3 -n 20 $B/lib_head.so $B/lib_v3.so $B/lib_STAGGER.so $B/lib_STAGED.so $B/lib_STAGGER_STAGED.so > gpurun_out/core_bench3.txt 2>&1
cat gpurun_out/core_bench3.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "nsplit" 2>&1 | tail -15 > gpurun_out/test_nsplit.log
cat gpurun_out/test_nsplit.log
timeout 900 python -m pytest tests/test_dmci_gpu.py tests/test_dmcht_gpu.py -m gpu -q -x 2>&1 | tail -8 > gpurun_out/test_codec.log
cat gpurun_out/test_codec.log
timeout 600 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-uhd > gpurun_out/bench3.log 2> gpurun_out/bench3.err
tail -1 gpurun_out/bench3.log | cut -c1-2500
tail -3 gpurun_out/bench3.err
