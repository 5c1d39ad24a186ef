#!/bin/bash
# One gpurun session of round 3: GPU tests (optionally without the full-size digests), bench line of every workload,
# kernel-trace summary. Outputs under gpurun_out/.   usage: tools/r3_session.sh [nodigest] [noprof]
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
IGN=""
[[ " $* " == *" nodigest "* ]] && IGN="--ignore=tests/test_fullsize_gpu.py"
timeout 1500 python -m pytest tests -m gpu -q -x $IGN 2>&1 | tail -25 > gpurun_out/test_gpu.log
tail -3 gpurun_out/test_gpu.log
timeout 600 python bench.py --steps 100 --warmup 10 > gpurun_out/bench.log 2> gpurun_out/bench.err
tail -1 gpurun_out/bench.log | cut -c1-3000
if [[ " $* " != *" noprof "* ]]; then
  rm -rf gpurun_out/prof
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o bench -- python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-extras > gpurun_out/bench_prof.log 2>&1
  python tools/rocpd_stats.py gpurun_out/prof/bench_results.db gpurun_out/kernel_stats.csv > /dev/null
  rm -rf gpurun_out/prof
  head -12 gpurun_out/kernel_stats.csv | cut -c1-200
fi
