#!/bin/bash
# One gpurun session: tests, smoke, bench (all workloads), rocprof kernel trace. Outputs under gpurun_out/.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/test_gpu.log
python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
python bench.py --steps 30 --warmup 8 > gpurun_out/bench.log 2>&1
tail -1 gpurun_out/bench.log
for w in ld hts htl; do python bench.py --workload $w --steps 24 --warmup 6 > gpurun_out/bench_$w.log 2>&1; tail -1 gpurun_out/bench_$w.log | cut -c1-260; done
rm -rf gpurun_out/prof
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o bench -- python bench.py --steps 10 --warmup 4 --no-cpu-baseline > gpurun_out/bench_prof.log 2>&1
python tools/rocpd_stats.py gpurun_out/prof/bench_results.db gpurun_out/kernel_stats.csv > /dev/null
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ld -o ld -- python bench.py --workload ld --steps 20 --warmup 6 --no-roofline > gpurun_out/bench_prof_ld.log 2>&1
python tools/rocpd_stats.py gpurun_out/prof_ld/ld_results.db gpurun_out/ld_kernel_stats.csv > /dev/null
rm -rf gpurun_out/prof_ld gpurun_out/prof
head -8 gpurun_out/kernel_stats.csv | cut -c1-160
head -8 gpurun_out/ld_kernel_stats.csv | cut -c1-160
