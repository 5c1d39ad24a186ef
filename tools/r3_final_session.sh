#!/bin/bash
# round 3, closing session: the whole GPU suite, smoke(), the default bench line, kernel-trace summaries per workload
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r03_test_gpu.log
tail -4 gpurun_out/r03_test_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r03_bench_line.json 2> gpurun_out/r03_bench.err
tail -1 gpurun_out/r03_bench_line.json | cut -c1-1500
tail -2 gpurun_out/r03_bench.err
cd /tmp
for w in intra hts htl ld; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o t -- python $R/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-uhd --no-extras --no-roofline --min-seconds 0 > $R/gpurun_out/r03_prof_$w.log 2>&1
  cp /tmp/prof_$w/t_kernel_stats.csv $R/gpurun_out/r03_${w}_kernel_stats.csv
  head -8 /tmp/prof_$w/t_kernel_stats.csv | cut -c1-200
done
