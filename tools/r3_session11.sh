#!/bin/bash
# round 3, session 11: where the inter workloads' time goes (contraction shapes + whole kernel trace)
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
for w in hts htl ld; do
  DCVC_BENCH_SHAPES=gpurun_out/shapes11_$w.csv timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-uhd --no-extras --min-seconds 0 > gpurun_out/bench11_$w.log 2> gpurun_out/bench11_$w.err
  head -40 gpurun_out/shapes11_$w.csv
done
cd /tmp
for w in hts ld; do
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof11_$w -o t -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-uhd --no-extras --no-roofline --min-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/prof11_$w.log 2>&1
  f=$(find /tmp/prof11_$w -name "*kernel_stats.csv" | head -1)
  cp "$f" $GRAFT_REPO_ROOT/gpurun_out/prof11_${w}_kernel_stats.csv
  head -30 "$f" | cut -c1-200
done
