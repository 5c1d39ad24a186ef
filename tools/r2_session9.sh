#!/bin/bash
# Round 2, GPU session 9: fine issue-time stamps of a dc.3 step; rocprofv3 kernel-trace summaries of the three inter
# workloads (profiles/r02_bench_{ld,hts,htl}_kernel_stats.csv) and their bench lines
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 python tools/core_timeline.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s9_timeline.txt
R=$GRAFT_REPO_ROOT
for w in ld hts htl; do
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/s9_prof_$w -o s9 -- python $R/bench.py --workload $w --steps 16 --warmup 4 --no-extras --no-roofline > $R/gpurun_out/s9_bench_$w.json 2> $R/gpurun_out/s9_bench_$w.err
  cd $R
  python tools/rocpd_stats.py $(find gpurun_out/s9_prof_$w -name "*.db" | head -1) gpurun_out/r02_bench_${w}_kernel_stats.csv | head -8 | cut -c1-150
  rm -rf gpurun_out/s9_prof_$w
done
