#!/bin/bash
# round 3, session 15: LD eager launches against hipGraph replay now that a block is 2 launches
set -x
mkdir -p gpurun_out
for g in 0 1; do
  DCVC_BENCH_GRAPHS=$g timeout 300 python bench.py --workload ld --steps 60 --warmup 10 --no-cpu-baseline --no-uhd --no-extras --no-roofline > gpurun_out/bench15_ld_g$g.log 2> gpurun_out/bench15_ld_g$g.err
  tail -1 gpurun_out/bench15_ld_g$g.log | cut -c1-260
  tail -1 gpurun_out/bench15_ld_g$g.err
done
for g in 0 1; do
  DCVC_BENCH_GRAPHS=$g timeout 300 python bench.py --workload hts --steps 30 --warmup 6 --no-cpu-baseline --no-uhd --no-extras --no-roofline > gpurun_out/bench15_hts_g$g.log 2> gpurun_out/bench15_hts_g$g.err
  tail -1 gpurun_out/bench15_hts_g$g.log | cut -c1-260
done
