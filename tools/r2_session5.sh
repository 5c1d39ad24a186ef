#!/bin/bash
# Round 2, GPU session 5: the restructured bench (all four workloads in one line), kernel trace, PMC traffic
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python bench.py > gpurun_out/r2_bench_full.json 2> gpurun_out/r2_bench_full.err
tail -c 3000 gpurun_out/r2_bench_full.json; tail -3 gpurun_out/r2_bench_full.err
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/s5_prof -o s5 -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/s5_prof.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/s5_fetch -o f -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/s5_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/s5_write -o w -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/s5_write.log 2>&1
cd $R
python tools/hbm_traffic.py gpurun_out/s5_fetch gpurun_out/s5_write gpurun_out/r02_hbm_traffic.json 6116050 | tail -30
python tools/rocpd_stats.py $(find gpurun_out/s5_prof -name "*.db" | head -1) gpurun_out/r02_bench_kernel_stats.csv | head -14 | cut -c1-150
find gpurun_out/s5_fetch gpurun_out/s5_write -name "*.csv" -size +3M -delete
