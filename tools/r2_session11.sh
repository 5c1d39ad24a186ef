#!/bin/bash
# Round 2, GPU session 11: final-state measurements - default bench line, kernel trace summary, PMC traffic
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py > gpurun_out/s11_bench.json 2> gpurun_out/s11_bench.err
wc -l gpurun_out/s11_bench.json; cut -c1-250 gpurun_out/s11_bench.json; tail -2 gpurun_out/s11_bench.err
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/s11_prof -o s11 -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/s11_prof.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/s11_fetch -o f -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/s11_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/s11_write -o w -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/s11_write.log 2>&1
cd $R
python tools/hbm_traffic.py gpurun_out/s11_fetch gpurun_out/s11_write gpurun_out/r02_hbm_traffic_final.json ${DCVC_COMMIT:-final} | tail -24
python tools/rocpd_stats.py $(find gpurun_out/s11_prof -name "*.db" | head -1) gpurun_out/r02_bench_kernel_stats_final.csv | head -16 | cut -c1-150
find gpurun_out/s11_fetch gpurun_out/s11_write -name "*.csv" -size +3M -delete
rm -rf gpurun_out/s11_prof
