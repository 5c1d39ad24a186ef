#!/bin/bash
# Round 2, GPU session 6: regenerated full-size digests (corrected matrix-core model), native CLI, recon-head fan-out,
# host entropy-decoder share, copy launches per step
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== new tests"; ( timeout 900 python -m pytest tests/test_cli_gpu.py tests/test_dmcht_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x --durations=8 2>&1 | tail -30 ) | tee gpurun_out/s6_tests.log
echo "== host time of decompress()"
( DCVC_TIMING=1 timeout 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-roofline --no-extras 2>&1 >/dev/null | grep "decompress host" | tail -12 ) | tee gpurun_out/s6_timing.log
echo "== bench"
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/s6_bench.json 2> gpurun_out/s6_bench.err; wc -l gpurun_out/s6_bench.json; cut -c1-400 gpurun_out/s6_bench.json; tail -3 gpurun_out/s6_bench.err
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/s6_prof -o s6 -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/s6_prof.log 2>&1
cd $R
python tools/rocpd_stats.py $(find gpurun_out/s6_prof -name "*.db" | head -1) gpurun_out/s6_kernel_stats.csv | head -24 | cut -c1-160
