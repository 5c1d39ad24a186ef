#!/bin/bash
# round 3, closing session: PMC traffic of the bench's kernels, the whole GPU suite, smoke(), the default bench line,
# kernel-trace summaries per workload, the block bench for every shape
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
B=tools/_bin
L=dcvc_amd/libdcvc_amd.so
# a throttled box (seen once: everything 1.7x slower) is not worth the GPU minutes: check the block kernel first
us=$(timeout 120 $B/core_bench -r 2 -n 10 $L | grep "dcb_nsplit + next" | head -1 | awk '{print $5}')
echo "block kernel: $us us"
if [ -z "$us" ] || awk -v u="$us" 'BEGIN { exit !(u > 105) }'; then echo "SLOW BOX - stopping"; exit 7; fi
BENCH="python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-roofline --no-extras --no-uhd --min-seconds 0"
cd /tmp
timeout 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc3/bench_fetch -o bench_fetch -- $BENCH > $R/gpurun_out/pmc_bench_fetch.log 2>&1
timeout 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc3/bench_write -o bench_write -- $BENCH > $R/gpurun_out/pmc_bench_write.log 2>&1
cd $R
python tools/hbm_traffic.py /tmp/pmc3/bench_fetch /tmp/pmc3/bench_write gpurun_out/r03_hbm_traffic.json "$(cat .git_head 2>/dev/null)" | grep -A5 "nsplit_kernel<384"
cp gpurun_out/r03_hbm_traffic.json profiles/r03_hbm_traffic.json
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r03_test_gpu.log
tail -4 gpurun_out/r03_test_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r03_bench_line.json 2> gpurun_out/r03_bench.err
tail -1 gpurun_out/r03_bench_line.json | cut -c1-700
tail -2 gpurun_out/r03_bench.err
{ timeout 200 $B/core_bench -r 3 -n 20 $L
  timeout 200 $B/core_bench -r 3 -n 20 -c 512 -i 256 $L
  timeout 200 $B/core_bench -r 3 -n 20 -c 512 -i 512 $L
  timeout 200 $B/core_bench -r 3 -n 20 -c 256 -i 256 $L
  timeout 200 $B/core_bench -r 3 -n 20 -c 256 -i 128 $L
  timeout 200 $B/core_bench -r 3 -n 20 -c 512 -i 512 -p 8160 $L
  timeout 200 $B/core_bench -r 3 -n 20 -c 768 -i 768 -p 8160 $L ; } > gpurun_out/r03_core_bench_shapes.txt 2>&1
grep "dcb_nsplit + next" gpurun_out/r03_core_bench_shapes.txt | cut -c1-200
cd /tmp
for w in intra hts htl ld; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o t -- python $R/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-uhd --no-extras --no-roofline --min-seconds 0 > $R/gpurun_out/r03_prof_$w.log 2>&1
  cp /tmp/prof_$w/t_kernel_stats.csv $R/gpurun_out/r03_${w}_kernel_stats.csv
  head -4 /tmp/prof_$w/t_kernel_stats.csv | cut -c1-200
done
