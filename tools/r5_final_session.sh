#!/bin/bash
# round 5, closing session (ONE gpurun call, ~ 25 minutes): box identity, PMC traffic of the bench's kernels, the whole GPU
# suite, smoke(), the default bench line, kernel-trace summaries per workload (+ launches per picture behind the set-up),
# the 64-point rate sweep at 3840x2160 (BASELINE configs[4]) and the GOP hand-off mode (two ranks on this one GPU, gloo).
# Outputs under gpurun_out/r05/; the ones quoted in DESIGN.md are copied to profiles/r05_*.
set -x
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05
mkdir -p $O
B=tools/_bin
L=dcvc_amd/libdcvc_amd.so
{ hostname; lscpu | grep -i "model name"; rocm-smi --showuniqueid --showproductname 2>/dev/null | grep -v "^=\|^$"; cat .git_head 2>/dev/null; } > $O/r05_box.txt 2>&1
# a throttled box is not worth the GPU minutes: check the block kernel first
us=$(timeout 120 $B/core_bench -r 2 -n 10 $L | grep "dcb_nsplit + next" | head -1 | awk '{print $5}')
echo "block kernel: $us us" | tee -a $O/r05_box.txt
if [ -z "$us" ] || awk -v u="$us" 'BEGIN { exit !(u > 100) }'; then echo "SLOW BOX - stopping"; exit 7; fi
# block bench, every shape (one process per shape)
{ for sh in "384 384 32640" "512 256 32640" "512 512 32640" "256 256 32640" "256 128 32640" "512 512 8160" "768 768 8160" "384 384 129600"; do
    set -- $sh; echo "=== C $1 CI $2 pixels $3"
    if [ $3 = 32640 ]; then timeout 200 $B/core_bench -r 3 -n 20 -c $1 -i $2 $L; else timeout 200 $B/core_bench -r 3 -n 20 -c $1 -i $2 -p $3 $L; fi
  done; } 2>&1 | grep "===\|dcb_nsplit + next\|dw3x3" | grep -o "===.*\|dcb_nsplit + next[^|]*|[^|]*\|dw3x3 *[0-9.]* us" > $O/r05_core_bench_shapes.txt
cat $O/r05_core_bench_shapes.txt
BENCH="python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-roofline --no-extras --no-uhd --no-pipeline --min-seconds 0"
cd /tmp
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc5/bench_fetch -o bench_fetch -- $BENCH > $O/pmc_bench_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc5/bench_write -o bench_write -- $BENCH > $O/pmc_bench_write.log 2>&1
cd $R
python tools/hbm_traffic.py /tmp/pmc5/bench_fetch /tmp/pmc5/bench_write $O/r05_hbm_traffic.json "$(cat .git_head 2>/dev/null)" | grep -A5 "nsplit8_kernel<384"
cp $O/r05_hbm_traffic.json profiles/r05_hbm_traffic.json
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/r05_test_gpu.log
tail -6 $O/r05_test_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.log
timeout 900 python bench.py > $O/r05_bench_line.json 2> $O/r05_bench.err
tail -c 400 $O/r05_bench_line.json
tail -2 $O/r05_bench.err
cd /tmp
for w in intra hts htl ld; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof5_$w -o t -- python $R/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-uhd --no-extras --no-roofline --no-pipeline --min-seconds 0 > $O/r05_prof_$w.log 2>&1
  find /tmp/prof5_$w -name "t_kernel_stats.csv" -exec cp {} $O/r05_${w}_kernel_stats.csv \;
  head -4 $O/r05_${w}_kernel_stats.csv | cut -c1-200
  case $w in intra|htl) M="y_step_enc"; P=4;; *) M="mask_step_enc"; P=2;; esac
  python $R/tools/trace_after_setup.py /tmp/prof5_$w --marker $M --per $P > $O/r05_${w}_per_picture.txt 2>&1
  head -12 $O/r05_${w}_per_picture.txt
done
cd $R
# BASELINE configs[4]: the 64-point rate sweep at 3840x2160 (one rank here; the ranks of an N-GPU run shard the rate points)
timeout 600 python bench.py --sweep64 --workload ld > $O/r05_sweep64_ld.json 2> $O/r05_sweep64_ld.err
timeout 600 python bench.py --sweep64 --workload hts --sweep-units 1 > $O/r05_sweep64_hts.json 2> $O/r05_sweep64_hts.err
for f in $O/r05_sweep64_ld.json $O/r05_sweep64_hts.json; do python - <<EOF
import json
d = json.load(open("$f"))
s = d["sweep64"]
print("sweep64 %s: %.1f pictures/s, %d rate points, closure %s, bpp q0 %.3f .. q63 %.3f" % (s["workload"], s["value"], s["rate_points"], s["closure_ok"], s["bpp_per_q"][0], s["bpp_per_q"][63]))
EOF
done
# north_star's context exchange, rehearsed: ONE LD stream at 1080p handed between two ranks on this GPU every 8 pictures
# (gloo: the state crosses host memory - the time per hand-off is NOT an xGMI number)
DCVC_BENCH_BACKEND=gloo DCVC_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --workload ld --handoff 8 --steps 64 --warmup 8 > $O/r05_handoff_ld_gloo.json 2> $O/r05_handoff_ld.err
python - <<EOF
import json
d = json.load(open("$O/r05_handoff_ld_gloo.json"))
print("handoff:", d["value"], d["handoff"])
EOF
