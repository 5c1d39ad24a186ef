#!/bin/bash
# round 6, closing session (ONE gpurun call, ~ 30 minutes): box identity, the block bench for every shape, PMC traffic of the
# bench's kernels, counters of the block kernel, the whole GPU suite, smoke(), the default bench line, kernel-trace summaries per
# workload (+ launches per coded unit behind the set-up). Outputs under gpurun_out/r06/; the ones quoted in DESIGN.md are copied
# to profiles/r06_*.
set -x
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06
mkdir -p $O
B=tools/_bin
L=dcvc_amd/libdcvc_amd.so
{ hostname; lscpu | grep -i "model name"; rocm-smi --showuniqueid --showproductname 2>/dev/null | grep -v "^=\|^$"; cat .git_head 2>/dev/null; } > $O/r06_box.txt 2>&1
us=$(timeout 120 $B/core_bench -r 2 -n 10 $L | grep "dcb_nsplit + next" | head -1 | awk '{print $5}')
echo "block kernel: $us us" | tee -a $O/r06_box.txt
if [ -z "$us" ] || awk -v u="$us" 'BEGIN { exit !(u > 100) }'; then echo "SLOW BOX - stopping"; exit 7; fi
{ for sh in "384 384 32640" "512 256 32640" "512 512 32640" "256 256 32640" "256 128 32640" "512 512 8160" "768 768 8160" "384 192 8160" "384 384 129600"; do
    set -- $sh; echo "=== C $1 CI $2 pixels $3"
    if [ $3 = 32640 ]; then timeout 200 $B/core_bench -r 3 -n 20 -c $1 -i $2 $L; else timeout 200 $B/core_bench -r 3 -n 20 -c $1 -i $2 -p $3 $L; fi
  done; } 2>&1 | grep "===\|dcb_nsplit + next\|dw3x3" | grep -o "===.*\|dcb_nsplit + next[^|]*|[^|]*\|dw3x3 *[0-9.]* us" > $O/r06_core_bench_shapes.txt
cat $O/r06_core_bench_shapes.txt
BENCH="python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-roofline --no-extras --no-uhd --no-resolutions --no-pipeline --min-seconds 0"
cd /tmp
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc6/bench_fetch -o bench_fetch -- $BENCH > $O/pmc_bench_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc6/bench_write -o bench_write -- $BENCH > $O/pmc_bench_write.log 2>&1
cd $R
python tools/hbm_traffic.py /tmp/pmc6/bench_fetch /tmp/pmc6/bench_write $O/r06_hbm_traffic.json "$(cat .git_head 2>/dev/null)" | grep -A5 "nsplit8_kernel<384"
cp $O/r06_hbm_traffic.json profiles/r06_hbm_traffic.json
# MFMA-busy / busy cycles of the block launches (core_bench, (384, 384) at 1080p / 8)
cd /tmp
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmc6/clk_lib -o clk -- $R/$B/core_bench -r 2 -n 10 $R/$L > $O/clk.log 2>&1
cd $R
python tools/pmc_summary.py /tmp/pmc6 2>/dev/null | grep -i "clk\|nsplit" | cut -c1-260 > $O/r06_block_counters.txt
cat $O/r06_block_counters.txt
timeout 2400 python -m pytest tests -m gpu -q -rs 2>&1 | tail -25 > $O/r06_test_gpu.log
tail -6 $O/r06_test_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.log
timeout 900 python bench.py > $O/r06_bench_line.json 2> $O/r06_bench.err
tail -c 600 $O/r06_bench_line.json
tail -2 $O/r06_bench.err
cd /tmp
for w in intra hts htl ld; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof6_$w -o t -- python $R/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-uhd --no-resolutions --no-extras --no-roofline --no-pipeline --min-seconds 0 > $O/r06_prof_$w.log 2>&1
  find /tmp/prof6_$w -name "t_kernel_stats.csv" -exec cp {} $O/r06_${w}_kernel_stats.csv \;
  head -4 $O/r06_${w}_kernel_stats.csv | cut -c1-200
  case $w in intra|htl) M="y_step_enc"; P=4;; hts) M="mask_step_enc"; P=4;; *) M="mask_step_enc"; P=2;; esac
  python $R/tools/trace_after_setup.py /tmp/prof6_$w --marker $M --per $P > $O/r06_${w}_per_picture.txt 2>&1
  head -12 $O/r06_${w}_per_picture.txt
done
cd $R
# per-shape timing of the contraction launches (bench.py's event-stamped pass)
for w in ld intra hts htl; do
  DCVC_BENCH_SHAPES=$O/r06_gemm_shapes_$w.csv timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-uhd --no-resolutions --no-extras --no-pipeline --min-seconds 0 > /dev/null 2>&1
done
# round 6 against round 5's library on this box (tools/_bin/r05.so, when it travelled with the tree)
if [ -f tools/_bin/r05.so ]; then bash tools/r6_ab_round.sh 2>/dev/null | grep "^pass" > $O/r06_round_ab_closing.txt; cat $O/r06_round_ab_closing.txt; fi
# round 6, last session: the depthwise conv inside the (256, 128) / (384, 192) block launches - the LD codec with and without on this box
# (fps, kernel trace), the launch under core_bench with its in-kernel stamps, the issue-rate probe behind its arithmetic
B2="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-uhd --no-extras --no-roofline --no-pipeline --no-resolutions --min-seconds 0"
for pass in 1 2 3; do for g in 0 1; do
  DCVC_NSPLIT_DW=$g timeout 300 $B2 --workload ld 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('pass $pass ld 1920x1080 depthwise_inside=$g', round(d['value'],1), 'enc', round(d['encode_fps'],1), 'dec', round(d['decode_fps'],1))"
done; done > $O/r06_dw_ab.txt
for g in 0 1; do
  DCVC_NSPLIT_DW=$g timeout 300 $B2 --workload ld --resolution 3840x2160 --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ld 3840x2160 depthwise_inside=$g', round(d['value'],1), 'enc', round(d['encode_fps'],1), 'dec', round(d['decode_fps'],1))"
done >> $O/r06_dw_ab.txt
cat $O/r06_dw_ab.txt
cd /tmp
DCVC_NSPLIT_DW=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof6_ld_dw0 -o t -- python $R/bench.py --workload ld --steps 10 --warmup 3 --no-cpu-baseline --no-uhd --no-resolutions --no-extras --no-roofline --no-pipeline --min-seconds 0 > $O/r06_prof_ld_dw0.log 2>&1
python $R/tools/trace_after_setup.py /tmp/prof6_ld_dw0 --marker mask_step_enc --per 2 > $O/r06_ld_per_picture_dw0.txt 2>&1
for g in 0 1; do
  DCVC_NSPLIT_DW=$g timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof6_ld4k_$g -o t -- python $R/bench.py --workload ld --resolution 3840x2160 --steps 6 --warmup 2 --no-cpu-baseline --no-uhd --no-resolutions --no-extras --no-roofline --no-pipeline --min-seconds 0 > $O/r06_prof_ld4k_$g.log 2>&1
  python $R/tools/trace_after_setup.py /tmp/prof6_ld4k_$g --marker mask_step_enc --per 2 > $O/r06_ld4k_per_picture_dw$g.txt 2>&1
done
cd $R
bash tools/r6_dw_timeline.sh 2>&1 | cut -c1-1200 > $O/r06_dw_timeline.txt
timeout 60 $B/fma_rate > $O/r06_fma_rate.txt 2>&1
