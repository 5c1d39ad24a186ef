#!/bin/bash
# round 5, first measuring session (ONE gpurun call, ~ 20 minutes): the tests that are new this round, the depthwise kernel
# A/B (three forms, one process each), the intra bench with the old and the new depthwise kernel, the hierarchical workloads
# as processes of their own (the default line's `other_workloads` ran them 10 - 25 % slower in the round's first session:
# order / box state or real?), then the default line. Outputs under gpurun_out/r05a/.
set -x
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05a
mkdir -p $O
B=tools/_bin
L=dcvc_amd/libdcvc_amd.so
{ hostname; lscpu | grep -i "model name"; rocm-smi --showuniqueid --showproductname 2>/dev/null | grep -v "^=\|^$"; cat .git_head 2>/dev/null; } > $O/box.txt 2>&1
# depthwise conv, 136 x 240 x 384 channels and x 128 channels: us per launch (core_bench prints it as "dw3x3")
for m in sliding ahead deep; do
  for ci in 384 128; do
    echo "== mode $m CI $ci" >> $O/dwconv_ab.txt
    DCVC_DWCONV_MODE=$m timeout 120 $B/core_bench -r 3 -n 20 -c $([ $ci = 384 ] && echo 384 || echo 256) -i $ci $L 2>&1 | grep -o "dw3x3 *[0-9.]* us\|dcb_nsplit + next[^|]*" >> $O/dwconv_ab.txt
  done
done
cat $O/dwconv_ab.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "dwconv" -s 2>&1 | grep -v "^$" | tail -8 | tee $O/tests_dwconv.log
A="--steps 40 --warmup 5 --no-extras --no-cpu-baseline --no-pipeline --min-seconds 2"
for m in sliding deep; do
  DCVC_DWCONV_MODE=$m timeout 300 python bench.py $A > $O/intra_dw_$m.json 2> $O/intra_dw_$m.err
  python - <<EOF
import json
d = json.load(open("$O/intra_dw_$m.json"))
print("intra, dwconv $m: value %.1f sustained %.1f enc %.1f dec %.1f closure %s" % (d["value"], d["sustained"]["value"], d["encode_fps"], d["decode_fps"], d["closure_ok"]))
EOF
done 2>&1 | tee $O/intra_dw.txt
for w in htl hts; do
  timeout 300 python bench.py --workload $w --steps 48 --warmup 12 --no-extras --no-cpu-baseline --no-roofline --min-seconds 2 > $O/alone_$w.json 2> $O/alone_$w.err
  python - <<EOF
import json
d = json.load(open("$O/alone_$w.json"))
print("$w alone: value %.1f sustained %.1f pipelined %.1f enc %.1f dec %.1f closure %s" % (d["value"], d["sustained"]["value"], d["pipelined"]["value"], d["encode_fps"], d["decode_fps"], d["closure_ok"]))
EOF
done 2>&1 | tee $O/alone.txt
timeout 1500 python -m pytest tests/test_qsweep_gpu.py tests/test_bench_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x -k "all_64 or order_covers or hand_a_running or rate_sweep or tolerance or qends_1920 or 3840x2160_q" -s > $O/tests_new.log 2>&1
echo "tests rc $?" >> $O/tests_new.log
tail -25 $O/tests_new.log
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench.err
echo "bench rc $?"
python - <<EOF
import json
d = json.load(open("$O/bench_line.json"))
print("default line: value %.1f sustained %.1f pipelined %.1f enc %.1f dec %.1f closure %s frac %.3f" % (d["value"], d["sustained"]["value"], d["pipelined"]["value"], d["encode_fps"], d["decode_fps"], d["closure_ok"], d["roofline"]["frac"]))
for k, o in d["other_workloads"].items():
    print(k, "value %.1f pipelined %.1f enc %.1f dec %.1f" % (o["value"], o["pipelined"]["value"], o["encode_fps"], o["decode_fps"]))
EOF
