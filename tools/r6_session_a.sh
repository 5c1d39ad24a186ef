#!/bin/bash
# round 6, session A (one gpurun call): the whole GPU suite, then launches per coded picture (kernel trace behind the set-up) and a
# short bench line for the four workloads. Outputs under gpurun_out/r06/.
set -x
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06
mkdir -p $O
{ hostname; lscpu | grep -i "model name"; rocm-smi --showuniqueid 2>/dev/null | grep -i "unique"; cat .git_head 2>/dev/null; } > $O/box.txt 2>&1
if [ "$1" != "notests" ]; then
  timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/test_gpu.log
  tail -6 $O/test_gpu.log
fi
cd /tmp
for w in intra ld hts htl; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof6_$w -o t -- python $R/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-uhd --no-extras --no-roofline --no-pipeline --min-seconds 0 > $O/prof_$w.log 2>&1
  find /tmp/prof6_$w -name "t_kernel_stats.csv" -exec cp {} $O/${w}_kernel_stats.csv \;
  case $w in intra|htl) M="y_step_enc"; P=4;; hts) M="mask_step_enc"; P=4;; *) M="mask_step_enc"; P=2;; esac
  python $R/tools/trace_after_setup.py /tmp/prof6_$w --marker $M --per $P > $O/${w}_per_picture.txt 2>&1
  head -14 $O/${w}_per_picture.txt
done
cd $R
timeout 900 python bench.py --no-cpu-baseline --no-uhd --no-resolutions > $O/bench_line.json 2> $O/bench.err
python - <<PY
import json
d = json.load(open("$O/bench_line.json"))
print("intra", round(d["value"], 1), round(d["encode_fps"], 1), round(d["decode_fps"], 1), d["roofline"]["frac"])
print(d["config"].get("other"))
PY
