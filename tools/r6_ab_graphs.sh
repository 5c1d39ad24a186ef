set -x
B="python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-uhd --no-extras --no-roofline --no-pipeline --min-seconds 0"
for w in ld hts; do
for g in 0 1; do
  DCVC_BENCH_GRAPHS=$g $B --workload $w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$w graphs=$g', round(d['value'],1), round(d['encode_fps'],1), round(d['decode_fps'],1))"
done; done
