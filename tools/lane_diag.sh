p() { tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$1', d['config']['lanes'], round(d['value'],1), round(d.get('one_lane',{}).get('value',0),1), round(d['roofline']['achieved'],1))"; }
DCVC_BENCH_USER_STREAM=side python bench.py --workload intra --lanes 2 --no-cpu-baseline 2>/dev/null | p intra2_side
DCVC_BENCH_POOL1=1 python bench.py --workload ld --lanes 1 2>/dev/null | p ld1_pool_null
DCVC_BENCH_POOL1=1 DCVC_BENCH_USER_STREAM=side python bench.py --workload ld --lanes 1 2>/dev/null | p ld1_pool_side
DCVC_BENCH_USER_STREAM=side python bench.py --workload ld --lanes 1 2>/dev/null | p ld1_side
python bench.py --workload ld --lanes 1 2>/dev/null | p ld1_plain
