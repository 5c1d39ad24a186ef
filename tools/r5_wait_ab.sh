A="--steps 60 --warmup 8 --no-extras --no-cpu-baseline --no-roofline --no-pipeline --min-seconds 0"
run() { name=$1; shift; for w in intra ld; do env "$@" timeout 300 python bench.py --workload $w $A 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$name', '$w', 'value %.1f enc %.1f dec %.1f' % (d['value'], d['encode_fps'], d['decode_fps']))"; done; }
run default X=1
run active_wait_500 ROC_ACTIVE_WAIT_TIMEOUT=500
run hsa_no_interrupt HSA_ENABLE_INTERRUPT=0
run both ROC_ACTIVE_WAIT_TIMEOUT=500 HSA_ENABLE_INTERRUPT=0
