#!/bin/bash
# round 6 against round 5 on ONE box in ONE session: bench.py's sequential loop with round 5's library (tools/_bin/r05.so =
# tools/build_variant.sh r05 --rev 6d43745, loaded through DCVC_LIB; the build / ABI check of this tree is skipped for it: it
# lacks the round-6 entry points) and with this tree's, interleaved, two passes
ARGS="--steps 60 --warmup 10 --no-cpu-baseline --no-uhd --no-resolutions --no-extras --no-roofline --no-pipeline --min-seconds 0"
for pass in 1 2; do
for w in intra ld hts htl; do
for l in r05 r06; do
  if [ $l = r05 ]; then export DCVC_LIB=$PWD/tools/_bin/r05.so; else unset DCVC_LIB; fi
  python -c "
import sys, runpy
import __graft_entry__
if '$l' == 'r05': __graft_entry__.build = lambda: None
sys.argv = ['bench.py'] + '$ARGS --workload $w'.split()
runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('pass $pass $w $l', round(d['value'],1), round(d['encode_fps'],1), round(d['decode_fps'],1), d['closure_ok'])"
done; done; done
