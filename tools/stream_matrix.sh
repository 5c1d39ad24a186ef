#!/bin/bash
# First GPU call of the next round (profiles/README.md v5 left these unmeasured): every workload x
# {caller on a non-default stream (harness), legacy null stream with the blocking join, null stream
# with the plain event wait} x {hipGraph stages, eager launches}, bench.py --no-roofline each.
# ~25 s per cell; WORKLOADS / CELLS narrow it. Output: gpurun_out/stream_matrix.txt
mkdir -p gpurun_out
out=gpurun_out/stream_matrix.txt
: > $out
cell() {   # workload user_stream join eager
    local tag="$1 user=$2 join=$3 eager=$4"
    local line
    line=$(DCVC_BENCH_USER_STREAM=$2 DCVC_NULL_STREAM_JOIN=$3 DCVC_BENCH_EAGER=$4 \
           python bench.py --workload $1 --no-roofline --no-cpu-baseline 2>/dev/null | tail -1)
    python - "$tag" "$line" <<'PY' | tee -a $out
import json, sys
try:
    d = json.loads(sys.argv[2])
    print("%-46s %8.1f pictures/s  %7.2f ms/step" % (sys.argv[1], d["value"], d["ms_per_step"]))
except Exception as e:
    print("%-46s FAILED (%s)" % (sys.argv[1], e))
PY
}
for w in ${WORKLOADS:-intra ld hts htl}; do
    cell $w side blocking ""
    cell $w null blocking ""
    cell $w null event ""
    cell $w side blocking 1
    cell $w null blocking 1
done
echo "LD with hipGraph stages (default: eager):" | tee -a $out
DCVC_BENCH_GRAPHS=1 cell ld side blocking ""
DCVC_BENCH_GRAPHS=1 cell ld null blocking ""
