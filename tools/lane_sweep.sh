#!/bin/bash
# bench.py for every workload at LANES (default 0 = each workload's default lane count); JSON lines -> gpurun_out/lanes/<workload>_<lanes>.json
mkdir -p gpurun_out/lanes
run() {
    python bench.py --workload $1 --lanes $2 2> gpurun_out/lanes/$1_$2.err | tail -1 > gpurun_out/lanes/$1_$2.json
    python - <<PY
import json
d = json.load(open("gpurun_out/lanes/$1_$2.json"))
print("$1 lanes", d["config"]["lanes"], round(d["value"], 1), "ms/step", round(d["ms_per_step"], 2),
      "one_lane", round(d.get("one_lane", {}).get("value", 0), 1), "gemm TF", round(d["roofline"]["achieved"], 1))
PY
}
for w in ${WORKLOADS:-intra ld hts htl}; do for l in ${LANES:-0}; do run $w $l; done; done
