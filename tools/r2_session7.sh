#!/bin/bash
# Round 2, GPU session 7: fragment loads fenced in front of the MFMAs (dcb_core), bench A/B of one vs two codec objects
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "dcb_core" 2>&1 | tail -3 ) | tee gpurun_out/s7_core_test.log
timeout 120 python tools/core_timeline.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s7_timeline.txt
for mode in "" "--one-codec"; do
  timeout 300 python bench.py --no-cpu-baseline --no-extras --no-roofline $mode > gpurun_out/s7_bench$mode.json 2> gpurun_out/s7_bench$mode.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/s7_bench$mode.json").read().splitlines()[-1])
print("$mode", "value %.1f  enc %.1f dec %.1f  ms/step %.2f" % (d["value"], d["encode_fps"], d["decode_fps"], d["ms_per_step"]))
PY
done
