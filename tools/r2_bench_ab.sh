#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 200 python bench.py --no-cpu-baseline --steps 60 > gpurun_out/ab_core.json 2> gpurun_out/ab_core.err
DCVC_NO_DCB_CORE=1 timeout 200 python bench.py --no-cpu-baseline --steps 60 > gpurun_out/ab_nocore.json 2> gpurun_out/ab_nocore.err
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/ab_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-36s %8.1f pictures/s  %.2f ms/step" % (f, d["value"], d["ms_per_step"]))
    except Exception as e:
        print(f, "unreadable:", e, open(f.replace(".json", ".err")).read()[-500:])
PY
( timeout 600 python -m pytest tests/test_fullsize_gpu.py tests/test_dmci_gpu.py -m gpu -q -x 2>&1 | tail -8 )
