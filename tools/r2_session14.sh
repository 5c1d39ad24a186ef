#!/bin/bash
# Round 2, GPU session 14: what are the copy / fill launches of the LD workload? (memory-copy trace + HIP API trace summary)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --stats --output-format csv -d $R/gpurun_out/s14 -o t -- python $R/bench.py --workload ld --steps 8 --warmup 2 --no-extras --no-roofline > $R/gpurun_out/s14.log 2>&1
cd $R
ls gpurun_out/s14 | head -20
python - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/s14/*memory_copy_trace.csv'):
    c=collections.Counter()
    for r in csv.DictReader(open(f)):
        size=int(r.get('Size', r.get('Bytes', 0)) or 0)
        b='<=64' if size<=64 else '<=4K' if size<=4096 else '<=1M' if size<=(1<<20) else '>1M'
        c[(r.get('Direction','?'), b)]+=1
    print(f); print(sorted(c.items(), key=lambda kv:-kv[1]))
for f in glob.glob('gpurun_out/s14/*hip_api_stats.csv'):
    print(f)
    for i,r in enumerate(csv.DictReader(open(f))):
        if i<25: print({k:r[k] for k in list(r)[:3]})
PY
find gpurun_out/s14 -name "*.csv" -size +2M -delete
