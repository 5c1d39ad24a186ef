#!/bin/bash
# round 3, session 19: LD with dcb_tail forced for the small (128, 64) blocks (DCVC_DCB_TAIL=2) against the launch sequence
set -x
mkdir -p gpurun_out
for t in 1 2 1 2; do
  DCVC_DCB_TAIL=$t timeout 300 python bench.py --workload ld --steps 60 --warmup 10 --no-cpu-baseline --no-uhd --no-extras --no-roofline 2>/dev/null | tail -1 | cut -c1-240
done
