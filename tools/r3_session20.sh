#!/bin/bash
# round 3, session 20: LD, dcb_tail forced on small grids, with / without the fused FFN forced for the (384, 192) blocks at / 16
set -x
for f in 1 2 1 2; do
  DCVC_DCB_TAIL=2 DCVC_FFN_FUSED=$f timeout 300 python bench.py --workload ld --steps 60 --warmup 10 --no-cpu-baseline --no-uhd --no-extras --no-roofline 2>/dev/null | tail -1 | cut -c1-240
done
