#!/bin/bash
for d in 0 1 2 3 4 7; do echo "== dbg $d"; DCVC_CORE_DBG=$d timeout 100 python tools/core_timeline.py 2>&1 | grep -E "dcb_core \+|dc.3 \(18|super-chunk 3|next dc.0|total"; done
