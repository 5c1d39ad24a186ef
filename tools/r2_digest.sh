#!/bin/bash
export TMPDIR=/tmp
echo "== fused core"; ( timeout 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -q 2>&1 | tail -22 )
echo "== DCVC_NO_DCB_CORE=1"; ( DCVC_NO_DCB_CORE=1 timeout 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -k "dmci" 2>&1 | tail -16 )
