#!/bin/bash
# Round 2, GPU session 8: the whole GPU suite (incl. regenerated full-size digests, 4K, CLI, fan-out) + smoke()
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q --durations=12 2>&1 | tail -32 ) | tee gpurun_out/s8_test_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -4 ) | tee gpurun_out/s8_smoke.log
