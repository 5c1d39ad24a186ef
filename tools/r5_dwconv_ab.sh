#!/bin/bash
# depthwise walk variants: us per launch (core_bench "dw3x3", 136 x 240, 384 and 128 channels), one process per variant
O=gpurun_out/r05b; mkdir -p $O
for v in "8,1" "8,2" "8,3" "4,1" "4,2" "16,1" "16,2"; do
  for ci in 384 128; do
    us=$(DCVC_DWCONV_VARIANT=$v timeout 120 tools/_bin/core_bench -r 3 -n 30 -c $([ $ci = 384 ] && echo 384 || echo 256) -i $ci dcvc_amd/libdcvc_amd.so 2>&1 | grep -o "dw3x3 *[0-9.]* us")
    echo "variant $v  channels $ci  $us"
  done
done | tee $O/dwconv_variants.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "dwconv" -s 2>&1 | grep -v "^$" | tail -14 | tee $O/tests_dwconv.log
