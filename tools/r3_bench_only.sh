#!/bin/bash
# the default bench line once more (box-to-box variation of the same tree: 130 - 139 pictures/s)
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/r03_bench_line_again.json 2> gpurun_out/r03_bench_again.err
tail -1 gpurun_out/r03_bench_line_again.json | cut -c1-400
