#!/bin/bash
# round 3, session 16: the previous pass's epilogue dealt out between the MFMAs with sched_group_barrier (NS_MIX VALU per MFMA)
set -x
mkdir -p gpurun_out
B=tools/_bin
L=dcvc_amd/libdcvc_amd.so
{ timeout 300 $B/core_bench -r 5 -n 20 $B/mix0.so $B/mix5.so $L $B/mix12.so
  timeout 300 $B/core_bench -r 5 -n 20 -c 512 -i 256 $B/mix0.so $B/mix5.so $L $B/mix12.so
  timeout 300 $B/core_bench -r 5 -n 20 -c 512 -i 512 $B/mix0.so $B/mix5.so $L $B/mix12.so
  timeout 300 $B/core_bench -r 5 -n 20 -c 256 -i 256 $B/mix0.so $B/mix5.so $L $B/mix12.so ; } > gpurun_out/core_bench16.txt 2>&1
grep -v "^  timeline" gpurun_out/core_bench16.txt | cut -c1-1000
