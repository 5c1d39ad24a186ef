#!/bin/bash
# round 6 experiment: TWO 32-pixel workgroups per CU (128 registers, ring of 4) against one 64-pixel workgroup (base)
B=tools/_bin
for pass in 1 2; do
for sh in "256 128 32640" "256 256 32640" "256 128 129600" "256 128 8160" "256 256 8160"; do
  set -- $sh
  echo "=== pass $pass base C $1 CI $2 pixels $3"
  timeout 120 $B/core_bench -r 3 -n 20 -c $1 -i $2 -p $3 $B/base.so 2>&1 | grep "dcb_nsplit + next" | grep -o "dcb_nsplit + next[^|]*|[^|]*"
  echo "=== pass $pass base-px32 C $1 CI $2 pixels $3"
  DCVC_NSPLIT_PX=32 timeout 120 $B/core_bench -r 3 -n 20 -c $1 -i $2 -p $3 $B/base.so 2>&1 | grep "dcb_nsplit + next" | grep -o "dcb_nsplit + next[^|]*|[^|]*"
  echo "=== pass $pass occ2-px32 C $1 CI $2 pixels $3"
  DCVC_NSPLIT_PX=32 timeout 120 $B/core_bench -r 3 -n 20 -c $1 -i $2 -p $3 $B/occ2.so 2>&1 | grep "dcb_nsplit + next" | grep -o "dcb_nsplit + next[^|]*|[^|]*"
done; done
