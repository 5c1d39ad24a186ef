#!/bin/bash
# round 6: ring depth of the block kernel's 32-pixel workgroups (NS8_RING1 = 8 / 12 / 16), one process per library and shape
B=tools/_bin
for sh in "512 512 8160" "768 768 8160" "384 192 8160" "256 128 8160" "256 256 8160" "512 512 2040"; do
  set -- $sh
  for l in ring8 ring12 ring16; do
    echo "=== $l C $1 CI $2 pixels $3"
    timeout 120 $B/core_bench -r 3 -n 20 -c $1 -i $2 -p $3 $B/$l.so 2>&1 | grep "dcb_nsplit + next" | grep -o "dcb_nsplit + next[^|]*|[^|]*"
  done
done
