# Round 6: the first tile's rows fetched once by the eight waves together (this tree) against every wave its own pieces (tools/_bin/base.so)
B=tools/_bin
for pass in 1 2 3; do
for g in "136 240" "68 120" "270 480"; do
  for l in $B/base.so dcvc_amd/libdcvc_amd.so; do
      echo "=== pass $pass $l picture $g -w"
      timeout 120 $B/core_bench -r 3 -n 20 -c 256 -i 128 -g $g -w $l 2>&1 | grep "dcb_nsplit + next\|nsplit timeline" | cut -c1-160
  done
done; done
