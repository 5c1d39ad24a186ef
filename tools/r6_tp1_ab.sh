# Round 6: ffn.0 of the (256, 128) blocks in passes of one tile (NS8_TP1) against passes of two, core_bench, one process per library
B=tools/_bin
for pass in 1 2; do
for g in "136 240" "270 480" "68 120"; do
  for l in dcvc_amd/libdcvc_amd.so $B/tp1.so; do
    for w in "" "-w"; do
      echo "=== pass $pass $l picture $g $w"
      timeout 120 $B/core_bench -r 3 -n 20 -c 256 -i 128 -g $g $w $l 2>&1 | grep "dcb_nsplit + next\|nsplit timeline" | cut -c1-260
    done
  done
done; done
