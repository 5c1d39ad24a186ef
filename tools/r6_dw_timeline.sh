# Round 6: the (256, 128) block launch with / without its depthwise conv inside under core_bench: launch time, in-kernel stamps
L=dcvc_amd/libdcvc_amd.so; B=tools/_bin
for g in "136 240" "270 480"; do
  for w in "" "-w"; do
    echo "=== C 256 CI 128 picture $g $w"
    timeout 120 $B/core_bench -r 3 -n 20 -c 256 -i 128 -g $g $w $L 2>&1 | grep -v "^  dcb_core" | cut -c1-1500
  done
done
