# Round 6: HIP_FORCE_DEV_KERNARG (kernel arguments in device memory) - does the launches' ramp depend on it? core_bench stamps + the codecs' loops
B=tools/_bin
for k in 0 1; do
  echo "== HIP_FORCE_DEV_KERNARG=$k core_bench (256, 128) 136 x 240"
  HIP_FORCE_DEV_KERNARG=$k timeout 120 $B/core_bench -r 3 -n 20 -c 256 -i 128 -g 136 240 dcvc_amd/libdcvc_amd.so 2>&1 | grep "dcb_nsplit + next\|timeline" | cut -c1-200
done
BB="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-uhd --no-extras --no-roofline --no-pipeline --no-resolutions --min-seconds 0"
for pass in 1 2; do for w in ld intra; do for k in 0 1; do
  HIP_FORCE_DEV_KERNARG=$k timeout 300 $BB --workload $w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('pass $pass $w HIP_FORCE_DEV_KERNARG=$k', round(d['value'],1), 'enc', round(d['encode_fps'],1), 'dec', round(d['decode_fps'],1))"
done; done; done
