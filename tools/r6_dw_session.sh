# Round 6, depthwise conv inside the (256, 128) block launch: parity, then the codecs with and without it on the same box
# usage (GPU box): bash tools/r6_dw_session.sh [quick]
mkdir -p gpurun_out/r06dw
O=gpurun_out/r06dw
rocminfo | grep -m1 "Marketing Name.*MI3" > $O/box.txt; cat /proc/cpuinfo | grep -m1 "model name" >> $O/box.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "depthwise_inside or dcb_nsplit" > $O/test_dw.log 2>&1; tail -3 $O/test_dw.log
if [ "$1" != "quick" ]; then
  timeout 1500 python -m pytest tests -x -q -m gpu > $O/test_gpu.log 2>&1; tail -3 $O/test_gpu.log
fi
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-uhd --no-extras --no-roofline --no-pipeline --no-resolutions --min-seconds 0"
for pass in 1 2 3; do
for w in ld; do
for g in 0 1; do
  DCVC_NSPLIT_DW=$g timeout 300 $B --workload $w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('pass $pass $w dw_inside=$g', round(d['value'],1), 'enc', round(d['encode_fps'],1), 'dec', round(d['decode_fps'],1))" | tee -a $O/ab.txt
done; done; done
for g in 0 1; do
  DCVC_NSPLIT_DW=$g timeout 300 $B --workload ld --resolution 3840x2160 --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ld 3840x2160 dw_inside=$g', round(d['value'],1), 'enc', round(d['encode_fps'],1), 'dec', round(d['decode_fps'],1))" | tee -a $O/ab.txt
done
