# Round 6: quick loop for the depthwise-inside block launch: parity of the kernel tests, LD traces at both resolutions with / without
R=$(pwd); O=$R/gpurun_out/r06dw; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "depthwise_inside" 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp
rm -f $O/quick.txt
for g in 1 0; do
  for res in 1920x1080 3840x2160; do
  DCVC_NSPLIT_DW=$g timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profq_${g}_$res -o t -- python $R/bench.py --workload ld --resolution $res --steps 6 --warmup 2 --no-cpu-baseline --no-uhd --no-resolutions --no-extras --no-roofline --no-pipeline --min-seconds 0 > $O/prof_q_${g}_$res.log 2>&1
  echo "== dw_inside=$g $res" >> $O/quick.txt
  python $R/tools/trace_after_setup.py /tmp/profq_${g}_$res --marker mask_step_enc --per 2 2>&1 | grep "nsplit8_kernel\|dwconv\|^all" | cut -c1-100 >> $O/quick.txt
  done
done
cat $O/quick.txt
