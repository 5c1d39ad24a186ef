# Round 6: where the cost of the depthwise conv inside the block launch sits - timing variants (wrong results) under the LD trace.
# (The switches it was run with - build_variant.sh dw_nodma -DNS8_DW_NO_DMA, dw_nocomp -DNS8_DW_NO_COMPUTE, dw_none with both - were in the
# kernel for that session only (git history: the commit before "tiles in XCD bands"); the result is profiles/r06_dw_variants.txt.)
R=$(pwd); O=$R/gpurun_out/r06dw; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for v in "" dw_nodma dw_nocomp dw_none; do
  for res in 1920x1080 3840x2160; do
  L=""; [ -n "$v" ] && L=$R/tools/_bin/$v.so
  DCVC_LIB=$L timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profv_${v}_$res -o t -- python $R/bench.py --workload ld --resolution $res --steps 6 --warmup 2 --no-cpu-baseline --no-uhd --no-resolutions --no-extras --no-roofline --no-pipeline --min-seconds 0 > $O/prof_v_${v}_$res.log 2>&1
  echo "== variant '$v' $res" >> $O/variants.txt
  python $R/tools/trace_after_setup.py /tmp/profv_${v}_$res --marker mask_step_enc --per 2 2>&1 | grep "nsplit8_kernel<256, 128, 2\|^all" | cut -c1-100 >> $O/variants.txt
  done
done
cat $O/variants.txt
