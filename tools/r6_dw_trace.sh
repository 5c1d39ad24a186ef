# Round 6: kernel trace of the LD codec with and without the depthwise conv inside the (256, 128) block launch (same box)
R=$(pwd); O=$R/gpurun_out/r06dw; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for g in 0 1; do
  DCVC_NSPLIT_DW=$g timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profdw_$g -o t -- python $R/bench.py --workload ld --steps 10 --warmup 3 --no-cpu-baseline --no-uhd --no-resolutions --no-extras --no-roofline --no-pipeline --min-seconds 0 > $O/prof_ld_$g.log 2>&1
  python $R/tools/trace_after_setup.py /tmp/profdw_$g --marker mask_step_enc --per 2 > $O/ld_per_picture_dw$g.txt 2>&1
  head -24 $O/ld_per_picture_dw$g.txt | cut -c1-110
done
for g in 0 1; do
  DCVC_NSPLIT_DW=$g timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profdw4k_$g -o t -- python $R/bench.py --workload ld --resolution 3840x2160 --steps 6 --warmup 2 --no-cpu-baseline --no-uhd --no-resolutions --no-extras --no-roofline --no-pipeline --min-seconds 0 > $O/prof_ld4k_$g.log 2>&1
  python $R/tools/trace_after_setup.py /tmp/profdw4k_$g --marker mask_step_enc --per 2 > $O/ld4k_per_picture_dw$g.txt 2>&1
done
