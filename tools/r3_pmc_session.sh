#!/bin/bash
# round 3 PMC passes (counters in their own runs: --pmc with --kernel-trace only):
#   HBM traffic per launch of the bench's kernels (FETCH_SIZE / WRITE_SIZE, separate passes) -> gpurun_out/r03_hbm_traffic.json
#   SQ / TCC counters of the N-split block kernel on the stand-alone block bench (tools/probes/core_bench.hip)
set -x
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/pmc3
BENCH="python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-roofline --no-extras --no-uhd --min-seconds 0"
cd /tmp
timeout 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc3/bench_fetch -o bench_fetch -- $BENCH > $R/gpurun_out/pmc3/bench_fetch.log 2>&1
timeout 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc3/bench_write -o bench_write -- $BENCH > $R/gpurun_out/pmc3/bench_write.log 2>&1
cd $R
python tools/hbm_traffic.py /tmp/pmc3/bench_fetch /tmp/pmc3/bench_write gpurun_out/r03_hbm_traffic.json "$(cat .git_head 2>/dev/null)" | tail -40
CB="$R/tools/_bin/core_bench -r 1 -n 5 $R/dcvc_amd/libdcvc_amd.so"
run() {  # name, counters...
    name=$1; shift
    (cd /tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc3/$name -o $name -- $CB > $R/gpurun_out/pmc3/$name.log 2>&1)
}
run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES
run sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS
run sq3 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
python tools/pmc_summary.py /tmp/pmc3 > gpurun_out/pmc3/summary.txt 2>&1
grep -v "^  [a-z_A-Z:]*copy\|fill" gpurun_out/pmc3/summary.txt | cut -c1-400 | head -120
