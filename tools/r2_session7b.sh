#!/bin/bash
# Round 2, GPU session 7b: SQ / LDS counters of dcb_core (stand-alone launches of tools/core_timeline.py), counters in
# their own passes (--pmc with --kernel-trace only)
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc7
R=$GRAFT_REPO_ROOT
cd /tmp
run() {
    name=$1; shift
    timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc7/$name -o $name -- python $R/tools/core_timeline.py > $R/gpurun_out/pmc7/$name.log 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES
run sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS
run sq3 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE
run sq4 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM
cd $R
python tools/pmc_summary.py gpurun_out/pmc7 2>&1 | grep -v "^ .*n=1 " | cut -c1-400 | tee gpurun_out/pmc7/summary.txt
find gpurun_out/pmc7 -name "*.csv" -size +2M -delete
