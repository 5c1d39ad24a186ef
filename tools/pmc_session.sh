#!/bin/bash
# PMC passes over the contraction micro-benchmark and the bench (rocprofv3, counters in their own
# runs: --pmc with --kernel-trace only). Outputs CSVs under gpurun_out/pmc/.
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
SHAPES="32640,384,1536,3 32640,384,384,4 32640,384,384,1 8160,512,2048,3 8160,512,512,4"
run() {  # name, counters...
    name=$1; shift
    REPS=5 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc/$name -o $name -- python tools/gemm_bench.py $SHAPES > gpurun_out/pmc/$name.log 2>&1
}
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES
run sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS
run sq3 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE
# bench-level HBM traffic of the dominant kernel
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc/bench_fetch -o bench_fetch -- python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/pmc/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc/bench_write -o bench_write -- python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/pmc/bench_write.log 2>&1
ls -R gpurun_out/pmc | head -50
python tools/pmc_summary.py gpurun_out/pmc > gpurun_out/pmc/summary.txt 2>&1
cat gpurun_out/pmc/summary.txt | head -80
# keep only small files
find gpurun_out/pmc -name "*.csv" -size +3M -delete
