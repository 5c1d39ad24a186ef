export TMPDIR=/tmp
R=$PWD; cd /tmp
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CU_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc7/$n -o c -- $R/tools/_bin/core_bench -r 2 -n 10 $R/dcvc_amd/libdcvc_amd.so > /tmp/pmc7_$n.log 2>&1 || echo "set failed: $set"
done
cd $R
python tools/pmc_summary.py /tmp/pmc7 2>/dev/null | grep -i "nsplit8_kernel<384, 384, 2, 1\|^==" | cut -c1-300
