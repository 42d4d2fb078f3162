#!/bin/bash
# Round 4: SQ counters of the 64 x 128 conv kernel on a launch that fills its rounds (3 x 192 x 256, 128 -> 128, 3 x 3)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r4_r
mkdir -p $O
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA" "SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_VALU"; do
  i=$((i+1))
  rm -rf /tmp/pmc_sq$i
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_sq$i -o pmc -- python $R/tools/runs/r4_conv_one.py > $O/pmc_sq$i.log 2>&1
  python $R/tools/pmc_dump.py "/tmp/pmc_sq$i/**/*.db" 2>&1 | grep "k_conv_igemm" | cut -c1-200 | tee -a $O/pmc_sq.txt
done
