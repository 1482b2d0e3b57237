#!/bin/bash
# PMC passes over the fused W-MSA microbenchmark (program directly after --, counters only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for p in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" \
         "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT"; do
  n=$(echo $p | cut -d' ' -f1)
  rocprofv3 --pmc $p --output-format csv -d $R/gpurun_out/pmc_wmsa_$n -- python3 $R/tools/mb_wmsa.py 8 > $R/gpurun_out/pmc_wmsa_$n.log 2>&1
done
