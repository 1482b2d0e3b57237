#!/bin/bash
# Run on the GPU box (gpurun): PMC passes over tools/mb_wmsa.py (fused W-MSA block kernel at the bench shape, inference and
# training form).  Separate --pmc runs, program directly after "--".   Usage: tools/pmc_wmsa.sh <tag>
set -e
TAG=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P="python3 $GRAFT_REPO_ROOT/tools/mb_wmsa.py 8"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $OUT/a -- $P > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM_RD --kernel-trace --output-format csv -d $OUT/b -- $P > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/c -- $P > $OUT/c.log 2>&1
# (a TA_* / TCP_* pass did not finish within 7 minutes on this pool - the run was killed as silent; not collected)
ls $OUT/*/*/ | head
