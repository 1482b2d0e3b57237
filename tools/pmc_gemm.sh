#!/bin/bash
# Run on the GPU box (gpurun): HBM traffic and SQ / TCC counters of the pipelined bf16 NT GEMM at four bench shapes, one launch per
# shape and process (tools/pmc_gemm.py); one --pmc group per pass, program directly after "--".   Usage: tools/pmc_gemm.sh <tag>
TAG=${1:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcgemm_${TAG}_${2:-pmc_gemm}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P="python3 $GRAFT_REPO_ROOT/tools/${2:-pmc_gemm}.py"
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  echo "== pass $i: $grp" | tee -a $OUT/passes.log
  timeout -k 10 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- $P > $OUT/p$i.log 2>&1
  echo "   rc=$?" | tee -a $OUT/passes.log
done <<'LIST'
FETCH_SIZE
WRITE_SIZE
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS
SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC
GRBM_GUI_ACTIVE
LIST
ls $OUT
