#!/bin/bash
# Run on the GPU box (gpurun): memory-path counters of the fused W-MSA block kernel, ONE inference + ONE training launch per
# process (tools/pmc_one.py), at most four counters of ONE hardware block per --pmc pass, program directly after "--", every
# pass under its own timeout with its log kept (a TA/TCP pass over a many-launch loop was killed as silent
# and a 9-counter mixed pass aborted with 'Request exceeds the capabilities of the hardware').   Usage: tools/pmc_mem.sh <tag>
TAG=${1:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcmem_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_available.txt 2>&1
P="python3 $GRAFT_REPO_ROOT/tools/pmc_one.py 2"
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  echo "== pass $i: $grp" | tee -a $OUT/passes.log
  timeout -k 10 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- $P > $OUT/p$i.log 2>&1
  echo "   rc=$?" | tee -a $OUT/passes.log
done <<'LIST'
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
TCC_READ_sum TCC_WRITE_sum
TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
TCP_GATE_EN1_sum TCP_GATE_EN2_sum
TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
TA_TA_BUSY_sum TA_BUSY_avr
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum
TA_BUFFER_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum
TD_TD_BUSY_sum TD_TC_STALL_sum
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
SQ_BUSY_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY
GRBM_GUI_ACTIVE
LIST
ls $OUT
