#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 kernel stats + PMC passes of bench.py (separate --pmc runs, program directly after --).
# Usage: tools/collect_profiles.sh <tag>      -> gpurun_out/prof_<tag>/{stats,fetch,write,mfma}
set -e
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $OUT/write.log 2>&1
# MFMA utilisation: matrix-pipe busy cycles against the cycles the GPU was active (GRBM_GUI_ACTIVE sums the 8 XCDs)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $OUT/mfma.log 2>&1
grep -h metric $OUT/stats.log | cut -c1-200
