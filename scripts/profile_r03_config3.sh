#!/bin/bash
# rocprofv3 passes over BASELINE config 3 (scripts/run_config3.py, 4096 signals) for profiles/r03_config3_*:
# kernel trace + stats, then one PMC pass for the matrix-core counters.  Usage (GPU box, repo root): bash scripts/profile_r03_config3.sh
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r03_config3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R="python3 $GRAFT_REPO_ROOT/scripts/run_config3.py 4096"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o c3 -- $R 5 > $OUT/trace.log 2>&1 || exit 1
echo "trace done"
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o c3 -- $R 1 > $OUT/pmc_sq.log 2>&1 || exit 2
echo "sq done"
find $OUT -name "*.csv" | head
