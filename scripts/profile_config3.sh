#!/bin/bash
# rocprofv3 passes over BASELINE config 3 (the tiled kernels).  Usage (GPU box, repo root): bash scripts/profile_config3.sh <tag> [signals]
set -o pipefail
TAG=${1:-r02}
NS=${2:-2048}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof3_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
RUN="python3 $GRAFT_REPO_ROOT/scripts/run_config3.py $NS 2"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o c3 -- $RUN > $OUT/trace.log 2>&1 || exit 1
echo "trace done"
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o c3 -- $RUN > $OUT/pmc_sq.log 2>&1 || exit 2
echo "sq done"
timeout -s KILL 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_lds -o c3 -- $RUN > $OUT/pmc_lds.log 2>&1 || exit 3
echo "lds done"
find $OUT -name "*.csv" | head -30
