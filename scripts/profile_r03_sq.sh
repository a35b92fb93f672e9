#!/bin/bash
# Instruction mix / wait view of the default bench step (config 2), round 3.  Usage (GPU box, repo root): bash scripts/profile_r03_sq.sh <tag>
set -o pipefail
TAG=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof2sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
RUN="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fp32-leg --no-config5 --no-configs --steps 10 --warmup 2"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_a -o c2 -- $RUN > $OUT/pmc_a.log 2>&1 || exit 1
echo "a done"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_b -o c2 -- $RUN > $OUT/pmc_b.log 2>&1 || exit 2
echo "b done"
python3 $GRAFT_REPO_ROOT/scripts/pmc_quick.py $OUT > $OUT/summary.json
