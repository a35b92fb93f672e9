#!/bin/bash
# one extra PMC pass: how busy is the LDS in the level kernels?  Usage (GPU box): bash scripts/profile_lds.sh <tag>
set -o pipefail
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fp32-leg --steps 10 --warmup 2"
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc_lds -o bench -- $BENCH > $OUT/pmc_lds.log 2>&1 || exit 5
echo "lds done"
