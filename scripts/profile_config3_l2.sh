#!/bin/bash
# L2 view of BASELINE config 3 (does the bank stream hit in L2?).  Usage (GPU box, repo root): bash scripts/profile_config3_l2.sh <tag> [signals]
set -o pipefail
TAG=${1:-r02}
NS=${2:-2048}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof3l2_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
RUN="python3 $GRAFT_REPO_ROOT/scripts/run_config3.py $NS 2"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o c3 -- $RUN > $OUT/pmc_fetch.log 2>&1 || exit 1
echo "fetch done"
timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/pmc_tcc -o c3 -- $RUN > $OUT/pmc_tcc.log 2>&1 || exit 2
echo "tcc done"
timeout -s KILL 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_tcp -o c3 -- $RUN > $OUT/pmc_tcp.log 2>&1 || exit 3
echo "tcp done"
