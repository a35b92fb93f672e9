#!/bin/bash
# rocprofv3 passes for profiles/: kernel trace + stats, then one PMC pass per counter group.
# Usage (on the GPU box, from the repo root): bash scripts/profile_r01.sh <tag>
set -o pipefail
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fp32-leg --steps 10 --warmup 2"
# the kernel trace runs long enough for the clocks to settle (its averages are compared with bench.py's live spans)
TRACE="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fp32-leg --steps 60 --warmup 30"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $TRACE > $OUT/trace.log 2>&1 || exit 1
echo "trace done"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- $BENCH > $OUT/pmc_fetch.log 2>&1 || exit 2
echo "fetch done"
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- $BENCH > $OUT/pmc_write.log 2>&1 || exit 3
echo "write done"
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -o bench -- $BENCH > $OUT/pmc_sq.log 2>&1 || exit 4
echo "sq done"
find $OUT -name "*.csv" | head -30
