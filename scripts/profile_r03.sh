#!/bin/bash
# rocprofv3 passes for profiles/ (round 3): kernel trace + stats of the default bench step and of the fp32-only leg,
# then one PMC pass per counter group.  Usage (GPU box, repo root): bash scripts/profile_r03.sh <tag>
set -o pipefail
TAG=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (--streams 1: the kernels with the chip to themselves, what bench.py's roofline quotes; the default command, two
# streams, is traced once more at the end)
B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fp32-leg --no-config5 --no-configs --streams 1"
B2="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fp32-leg --no-config5 --no-configs"
BENCH="$B --steps 10 --warmup 2"
# the kernel traces run long enough for the clocks to settle (their averages are compared with bench.py's live spans)
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $B --steps 200 --warmup 100 > $OUT/trace.log 2>&1 || exit 1
echo "trace done"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_fp32 -o bench -- $B --fp32-only --steps 40 --warmup 10 > $OUT/trace_fp32.log 2>&1 || exit 1
echo "fp32 trace done"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- $BENCH > $OUT/pmc_fetch.log 2>&1 || exit 2
echo "fetch done"
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- $BENCH > $OUT/pmc_write.log 2>&1 || exit 3
echo "write done"
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o bench -- $BENCH > $OUT/pmc_sq.log 2>&1 || exit 4
echo "sq done"
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq_fp32 -o bench -- $BENCH --fp32-only > $OUT/pmc_sq_fp32.log 2>&1 || exit 5
echo "sq fp32 done"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_two_streams -o bench -- $B2 --steps 200 --warmup 100 > $OUT/trace_two_streams.log 2>&1 || exit 6
echo "two-stream trace done"
find $OUT -name "*.csv" | head -40
