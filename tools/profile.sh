#!/bin/bash
# Usage (on the GPU box, from the repo root):  bash tools/profile.sh <tag> [bench args...]
# Pass 1: kernel trace + stats.  Passes 2-4: PMC counters, each in its own run (TCC slots: FETCH_SIZE and
# WRITE_SIZE do not fit one pass).  Output: gpurun_out/prof_<tag>/...; summarise with tools/summarize_profile.py.
set -e -o pipefail
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
ARGS="bench.py --steps 60 --warmup 10 --no-cpu-baseline $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $ARGS > $OUT/trace.log 2>&1
echo "pass trace done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
echo "pass fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 $ARGS > $OUT/pmc_write.log 2>&1
echo "pass write done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o pmc -- python3 $ARGS > $OUT/pmc_mfma.log 2>&1
echo "pass mfma done"
find $OUT -name "*.csv" | head -30
