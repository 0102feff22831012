#!/bin/bash
# rocprofv3 passes over every bench context of a round (on the GPU box, from the repo root):
#   bash tools/profile_round.sh <tag> [contexts...]        default contexts: cfg2 cfg2zipf cfg2_emb128 cfg3 cfg4 cfg5 cfg5_b1048576 cfg5_b4194304
# Per context: pass 1 kernel trace + stats; passes 2-4 PMC counters, each in its own run (TCC slots: FETCH_SIZE and WRITE_SIZE do
# not fit one pass; no trace domains besides --kernel-trace beside --pmc).  The profiled command is the bench's own
# single-context form (in-process, no child processes).  Output: gpurun_out/prof_<tag>_<ctx>/; condensed into
# profiles/<tag>_<ctx>_* by tools/summarize_profile.py (run here as well, so the summaries travel back with gpurun_out/).
set -o pipefail
TAG=${1:-r03}; shift || true
CTXS=${@:-cfg2 cfg2zipf cfg2_emb128 cfg3 cfg4 cfg5 cfg5_b1048576 cfg5_b4194304}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd $ROOT
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
for C in $CTXS; do
  case $C in
    cfg2)     ARGS="bench.py --workload cfg2 --steps 60 --warmup 10 --no-cpu-baseline --no-variants" ;;
    cfg2zipf) ARGS="bench.py --workload cfg2 --steps 60 --warmup 10 --no-cpu-baseline --no-variants --zipf-users" ;;
    cfg5_b*)  export NCF_CFG5_LOCAL_BATCH=${C#cfg5_b}; ARGS="bench.py --workload cfg5 --steps 30 --warmup 5 --no-cpu-baseline --no-variants" ;;
    *)        unset NCF_CFG5_LOCAL_BATCH; ARGS="bench.py --workload $C --no-cpu-baseline --no-variants" ;;
  esac
  OUT=$ROOT/gpurun_out/prof_${TAG}_$C
  mkdir -p $OUT
  echo "$ARGS" > $OUT/command.txt
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $ARGS > $OUT/trace.log 2>&1 || echo "$C trace pass failed"
  grep -E '^\{' $OUT/trace.log > $OUT/bench.json || true
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1 || echo "$C fetch pass failed"
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 $ARGS > $OUT/pmc_write.log 2>&1 || echo "$C write pass failed"
  rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $OUT/pmc_sq -o pmc -- python3 $ARGS > $OUT/pmc_sq.log 2>&1 || echo "$C sq pass failed"
  python3 tools/summarize_profile.py $TAG $C > $OUT/summary.log 2>&1 || echo "$C summary failed"
  mkdir -p $ROOT/gpurun_out/profiles_$TAG && cp $ROOT/profiles/${TAG}_${C}_* $ROOT/gpurun_out/profiles_$TAG/ 2>/dev/null
  # the raw traces (one row per dispatch, kilobyte-long kernel names) stay on the box: gpurun_out/ travels back only under 64 MiB
  rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
  echo "$C done"
done
