#!/bin/bash
# Kernel-trace + stats and PMC (FETCH_SIZE, WRITE_SIZE; separate passes) of the secondary workloads.
# Usage: bash tools/profile_extra.sh <tag> [workloads...]
set -e -o pipefail
TAG=${1:-r01}; shift || true
WL=${@:-cfg3 cfg4 cfg5}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd $ROOT
for W in $WL; do
  OUT=$ROOT/gpurun_out/prof_${TAG}_$W
  mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py --workload $W --no-cpu-baseline > $OUT/trace.log 2>&1
  grep -E '^\{' $OUT/trace.log > $OUT/bench.json || true
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 bench.py --workload $W --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 bench.py --workload $W --no-cpu-baseline > $OUT/pmc_write.log 2>&1
  echo "$W done"
done
