#!/bin/bash
# Kernel-trace + stats of the secondary workloads (cfg3 / cfg4 / cfg5).  Usage: bash tools/profile_extra.sh <tag>
set -e -o pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd $ROOT
for W in cfg3 cfg4 cfg5; do
  OUT=$ROOT/gpurun_out/prof_${TAG}_$W
  mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py --workload $W > $OUT/trace.log 2>&1
  grep -E '^\{' $OUT/trace.log > $OUT/bench.json || true
  echo "$W done"
done
