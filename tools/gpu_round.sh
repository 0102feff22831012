#!/bin/bash
# One GPU-box call: GPU tests, the default bench line, and a 2-rank rehearsal of `bench.py --gpus 2` on the one GPU
# (gloo transport, both ranks on cuda:0).  Logs under gpurun_out/<tag>/.
set -o pipefail
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee $OUT/pytest.rc
tail -5 $OUT/pytest.log
timeout -k 10 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
tail -c 600 $OUT/bench.err
NCF_BENCH_SINGLE_DEVICE=1 NCF_BENCH_BACKEND=gloo timeout -k 10 900 python bench.py --gpus 2 --steps 100 --warmup 10 > $OUT/bench_2rank.json 2> $OUT/bench_2rank.err; echo "bench 2-rank rc=$?"
tail -c 600 $OUT/bench_2rank.err
