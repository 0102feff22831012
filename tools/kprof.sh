#!/bin/bash
# rocprofv3 kernel trace of one python tool (GPU box): bash tools/kprof.sh <tag> <script.py> [args...]  -> top kernels by total time
TAG=$1; shift
export TMPDIR=/tmp
rm -rf /tmp/kprof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kprof_$TAG -o p -- python3 "$@" > /tmp/kprof_$TAG.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("/tmp/kprof_$TAG/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:${KPROF_TOP:-12}]:
        print("$TAG", r["Name"][:90], r["Calls"], "avg %.1f us" % (float(r["AverageNs"]) / 1e3), "min %.1f" % (float(r["MinNs"]) / 1e3), "max %.1f" % (float(r["MaxNs"]) / 1e3))
PY
