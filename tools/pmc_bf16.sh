#!/bin/bash
# PMC passes over one bf16 scoring kernel: tools/pmc_bf16.sh <tag> <kernel option> [B]
set -o pipefail
TAG=${1:-bf16}; K=${2:-ws8}; B=${3:-1048576}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
P2="SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE"
P3="SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_WAVES"
n=1
for P in "$P1" "$P2" "$P3"; do
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$n -o pmc -- python3 tools/run_bf16_once.py $K $B 12 > $OUT/p$n.log 2>&1 || echo "pass $n failed"
  n=$((n+1))
done
PMC_OUT=$OUT python3 - > $OUT/summary.md <<'PY'
import csv, glob, collections, os
csv.field_size_limit(1<<30)
agg=collections.defaultdict(list); dur=[]
for f in glob.glob(os.environ["PMC_OUT"] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "score_ws" not in r["Kernel_Name"] and "score_fused_bf16" not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(os.environ["PMC_OUT"] + "/p*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "score_ws" in r["Kernel_Name"] or "score_fused_bf16" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("| counter | mean per launch |"); print("|---|---|")
for c in sorted(agg): print(f"| {c} | {sum(agg[c])/len(agg[c]):.4g} |")
if dur: print(f"\nkernel duration under the counters: mean {sum(dur)/len(dur):.1f} us over {len(dur)} launches")
PY
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
cat $OUT/summary.md
