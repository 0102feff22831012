#!/bin/bash
# PMC passes over the cfg-3 grouped attention kernel (both forms): tools/pmc_attn.sh <tag>
set -o pipefail
TAG=${1:-attn}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
P2="SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SMEM"
P3="SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_WAVES"
n=1
for P in "$P1" "$P2" "$P3"; do
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$n -o pmc -- python3 tools/ab_attn_grouped.py pmc > $OUT/p$n.log 2>&1 || echo "pass $n failed"
  n=$((n+1))
done
mkdir -p $ROOT/gpurun_out/profiles_r02
PMC_OUT=$OUT python3 - > $ROOT/gpurun_out/profiles_r02/r02_attn_before_after.md <<'PY'
import csv, glob, collections, os
csv.field_size_limit(1<<30)
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.environ["PMC_OUT"] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "attn_grouped" not in k: continue
        k="sc" if "grouped_sc" in k else "lds"
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# Grouped attention kernel at config 3 (4096 pairs, 64 users x 256 rated, A = 128, Fdim = 64, 32 pairs per workgroup):")
print("# round 1's form (`attn_grouped_kernel`, operands broadcast from LDS) vs the scalar-operand form (`attn_grouped_sc_kernel`)")
print()
print("Command: `rocprofv3 --kernel-trace --pmc <8 SQ counters per pass> -- python3 tools/ab_attn_grouped.py pmc` (tools/pmc_attn.sh; 20 launches")
print("of each kernel; means per launch; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles).")
print()
names = sorted(set(c for k in agg for c in agg[k]))
print("| counter | lds form (round 1) | scalar-operand form | ratio |")
print("|---|---|---|---|")
for c in names:
    a = agg.get("lds", {}).get(c); b = agg.get("sc", {}).get(c)
    fa = sum(a)/len(a) if a else float("nan"); fb = sum(b)/len(b) if b else float("nan")
    print(f"| {c} | {fa:.4g} | {fb:.4g} | {fb/fa if fa else float('nan'):.2f} |")
PY
cat $ROOT/gpurun_out/profiles_r02/r02_attn_before_after.md
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
