"""Condense gpurun_out/prof_<tag>_<ctx>/ (tools/profile_round.sh output) into profiles/<tag>_<ctx>_{kernel_stats.csv, pmc.csv,
summary.md, digest.json, bench.json}.     python tools/summarize_profile.py <tag> <ctx>

Only this repo's kernels (ncf::*) are kept — torch's initialisation kernels have kilobyte-long names.
PMC handling follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE tallies 128-B read requests at 64 B, so the read side is DOUBLED before comparing with a byte count.
"""
import csv
import os
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(ncf::[A-Za-z0-9_]+(<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]


def main(tag, ctx=None):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "gpurun_out", f"prof_{tag}" + (f"_{ctx}" if ctx else ""))
    dst = os.path.join(root, "profiles")
    os.makedirs(dst, exist_ok=True)
    if ctx:
        tag = f"{tag}_{ctx}"
    cmd = "bench.py"
    try:
        cmd = open(os.path.join(src, "command.txt")).read().strip()
    except OSError:
        pass
    lines = [f"# rocprofv3 summary — {tag}", "",
             f"Command: `rocprofv3 --kernel-trace --stats -- python3 {cmd}`",
             "(plus one `--pmc` pass per counter group, each its own run; see tools/profile_round.sh).  Only `ncf::` kernels listed.", ""]
    stats = os.path.join(src, "trace", "trace_kernel_stats.csv")
    rows = []
    if os.path.exists(stats):
        with open(stats) as f:
            for r in csv.DictReader(f):
                if "ncf::" in r["Name"]:
                    rows.append(r)
        with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            for r in rows:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                            r["MinNs"], r["MaxNs"], r["StdDev"]])
        lines += ["## kernel stats (--kernel-trace --stats)", "",
                  "| kernel | calls | avg µs | min µs | max µs | % of GPU time |", "|---|---|---|---|---|---|"]
        for r in rows:
            lines.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | "
                         f"{float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} | {r['Percentage']} |")
        lines.append("")
    agg = defaultdict(lambda: defaultdict(list))
    for sub in sorted(os.listdir(src)) if os.path.isdir(src) else []:
        p = os.path.join(src, sub, "pmc_counter_collection.csv")
        if not os.path.exists(p):
            continue
        with open(p) as f:
            for r in csv.DictReader(f):
                if "ncf::" not in r["Kernel_Name"]:
                    continue
                k = short(r["Kernel_Name"])
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if agg:
        lines += ["## PMC counters (mean per launch)", "", "| kernel | counter | mean per launch | note |", "|---|---|---|---|"]
        with open(os.path.join(dst, f"{tag}_pmc.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "counter", "launches", "mean_per_launch"])
            for k, cs in sorted(agg.items()):
                for c, vals in sorted(cs.items()):
                    mean = sum(vals) / len(vals)
                    w.writerow([k, c, len(vals), f"{mean:.6g}"])
                    note = ""
                    if c == "FETCH_SIZE":
                        note = f"= {mean*1024*2/1e6:.2f} MB read per launch after the gfx950 x2 correction ({mean*1024/1e6:.2f} MB raw)"
                    elif c == "WRITE_SIZE":
                        note = f"= {mean*1024/1e6:.2f} MB written per launch"
                    lines.append(f"| `{k}` | {c} | {mean:.6g} | {note} |")
        lines.append("")
    # machine-readable digest for bench.py's roofline.traffic / rocprof cross-check
    import json
    digest = {"tag": tag, "kernels": {}}
    for r in rows:
        digest["kernels"].setdefault(short(r["Name"]), {})["rocprof_avg_us"] = float(r["AverageNs"]) / 1e3
    for k, cs in agg.items():
        d = digest["kernels"].setdefault(k, {})
        if "FETCH_SIZE" in cs:
            d["fetch_bytes"] = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]) * 1024 * 2  # gfx950: x2 (guide, HBM section)
        if "WRITE_SIZE" in cs:
            d["write_bytes"] = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"]) * 1024
        if "SQ_VALU_MFMA_BUSY_CYCLES" in cs:
            d["mfma_busy_cycles"] = sum(cs["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(cs["SQ_VALU_MFMA_BUSY_CYCLES"])
        if "GRBM_GUI_ACTIVE" in cs:
            d["gui_active_cycles_sum8xcd"] = sum(cs["GRBM_GUI_ACTIVE"]) / len(cs["GRBM_GUI_ACTIVE"])
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_LDS_BANK_CONFLICT", "SQ_BUSY_CYCLES", "SQ_WAVES"):
            if c in cs:
                d[c.lower()] = sum(cs[c]) / len(cs[c])
    # the bench line printed by the profiled trace run (its live timings were taken UNDER the profiler)
    bj = os.path.join(src, "bench.json")
    bench_line = None
    if os.path.exists(bj) and os.path.getsize(bj) > 0:
        try:
            bench_line = json.loads(open(bj).read().strip().splitlines()[-1])
            with open(os.path.join(dst, f"{tag}_bench_under_profiler.json"), "w") as f:
                json.dump(bench_line, f, indent=1)
        except ValueError:
            bench_line = None
    # cfg 4: one LightGCN layer = one spmm_seg_kernel launch per level of the segment tree (edge pass + ordered partial sums);
    # every layer has exactly one launch per level, so layer totals = per-launch means x levels
    if bench_line is not None:
        m = re.search(r"x(\d+) levels", str((bench_line.get("roofline") or {}).get("kernel", "")))
        seg = [k for k in digest["kernels"] if k.startswith("ncf::spmm_seg_kernel")]
        if m and seg:
            lv = int(m.group(1))
            d = digest["kernels"][seg[0]]
            digest["kernels"]["ncf::spmm_layer"] = {k: v * lv for k, v in d.items() if isinstance(v, (int, float))}
            digest["kernels"]["ncf::spmm_layer"]["levels"] = lv
            lines += [f"One LightGCN layer = {lv} `{seg[0]}` launches (edge pass + ordered partial-sum tree): per-layer figures = per-launch means x {lv}:",
                      "", "```", json.dumps(digest["kernels"]["ncf::spmm_layer"], indent=1), "```", ""]
    if digest["kernels"]:
        with open(os.path.join(dst, f"{tag}_digest.json"), "w") as f:
            json.dump(digest, f, indent=1)
    with open(os.path.join(dst, f"{tag}_summary.md"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r02", sys.argv[2] if len(sys.argv) > 2 else None)
