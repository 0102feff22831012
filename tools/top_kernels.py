"""Dev tool: top kernels (any namespace) of a rocprofv3 --kernel-trace --stats run; names truncated."""
import csv, sys
csv.field_size_limit(1 << 30)
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 15]:
    print(f"{float(r['TotalDurationNs'])/tot*100:5.1f}%  calls {int(r['Calls']):6d}  avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:110]}")
