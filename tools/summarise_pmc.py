"""Summarise gpurun_out/<label>_trace (kernel stats) and <label>_pmc* (counter CSVs) into gpurun_out/<label>_summary.txt."""
import csv, glob, os, sys
L = sys.argv[1]
base = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
lines = []
for f in glob.glob(os.path.join(base, L + "_trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        lines.append("%-90s calls=%s avg_us=%.2f total_ms=%.3f pct=%s" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                   float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
agg = {}
for d in glob.glob(os.path.join(base, L + "_pmc*")):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:60]
            c = r["Counter_Name"]
            s = agg.setdefault(k, {}).setdefault(c, [0.0, 0])
            s[0] += float(r["Counter_Value"]); s[1] += 1
for k, cs in agg.items():
    if "mvm" not in k and "ski" not in k and "bilinear" not in k:
        continue
    lines.append("== " + k)
    for c, (s, n) in sorted(cs.items()):
        lines.append("   %-28s mean per launch %.6g  (n=%d)" % (c, s / n, n))
open(os.path.join(base, L + "_summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
