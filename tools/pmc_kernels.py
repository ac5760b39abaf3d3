"""Per-kernel means of rocprofv3 --pmc counters.  usage: pmc_kernels.py <dir> [name filter ...]  (prints a table)"""
import csv, glob, os, sys, collections
d, filt = sys.argv[1], sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
        if filt and not any(x in n for x in filt):
            continue
        key = (n, r.get("Grid_Size", r.get("Grid_Size_X", "")))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key in sorted(agg):
    print("%s grid=%s" % key)
    for c in sorted(agg[key]):
        v = agg[key][c]
        print("    %-28s %14.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
