"""Print name / calls / average us of the kernels in a rocprofv3 kernel_stats.csv whose name matches any of the given substrings."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pats = sys.argv[2:]
for r in rows:
    if not pats or any(p in r["Name"] for p in pats):
        print("%-64s calls %5s  avg %9.1f us  %5s %%" % (r["Name"].replace("(anonymous namespace)::", "").split("(")[0][:64], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
