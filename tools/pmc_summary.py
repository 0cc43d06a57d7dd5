"""Summarise a rocprofv3 --pmc CSV (counter_collection.csv) per kernel name."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"][:48]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k].add(r["Dispatch_Id"])
for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
    if "gemm" not in k and len(sys.argv) < 3:
        continue
    n = len(cnt[k])
    print(k, "dispatches", n)
    for c, v in sorted(d.items()):
        print("   %-28s %16.0f  per-dispatch %14.1f" % (c, v, v / n))
