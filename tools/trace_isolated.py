"""Per-kernel average duration inside the timed (two-stream) steps vs the isolated pass (side stream off, last 5 steps of
bench.py) from a rocprofv3 kernel_trace.csv: shows which kernels are slowed by sharing the chip and which are slow alone."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sgd = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("sgd_kernel")]
iso = rows[sgd[-6] + 1:sgd[-1] + 1]                 # 5 isolated steps
con = rows[sgd[-6 - 1 - 20] + 1:sgd[-6 - 1] + 1]    # 20 timed steps before the switch step
def avg(sel):
    t, c = collections.Counter(), collections.Counter()
    for r in sel:
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        t[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); c[k] += 1
    return {k: (t[k] / c[k] / 1e3, c[k]) for k in t}
a, b = avg(con), avg(iso)
print("%-52s %10s %10s %7s" % ("kernel", "in-step us", "alone us", "ratio"))
for k in sorted(a, key=lambda k: -a[k][0] * a[k][1]):
    if k in b:
        print("%-52s %10.1f %10.1f %7.2f   x%.1f/step" % (k[:52], a[k][0], b[k][0], a[k][0] / b[k][0], a[k][1] / 20.0))
