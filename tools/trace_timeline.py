"""Timeline of one steady-state step from a rocprofv3 kernel_trace.csv: start (us, relative), queue, duration, kernel."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sgd = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("sgd_kernel")]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 20
lo, hi = sgd[k] - 6, sgd[k + 1] + 2
t0 = int(rows[lo]["Start_Timestamp"])
qs = sorted(set(r["Queue_Id"] for r in rows[lo:hi]))
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    q = qs.index(r["Queue_Id"])
    print("%9.1f %s%-2d %8.1f  %s" % (s / 1e3, "          " * q, q, (e - s) / 1e3, r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:60]))
