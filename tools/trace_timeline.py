"""Timeline of one steady-state step from a rocprofv3 kernel_trace.csv: start (us, relative), queue, duration, kernel."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step = from a few kernels before its first forward GEMM to a few after the next step's (the optimiser kernel is no step marker any more:
# the scheduled update launches one per backward stage); steps are counted by their loss-rows kernel
rowsk = [i for i, r in enumerate(rows) if "margin_softmax_rows" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 20


def first_forward(after):
    for i in range(after, len(rows)):
        n = rows[i]["Kernel_Name"]
        if ("xv_gemm_nt" in n and "<true" in n) or "xv_gemm16_nt_kernel<true" in n or "xv_gemm16_nt_conv_kernel<1" in n:      # (forward = the launches with BatchNorm statistics)
            return i
    return len(rows) - 1


lo, hi = first_forward(rowsk[k - 1]) - 9, first_forward(rowsk[k]) + 2
t0 = int(rows[lo]["Start_Timestamp"])
qs = sorted(set(r["Queue_Id"] for r in rows[lo:hi]))
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    q = qs.index(r["Queue_Id"])
    print("%9.1f %s%-2d %8.1f  %s" % (s / 1e3, "          " * q, q, (e - s) / 1e3, r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:60]))
