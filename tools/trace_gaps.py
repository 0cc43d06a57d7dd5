"""Per-queue busy time, idle gaps and per-kernel totals from a rocprofv3 kernel_trace.csv (steady-state steps only)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steady state: the last `steps` sgd_kernel launches delimit steps
sgd = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("sgd_kernel")]
lo, hi = sgd[-steps - 1 - 6], sgd[-1 - 6]      # skip the 6 isolated-pass steps at the end
sel = rows[lo + 1:hi + 1]
t0, t1 = int(sel[0]["Start_Timestamp"]), int(sel[-1]["End_Timestamp"])
print("steps %d  wall %.3f ms/step" % (steps, (t1 - t0) / steps / 1e6))
byq = collections.defaultdict(list)
for r in sel:
    byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
for q, iv in byq.items():
    busy = sum(e - s for s, e, _ in iv)
    print("queue %s: %d kernels/step, busy %.3f ms/step" % (q, len(iv) / steps, busy / steps / 1e6))
# union busy
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sel)
cur_s, cur_e = iv[0]
union = 0
for s, e in iv[1:]:
    if s > cur_e:
        union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
print("GPU busy (union) %.3f ms/step, idle %.3f ms/step" % (union / steps / 1e6, (t1 - t0 - union) / steps / 1e6))
tot = collections.Counter(); cnt = collections.Counter()
for r in sel:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    tot[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cnt[k] += 1
for k, v in tot.most_common(40):
    print("  %-46s %5.1f/step  avg %7.1f us  %7.1f us/step" % (k[:46], cnt[k] / steps, v / cnt[k] / 1e3, v / steps / 1e3))

# largest idle gaps on the busiest queue: (kernel before -> kernel after), averaged per step
mainq = max(byq, key=lambda q: sum(e - s for s, e, _ in byq[q]))
iv = sorted(byq[mainq])
gaps = collections.defaultdict(lambda: [0, 0])
for (s0, e0, k0), (s1, e1, k1) in zip(iv, iv[1:]):
    g = s1 - e0
    if g > 0:
        cl = lambda k: k.replace("(anonymous namespace)::", "").split("(")[0][:34]
        key = (cl(k0), cl(k1))
        gaps[key][0] += g; gaps[key][1] += 1
print("idle gaps on queue %s (us/step):" % mainq)
for key, (g, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:22]:
    print("  %7.1f  (%4.1f/step, avg %6.1f us)  %s -> %s" % (g / steps / 1e3, n / steps, g / n / 1e3, key[0], key[1]))
