"""profiles/rNN_pmc_mfma.json from the passes of tools/pmc_mfma.sh: per GEMM kernel and operand fill (random / all-zero), averaged
over its dispatches: duration (kernel trace), effective shader clock = GRBM_GUI_ACTIVE / 8 XCDs / duration (MI355X_MICROARCH.md,
DVFS give-back), MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), MFMA op counts."""
import collections, csv, glob, json, os, sys

root = sys.argv[1]
out = {"method": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 "
                 "SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace, one pass per program (tools/bench_kernel.py gemm = fp32-input MFMA kernels, "
                 "bench_kernel.py gemm16 = f16x3 kernels; S1 layer shapes) and operand fill (XV_DATA_SCALE=1 random normal, 0 all-zero); "
                 "values are means over the dispatches of each kernel; clock_ghz = GRBM_GUI_ACTIVE / 8 / duration (sum over the 8 XCDs, "
                 "reads high on dispatches < 0.3 ms per the guide); mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (256 CUs x 4 SIMDs x "
                 "GRBM_GUI_ACTIVE / 8)", "runs": {}}
for d in sorted(glob.glob(os.path.join(root, "*/"))):
    tag = os.path.basename(d.rstrip("/"))
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not cc:
        out["runs"][tag] = {"error": "no counter_collection.csv"}
        continue
    dur = {}
    if kt:
        for r in csv.DictReader(open(kt[0])):
            dur[r["Dispatch_Id"]] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3      # us
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    ids = collections.defaultdict(set)
    for r in csv.DictReader(open(cc[0])):
        k = r["Kernel_Name"]
        if "gemm" not in k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        ids[k].add(r["Dispatch_Id"])
    ks = {}
    for k, c in agg.items():
        n = len(ids[k])
        us = sum(dur.get(i, 0.0) for i in ids[k]) / n if dur else None
        gui = c.get("GRBM_GUI_ACTIVE", 0.0) / n
        cyc = gui / 8.0
        ks[k.replace("(anonymous namespace)::", "").split("(")[0][:80]] = {
            "dispatches": n, "avg_us": None if us is None else round(us, 1),
            "clock_ghz": None if not us else round(cyc / us / 1e3, 3),
            "mfma_util": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / n / (1024.0 * cyc), 4) if cyc else None,
            "per_dispatch": {name: round(v / n, 1) for name, v in sorted(c.items())}}
    out["runs"][tag] = ks
print(json.dumps(out, indent=1))
