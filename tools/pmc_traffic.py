"""profiles/rNN_pmc_traffic.json from two rocprofv3 counter_collection.csv files (separate --pmc FETCH_SIZE / WRITE_SIZE passes)."""
import csv, json, sys, collections
fetch_csv, write_csv, out, cmd = sys.argv[1:5]
def per_kernel(path, counter):
    tot, n = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        tot[r["Kernel_Name"]] += float(r["Counter_Value"]); n[r["Kernel_Name"]].add(r["Dispatch_Id"])
    return {k: (tot[k] / len(n[k]), len(n[k])) for k in tot}
f, w = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
kernels = {}
for k in sorted(f, key=lambda k: -f[k][0] * f[k][1]):
    if "gemm" not in k:
        continue
    fk, wk = f[k][0], w.get(k, (0.0, 0))[0]
    kernels[k] = {"dispatches": f[k][1], "fetch_kib_raw": round(fk, 1), "write_kib": round(wk, 1),
                  "bytes_per_launch_corrected": int((2.0 * fk + wk) * 1024)}


def nt_kernel_of(M, N, K, stats, co_running):
    """The kernel xv_launch_gemm_nt gives this problem - asked of the library itself (xv_debug_nt_schedule), so the map follows the launcher."""
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tf_kaldi_speaker_amd import _lib
    kind = _lib.load().xv_debug_nt_schedule(int(M), int(N), int(K), int(stats), int(co_running))
    return ("xv_gemm_nt_sk_kernel<%s," if kind == 1 else "xv_gemm_nt_kernel<%s,") % ("true" if stats else "false")


def s1_algorithmic_mb(names):
    """Algorithmic MB per launch (operands read once + the result written once, fp32) of every GEMM launch of one S1 step (128 x 200 x 30,
    tdnn.py:35-127, 7351 speakers), grouped by the kernel the launcher runs it on (nt_kernel_of)."""
    B, T, spk = 128, 200, 7351
    layers = [(5, 32, 512), (5, 512, 512), (7, 512, 512), (1, 512, 512), (1, 512, 1500)]      # (taps, padded input channels, outputs)
    mb = lambda *els: sum(els) * 4 / 1e6
    groups = collections.defaultdict(list)
    t_in = T
    for i, (k, c, o) in enumerate(layers):
        t_out = t_in - k + 1
        x, w, y = B * t_in * c, k * c * o, B * t_out * o
        groups[nt_kernel_of(B * t_out, o, k * c, True, False)].append(mb(x, w, y))                          # forward
        if i > 0:
            groups[nt_kernel_of(B * (t_out + k - 1), c, k * o, False, True)].append(mb(y, w, x))            # data gradient
        groups["xv_gemm_tn_kernel"].append(mb(x, y, w))                                   # weight gradient
        t_in = t_out
    for m, n in ((3000, 512), (512, 512), (512, spk + 1)):            # tdnn6, tdnn7, the loss head: weight gradients of [B][m]^T . [B][n]
        groups["xv_gemm_tn_kernel"].append(mb(B * m, B * n, m * n))
    return {g: v for g, v in groups.items() if v}


alg = s1_algorithmic_mb(list(kernels))
for name, v in kernels.items():
    for g, mbs in alg.items():
        if g in name:
            v["s1_algorithmic_mb_per_launch"] = round(sum(mbs) / len(mbs), 1)
            v["s1_launches_per_step"] = len(mbs)
            v["ratio_to_algorithmic"] = round(v["bytes_per_launch_corrected"] / 1e6 / (sum(mbs) / len(mbs)), 2)


def direction_ratios():
    """Per direction (all kernels that serve it together, dispatch-weighted): bytes past L2 per step / algorithmic bytes per step at S1 -
    independent of which kernel the launcher's schedule picked for which layer."""
    B, T, spk = 128, 200, 7351
    layers = [(5, 32, 512), (5, 512, 512), (7, 512, 512), (1, 512, 512), (1, 512, 1500)]
    mb = lambda *els: sum(els) * 4 / 1e6
    alg = {"forward": 0.0, "data_gradient": 0.0, "weight_gradient": 0.0}
    n = {"forward": 0, "data_gradient": 0, "weight_gradient": 0}
    t_in = T
    for i, (k, c, o) in enumerate(layers):
        t_out = t_in - k + 1
        x, w, y = B * t_in * c, k * c * o, B * t_out * o
        alg["forward"] += mb(x, w, y); n["forward"] += 1
        if i > 0:
            alg["data_gradient"] += mb(y, w, x); n["data_gradient"] += 1
        alg["weight_gradient"] += mb(x, y, w); n["weight_gradient"] += 1
        t_in = t_out
    for m, nn in ((3000, 512), (512, 512), (512, spk + 1)):
        alg["weight_gradient"] += mb(B * m, B * nn, m * nn); n["weight_gradient"] += 1
    pick = {"forward": lambda k: "gemm_nt" in k and "16" not in k and "<true," in k,
            "data_gradient": lambda k: "gemm_nt" in k and "16" not in k and "<false," in k,
            "weight_gradient": lambda k: "xv_gemm_tn_kernel" in k}
    out = {}
    for d, f in pick.items():
        ks = [k for k in kernels if f(k)]
        disp = sum(kernels[k]["dispatches"] for k in ks)
        if not disp:
            continue
        steps = disp / float(n[d])
        per_step = sum(kernels[k]["bytes_per_launch_corrected"] * kernels[k]["dispatches"] for k in ks) / steps / 1e6
        out[d] = {"mb_per_step": round(per_step, 1), "s1_algorithmic_mb_per_step": round(alg[d], 1), "ratio_to_algorithmic": round(per_step / alg[d], 2)}
    return out


json.dump({"s1_direction_totals_fp32": direction_ratios(), "algorithmic_note": "s1_algorithmic_mb_per_launch = operands once + result once, averaged over the launches of one S1 step that the "
                               "launcher gives to this kernel (valid for the default S1 bench command only)",
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes over `%s`; values are KiB per dispatch averaged "
                     "over all dispatches of the kernel; per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 reports 1/2 of a wide "
                     "coalesced read stream so it is doubled; WRITE_SIZE is taken as is; Infinity-Cache hits are counted as fetches, so this "
                     "is traffic past the XCD L2s, an upper bound on HBM bytes." % cmd, "kernels": kernels}, open(out, "w"), indent=1)
print(json.dumps(kernels, indent=1))
