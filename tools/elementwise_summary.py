"""Per-kernel HBM rates from the rocprofv3 kernel trace of tools/bench_kernel.py elementwise: algorithmic bytes (what the kernel must read and
write once, fp32 unless planes) / average duration, as a fraction of the 8 TB/s HBM3E peak.  Kernels are told apart by name and - where
one kernel serves two tensor widths - by grid size."""
import collections, csv, json, sys

rows = list(csv.DictReader(open(sys.argv[1])))
T = (int(sys.argv[2]) if len(sys.argv) > 2 else 200) - 14
R = 128 * T
agg = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    key = (name, r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""), r.get("Grid_Size_Y", ""))
    agg[key].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)


def nbytes(name, n):
    t = R * n * 4.0
    if name.startswith("col_stats"): return t
    if name.startswith("bn_apply_split"): return t + t            # fp32 in, two fp16 planes out
    if name.startswith("bn_apply_kernel"): return 2 * t
    if "stat_pool_fwd" in name: return t
    if "bn_bwd_reduce_pooled_kernel" in name: return t
    if name.startswith("bn_bwd_reduce_kernel") or "bn_bwd_reduce_kernel<false>" in name: return 2 * t
    if "bn_bwd_apply_dense_kernel<true>" in name: return 2 * t      # pooled upstream gradient: z in, dz out
    if "bn_bwd_apply_dense_kernel<false>" in name: return 3 * t     # da, z in, dz out (no padding rows)
    if "bn_bwd_apply_kernel<true>" in name or "bn_bwd_apply_split_kernel<true>" in name: return 2 * t
    if "bn_bwd_apply_kernel<false>" in name or "bn_bwd_apply_split_kernel<false>" in name: return 3 * t * (1 + 8.0 / T / 3)
    if name.startswith("sgd_kernel"): return 9.83e6 * 12
    if "vectorized_elementwise_kernel" in name: return 2 * t      # the yardstick: torch.add(z, 1, out=tmp)
    return None


out = []
for (name, gx, gy), ds in agg.items():
    ds = ds[2:] if len(ds) > 4 else ds
    us = sum(ds) / len(ds)
    for n in (512, 1500):
        b = nbytes(name, n)
        if b is None:
            continue
        # width by plausibility: the rate may not exceed 8 TB/s... pick the width whose grid matches
        out.append((name, gx, gy, n, us, b))
seen = set()
res = []
for name, gx, gy, n, us, b in sorted(out, key=lambda v: v[0]):
    # two candidate widths per (name, grid): keep 1500 for the larger grid of a name, 512 for the smaller
    grids = sorted({(int(a[1] or 0) * max(int(a[2] or 1), 1)) for a in out if a[0] == name})
    g = int(gx or 0) * max(int(gy or 1), 1)
    width = 1500 if (len(grids) > 1 and g == grids[-1]) or ("kernel<true>" in name or "stat_pool" in name or "reduce_pooled" in name) else 512
    if name.startswith("sgd_kernel"):
        width = 0
    if n != (width or 512) or (name, gx, gy) in seen:
        continue
    seen.add((name, gx, gy))
    res.append({"kernel": name[:60], "grid": [gx, gy], "channels": width, "avg_us": round(us, 1), "algorithmic_mb": round(b / 1e6, 1),
                "tb_per_s": round(b / us / 1e6, 2), "frac_of_8tbs": round(b / us / 1e6 / 8.0, 3)})
print(json.dumps({"unit": "S1 tensors: 128 chunks x %d frames x channels, fp32" % T, "kernels": res}, indent=1))
