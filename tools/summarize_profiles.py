#!/usr/bin/env python3
"""Turn gpurun_out/<tag>_{stats,fetch,write,sq} (rocprofv3 csv) into profiles/<tag>_*.csv."""
import collections
import csv
import glob
import os
import shutil
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)
ks = glob.glob(os.path.join(root, "gpurun_out", f"{tag}_stats", "*", "*kernel_stats.csv"))
if ks:
    shutil.copy(ks[0], os.path.join(out, f"{tag}_kernel_stats.csv"))
js = os.path.join(root, "gpurun_out", f"{tag}_stats.json")
if os.path.exists(js):
    shutil.copy(js, os.path.join(out, f"{tag}_bench_under_rocprof.json"))
# The line-search queue launches trial / gradient kernels that test an Armijo gate and return
# at once when it is closed.  Such empty dispatches (a few us) are not samples of the kernel: besides rocprofv3's own
# table, write one computed from the kernel trace without them -- the figure bench.py's HIP events report.
EMPTY_NS = 12000
TILE_KERNELS = ("ms::k_energy", "ms::k_gradient", "ms::k_reduce")
kt = glob.glob(os.path.join(root, "gpurun_out", f"{tag}_stats", "*", "*kernel_trace.csv"))
if kt:
    durs = collections.defaultdict(list)
    skipped = collections.Counter()
    for r in csv.DictReader(open(kt[0])):
        name = r["Kernel_Name"]
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if any(t in name for t in TILE_KERNELS[:2]) and d < EMPTY_NS:
            skipped[name] += 1
            continue
        durs[name].append(d)
    tot = sum(sum(v) for v in durs.values()) or 1
    with open(os.path.join(out, f"{tag}_kernel_stats_nonempty.csv"), "w", newline="") as f:
        f.write(f"# from {os.path.basename(kt[0])}: dispatches of the gated tile kernels shorter than {EMPTY_NS} ns (gate closed,\n")
        f.write("# immediate return) are left out; column EmptyDispatches counts them\n")
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "EmptyDispatches"])
        for name, v in sorted(durs.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([name, len(v), sum(v), "%.3f" % (sum(v) / len(v)), "%.2f" % (100.0 * sum(v) / tot), min(v), max(v),
                        skipped.get(name, 0)])
rows = []
for sub in ("fetch", "write", "sq"):
    for f in glob.glob(os.path.join(root, "gpurun_out", f"{tag}_{sub}", "*", "*counter_collection.csv")):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            empty = (any(t in k for t in TILE_KERNELS[:2])
                     and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) < EMPTY_NS)
            if "ms::" in k and not empty:
                acc[(r["Counter_Name"], k)].append(float(r["Counter_Value"]))
        for (c, k), v in sorted(acc.items()):
            rows.append((c, k, len(v), sum(v) / len(v), min(v), max(v)))
with open(os.path.join(out, f"{tag}_pmc_summary.csv"), "w", newline="") as f:
    f.write("# FETCH_SIZE/WRITE_SIZE in KiB per dispatch; on gfx950 FETCH_SIZE counts 1/2 of the bytes\n")
    f.write("# (calibrated on ms::k_direction / k_gradient's direction epilogue with known bytes) -> double it.\n")
    f.write(f"# dispatches of the gated tile kernels shorter than {EMPTY_NS} ns (gate closed) are left out.\n")
    w = csv.writer(f)
    w.writerow(["counter", "kernel", "dispatches", "avg", "min", "max"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], "%.6g" % r[3], "%.6g" % r[4], "%.6g" % r[5]])
print(open(os.path.join(out, f"{tag}_pmc_summary.csv")).read())
