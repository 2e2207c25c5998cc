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
rows = []
for sub in ("fetch", "write", "sq"):
    for f in glob.glob(os.path.join(root, "gpurun_out", f"{tag}_{sub}", "*", "*counter_collection.csv")):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "ms::" in k:
                acc[(r["Counter_Name"], k)].append(float(r["Counter_Value"]))
        for (c, k), v in sorted(acc.items()):
            rows.append((c, k, len(v), sum(v) / len(v), min(v), max(v)))
with open(os.path.join(out, f"{tag}_pmc_summary.csv"), "w", newline="") as f:
    f.write("# FETCH_SIZE/WRITE_SIZE in KiB per dispatch; on gfx950 FETCH_SIZE counts 1/2 of the bytes\n")
    f.write("# (calibrated on ms::k_direction / k_gradient's direction epilogue with known bytes) -> double it.\n")
    w = csv.writer(f)
    w.writerow(["counter", "kernel", "dispatches", "avg", "min", "max"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], "%.6g" % r[3], "%.6g" % r[4], "%.6g" % r[5]])
print(open(os.path.join(out, f"{tag}_pmc_summary.csv")).read())
