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
# the other BASELINE configurations (tools/r04_profiles.sh): gpurun_out/<tag>_<name>_stats -> profiles/<tag>_<name>_*
for d in sorted(glob.glob(os.path.join(root, "gpurun_out", f"{tag}_*_stats"))):
    name = os.path.basename(d)[len(tag) + 1:-len("_stats")]
    for f in glob.glob(os.path.join(d, "*", "*kernel_stats.csv")):
        shutil.copy(f, os.path.join(out, f"{tag}_{name}_kernel_stats.csv"))
    j = os.path.join(root, "gpurun_out", f"{tag}_{name}.json")
    if os.path.exists(j) and os.path.getsize(j) > 0:
        shutil.copy(j, os.path.join(out, f"{tag}_{name}.json"))
for name in ("bench", "bench_s20w5", "bench_forced_sharded", "tilt_single_field_2M_events_first",
             "tilt_two_leaflets_2M_events_first"):
    j = os.path.join(root, "gpurun_out", f"{tag}_{name}.json")
    if os.path.exists(j) and os.path.getsize(j) > 0:
        shutil.copy(j, os.path.join(out, f"{tag}_{name}.json"))
js = os.path.join(root, "gpurun_out", f"{tag}_stats.json")
if os.path.exists(js):
    shutil.copy(js, os.path.join(out, f"{tag}_bench_under_rocprof.json"))
# The line-search queue launches trial / gradient kernels that test an Armijo gate and return
# at once when it is closed.  Such empty dispatches (a few us) are not samples of the kernel: besides rocprofv3's own
# table, write one computed from the kernel trace without them -- the figure bench.py's HIP events report.
# A kernel trace does not say what a gated launch read in its decision word (bench.py's own HIP-event figures do: the
# library queues a one-lane probe behind every gated launch while profiling).  Here a dispatch of a gated tile kernel
# counts as empty when it is BOTH shorter than 5 us and shorter than 0.4 x the longest dispatch of the same
# instantiation: an empty dispatch is 2-3.5 us whatever the mesh, a real one is never a small fraction of its siblings
# (rounds 1-3 used a flat 12 us, which threw away every real launch of the 131 k-facet configuration).
EMPTY_NS = 5000
EMPTY_FRAC = 0.4
TILE_KERNELS = ("ms::k_energy", "ms::k_gradient", "ms::k_reduce")


def empty_limits(rows):
    longest = collections.defaultdict(int)
    for name, d in rows:
        if any(t in name for t in TILE_KERNELS[:2]):
            longest[name] = max(longest[name], d)
    return {name: min(EMPTY_NS, EMPTY_FRAC * m) for name, m in longest.items()}


def write_nonempty(trace_csv, dst):
    durs = collections.defaultdict(list)
    skipped = collections.Counter()
    trace = [(r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(trace_csv))]
    lim = empty_limits(trace)
    for name, d in trace:
        if d < lim.get(name, 0):
            skipped[name] += 1
            continue
        durs[name].append(d)
    tot = sum(sum(v) for v in durs.values()) or 1
    with open(dst, "w", newline="") as f:
        f.write(f"# from {os.path.basename(trace_csv)}: dispatches of the gated tile kernels shorter than {EMPTY_NS} ns AND than\n")
        f.write(f"# {EMPTY_FRAC} x the instantiation's longest dispatch (gate closed, immediate return) are left out; column EmptyDispatches counts them\n")
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "EmptyDispatches"])
        for name, v in sorted(durs.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([name, len(v), sum(v), "%.3f" % (sum(v) / len(v)), "%.2f" % (100.0 * sum(v) / tot), min(v), max(v),
                        skipped.get(name, 0)])


kt = glob.glob(os.path.join(root, "gpurun_out", f"{tag}_stats", "*", "*kernel_trace.csv"))
if kt:
    write_nonempty(kt[0], os.path.join(out, f"{tag}_kernel_stats_nonempty.csv"))
for d in sorted(glob.glob(os.path.join(root, "gpurun_out", f"{tag}_*_stats"))):
    name = os.path.basename(d)[len(tag) + 1:-len("_stats")]
    for f in glob.glob(os.path.join(d, "*", "*kernel_trace.csv")):
        write_nonempty(f, os.path.join(out, f"{tag}_{name}_kernel_stats_nonempty.csv"))
rows = []
for sub in ("fetch", "write", "sq"):
    for f in glob.glob(os.path.join(root, "gpurun_out", f"{tag}_{sub}", "*", "*counter_collection.csv")):
        acc = collections.defaultdict(list)
        recs = list(csv.DictReader(open(f)))
        lim = empty_limits([(r["Kernel_Name"].split("(")[0], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in recs])
        for r in recs:
            k = r["Kernel_Name"].split("(")[0]
            empty = int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) < lim.get(k, 0)
            if "ms::" in k and not empty:
                acc[(r["Counter_Name"], k)].append(float(r["Counter_Value"]))
        for (c, k), v in sorted(acc.items()):
            rows.append((c, k, len(v), sum(v) / len(v), min(v), max(v)))
with open(os.path.join(out, f"{tag}_pmc_summary.csv"), "w", newline="") as f:
    f.write("# FETCH_SIZE/WRITE_SIZE in KiB per dispatch; on gfx950 FETCH_SIZE counts 1/2 of the bytes\n")
    f.write("# (calibrated on ms::k_direction / k_gradient's direction epilogue with known bytes) -> double it.\n")
    f.write(f"# dispatches of the gated tile kernels shorter than {EMPTY_NS} ns and than {EMPTY_FRAC} x the longest (gate closed) are left out.\n")
    w = csv.writer(f)
    w.writerow(["counter", "kernel", "dispatches", "avg", "min", "max"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], "%.6g" % r[3], "%.6g" % r[4], "%.6g" % r[5]])
print(open(os.path.join(out, f"{tag}_pmc_summary.csv")).read())
