"""Per-kernel averages of a rocprofv3 --pmc counter_collection.csv, empty (gated-off) dispatches left out.

usage: python3 tools/pmc_quick.py <dir>
"""
import collections
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    if "ms::" not in k:
        continue
    acc[k][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for k, cs in sorted(acc.items()):
    # a dispatch is "empty" when its VALU count is far below the kernel's maximum
    ref = cs.get("SQ_INSTS_VALU") or next(iter(cs.values()))
    top = max(v for _, v in ref)
    keep = {d for d, v in ref if v > 0.5 * top}
    print(k[:80], "dispatches", len(keep))
    for c, vals in sorted(cs.items()):
        v = [x for d, x in vals if d in keep]
        if v:
            print(f"    {c:26s} avg {sum(v) / len(v):14.1f}")
