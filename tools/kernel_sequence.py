"""Dispatch-by-dispatch listing of a rocprofv3 --kernel-trace CSV (last N ms:: dispatches): start offset, duration,
gap to the previous dispatch's end.

usage: python3 tools/kernel_sequence.py <dir with *_kernel_trace.csv> [last_n]
"""
import csv
import glob
import sys

d = sys.argv[1]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 300
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f)) if "ms::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-last:]
t0 = int(rows[0]["Start_Timestamp"])
prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"]
    n = n[n.find("ms::") + 4:].split("(")[0]
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:7.2f}  gap {gap:8.2f}  grid {r.get('Grid_Size', '?'):>8}  {n[:70]}")
    prev = e
