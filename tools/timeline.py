"""Timeline of a rocprofv3 --kernel-trace CSV: busy time, gaps and per-kernel durations over the last N dispatches.

usage: python3 tools/timeline.py <dir with *_kernel_trace.csv> [last_n]
"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if r["Kernel_Name"].startswith("ms::") or "ms::" in r["Kernel_Name"]][-last:]
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
busy = 0
gaps = defaultdict(list)
dur = defaultdict(list)
prev_end, prev_name = None, None


def short(n):
    n = n.split("(")[0]
    for k in ("k_energy", "k_gradient", "k_reduce", "k_direction"):
        if k in n:
            multi = n[n.find("<") + 1:n.rfind(">")].split(",")[-1].strip() if k == "k_energy" and n.count(",") >= 5 else "0"
            return k + (f"<MULTI={multi}>" if multi in ("2", "3") else "")
    return n[:30]


for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = short(r["Kernel_Name"])
    dur[nm].append((e - s) / 1e3)
    busy += e - s
    if prev_end is not None:
        gaps[prev_name + " -> " + nm].append((s - prev_end) / 1e3)
    prev_end, prev_name = e, nm
print(f"span {1e-3 * (t1 - t0):.1f} us, busy {1e-3 * busy:.1f} us ({busy / (t1 - t0):.3f}), dispatches {len(rows)}")
for k, v in sorted(dur.items()):
    big = [x for x in v if x > 12.0] if k != "k_reduce" else [x for x in v if x > 4.0]
    print(f"  {k:22s} n={len(v):5d} avg={sum(v) / len(v):7.2f} us   non-empty n={len(big):5d} avg={sum(big) / max(1, len(big)):7.2f}")
print("gaps:")
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {k:40s} n={len(v):5d} avg={sum(v) / len(v):7.2f} us  total={sum(v):9.1f}")
