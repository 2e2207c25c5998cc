#!/usr/bin/env python3
"""Where does a relaxation / a step of the config-5 deck spend its time inside the one-workgroup interpreter?
Per-record durations (s_memrealtime around every record, ms_exec_trace).  usage: python3 tools/exec_trace.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from conftest import load_golden
from test_gpu_leaflet import _leaflet_minimizer

g = load_golden("traj_config5_deck_gd.npz")
mesh, mz, _ = _leaflet_minimizer(g, "gd", observe=False)
mesh.disk_rows_in = mesh.disk_rows_out = g["disk_rows"]
mir, dm = mz._device()
mz._relax_tilts(dm)
dm.exec_trace(True)
t0 = time.perf_counter()
mz._relax_tilts(dm)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
rows = dm.exec_trace(True)
tot = sum(r["total_us"] for r in rows)
print(f"one relaxation: wall {1e3 * dt:.2f} ms, records {sum(r['count'] for r in rows)}, record time {tot / 1e3:.2f} ms")
for r in sorted(rows, key=lambda r: -r["total_us"]):
    print(f"  {r['kind']:12s} mode {r['mode']} inst {r['inst']:3d}: {r['count']:5d} x {r['avg_us']:7.2f} us = {r['total_us'] / 1e3:7.3f} ms")
mz.minimize(2, sync_mesh=False)
dm.exec_trace(True)
t0 = time.perf_counter()
mz.minimize(4, sync_mesh=False)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
rows = dm.exec_trace(False)
tot = sum(r["total_us"] for r in rows)
print(f"four steps: wall {1e3 * dt:.2f} ms, records {sum(r['count'] for r in rows)}, record time {tot / 1e3:.2f} ms; stats {dm.exec_stats()}")
for r in sorted(rows, key=lambda r: -r["total_us"]):
    print(f"  {r['kind']:12s} mode {r['mode']} inst {r['inst']:3d}: {r['count']:5d} x {r['avg_us']:7.2f} us = {r['total_us'] / 1e3:7.3f} ms")
