#!/usr/bin/env python3
"""BASELINE config 5 on its own deck (the fixture tests/golden/traj_config5_deck_gd.npz: 109 vertices, 204 facets, the
deck's modules and parameters without its three constraint modules): one relax_leaflet_tilts call as the deck configures
it, and the deck's `g` steps.  204 facets are launch-latency scale -- this measures launches and host round trips, not
kernels.  Prints one JSON line.
usage: python3 tools/bench_config5.py [STEPS=40]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from conftest import load_golden
from test_gpu_leaflet import _leaflet_minimizer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
g = load_golden("traj_config5_deck_gd.npz")
mesh, mz, _ = _leaflet_minimizer(g, "gd", observe=False)
mesh.disk_rows_in = mesh.disk_rows_out = g["disk_rows"]
mir, dm = mz._device()
mz._relax_tilts(dm)  # warm
torch.cuda.synchronize()
t0 = time.perf_counter()
n_rel = 10
for _ in range(n_rel):
    mz._relax_tilts(dm)
torch.cuda.synchronize()
t_rel = (time.perf_counter() - t0) / n_rel
mz.minimize(4, sync_mesh=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
res = mz.minimize(steps, sync_mesh=False)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
gp = mesh.global_parameters
print(json.dumps({"workload": "config 5 deck (109 vertices, 204 facets): " + " + ".join(str(m) for m in g["modules"]),
                  "tilt_solve_mode": str(gp.get("tilt_solve_mode")), "tilt_inner_steps": int(gp.get("tilt_inner_steps", 0) or 0),
                  "relaxation_ms": 1e3 * t_rel, "steps": steps, "ms_per_step": 1e3 * dt / steps,
                  "steps_per_s": steps / dt, "energy_end": float(res["energy"])}))
