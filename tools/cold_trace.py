#!/usr/bin/env python3
"""Per-step record of the first N steps of the headline run (bench.py's mesh, modules, stepper and initial step size):
what every search did and how many rounds the queue spent on it.
usage: python3 tools/cold_trace.py [N=40]   (FREQ=320, MS_* switches as for bench.py)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from membrane_solver_amd import _lib as L
from membrane_solver_amd.device import DeviceMesh

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
P, T = bench.bench_mesh(int(os.environ.get("FREQ", "320")))
dm = DeviceMesh(P, T)
dm.set_surface_tension(np.ones(len(T)))
dm.set_bending_params(np.ones(len(P)), np.zeros(len(P)))
dm.set_params(modules=L.MS_MOD_SURFACE | L.MS_MOD_BENDING)
step = 1e-6
names = ["rounds", "multi", "wasted", "side_acc", "mismatch", "ahead", "adopted", "dropped"]
qs = lambda: np.array(list(dm.queue_stats().values()))
prev = qs()
print("step ok trials alpha next_step g.d  us   d(rounds multi wasted side_acc)")
for i in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = dm.step(stepper=L.MS_STEPPER_CG, step_size=step, tol=1e-6, reuse_energy0=2)
    torch.cuda.synchronize(); us = 1e6 * (time.perf_counter() - t0)
    cur = qs(); d = cur - prev; prev = cur
    print(f"{i:3d} {int(r.success)} {r.trials:2d} {r.alpha:.3e} {r.next_step:.3e} {r.g_dot_d:+.2e} {us:6.0f}  {d[0]} {d[1]} {d[2]} {d[3]}")
    step = r.next_step
    if not r.success:
        dm.reset_stepper()
