#!/usr/bin/env python3
"""Queue diagnostics (stamps build): run a few steps; if the line-search queue reports a missed gated launch, print
what every workgroup of the last gated gradient pass saw at its gate."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from membrane_solver_amd import _lib as L, meshgen
from membrane_solver_amd.device import DeviceMesh
P, T = meshgen.icosphere(320); P = meshgen.smooth_displace(P, 0.05)
nv, nf = len(P), len(T)
dm = DeviceMesh(P, T)
dm.set_surface_tension(np.ones(nf)); dm.set_bending_params(np.ones(nv), np.zeros(nv))
dm.set_params(modules=L.MS_MOD_SURFACE | L.MS_MOD_BENDING)
step = 1e-6
try:
    for i in range(12):
        r = dm.step(stepper=L.MS_STEPPER_CG, step_size=step)
        step = r.next_step
        if not r.success:
            dm.reset_stepper()
    print("no failure in 12 steps")
except L.MembraneHipError as e:
    print("FAILED:", str(e)[:200])
    n = dm.tile_stats()["n_tiles"]
    buf = np.zeros((n, 8), dtype=np.uint64)
    L.lib().ms_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(n))
    E = buf[:, 6].view(np.float64)
    run = buf[:, 7]
    vals, cnt = np.unique(E, return_counts=True)
    print("workgroups that ran:", int(run.sum()), "of", n, "; distinct E_last seen:", list(zip(vals.tolist(), cnt.tolist()))[:6])
    print("blocks that did NOT run (first 20):", np.flatnonzero(run == 0)[:20])
