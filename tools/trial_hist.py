#!/usr/bin/env python3
"""Trials-per-step histogram and wall time per step class on the bench workload (development aid)."""
import os
import sys
import time
from collections import defaultdict

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from membrane_solver_amd import _lib as L  # noqa: E402
from membrane_solver_amd import meshgen  # noqa: E402
from membrane_solver_amd.device import DeviceMesh  # noqa: E402

freq = int(sys.argv[1]) if len(sys.argv) > 1 else 320
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 230
P, T = meshgen.icosphere(freq)
P = meshgen.smooth_displace(P, 0.05)
dm = DeviceMesh(P, T, tile_vertices=256)
nv = len(P)
dm.set_surface_tension(np.ones(len(T)))
dm.set_bending_params(np.ones(nv), np.zeros(nv))
dm.set_params(modules=L.MS_MOD_SURFACE | L.MS_MOD_BENDING)
step = 1e-6
cls = defaultdict(list)
seq = []
for i in range(n_steps):
    t = time.perf_counter()
    r = dm.step(stepper=L.MS_STEPPER_CG, step_size=step, reuse_energy0=2)
    dt = time.perf_counter() - t
    if r.success and i >= 30:
        print(f"acc {i} start={step:.6e} alpha={r.alpha:.6e} trials={r.trials}")
    step = r.next_step
    if not r.success:
        dm.reset_stepper()
    if i >= 30:
        cls[(bool(r.success), r.trials)].append(dt * 1e6)
        seq.append(r.trials if r.success else -r.trials)
tot = sum(sum(v) for v in cls.values())
for k, v in sorted(cls.items()):
    print(f"success={k[0]} trials={k[1]}: n={len(v)} avg={np.mean(v):.1f} us  median={np.median(v):.1f}")
print(f"total {tot:.0f} us over {len(seq)} steps -> {len(seq) / tot * 1e6:.0f} steps/s")
print("sequence:", " ".join(str(s) for s in seq[:80]))
