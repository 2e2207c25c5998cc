#!/usr/bin/env python3
"""Quick wall-clock probe of the device-resident step (development aid)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from membrane_solver_amd import _lib as L  # noqa: E402
from membrane_solver_amd import meshgen  # noqa: E402
from membrane_solver_amd.device import DeviceMesh  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--freq", type=int, default=320)
ap.add_argument("--tile", type=int, default=256)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--modules", default="surface,bending")
ap.add_argument("--stepper", default="cg")
ap.add_argument("--volume", action="store_true")
args = ap.parse_args()

t = time.time()
P, T = meshgen.icosphere(args.freq)
P = meshgen.smooth_displace(P, 0.05)
print(f"mesh f={args.freq} nv={len(P)} nf={len(T)} gen {time.time()-t:.2f}s", flush=True)
t = time.time()
dm = DeviceMesh(P, T, tile_vertices=args.tile)
print(f"ms_create {time.time()-t:.2f}s  tiles={dm.tile_stats()}", flush=True)
mods = 0
if "surface" in args.modules:
    mods |= L.MS_MOD_SURFACE
if "bending" in args.modules:
    mods |= L.MS_MOD_BENDING
if args.volume:
    mods |= L.MS_CON_VOLUME
nv = len(P)
dm.set_surface_tension(np.ones(len(T)))
dm.set_bending_params(np.ones(nv), np.zeros(nv))
dm.set_params(modules=mods)
e, _ = dm.energy_and_gradient(want_grad=False)
print("energies", e, flush=True)
for name, fn in (("energy_and_gradient", lambda: dm.energy_and_gradient(want_grad=False)),
                 ("energy", dm.energy)):
    fn()
    t = time.time()
    n = 20
    for _ in range(n):
        fn()
    print(f"{name}: {(time.time()-t)/n*1e6:.1f} us/call", flush=True)
stp = L.MS_STEPPER_CG if args.stepper == "cg" else L.MS_STEPPER_GD
step = 1e-3
ap_warm = int(os.environ.get("QB_WARM", "3"))
for i in range(ap_warm):
    r = dm.step(stepper=stp, step_size=step)
    if i < 60 or i % 10 == 0:
        print(f"  warm {i}: ok={r.success} trials={r.trials} guard={r.guard_rejects} alpha={r.alpha:.3e} next={r.next_step:.3e} E={r.energy:.12f} |g|={r.grad_norm:.3e} gd={r.g_dot_d:.3e}", flush=True)
    step = r.next_step
    if not r.success:  # what minimize() does after a failed search (minimizer.py:1414-1426)
        dm.reset_stepper()
t = time.time()
acc = 0
trials = 0
for _ in range(args.steps):
    r = dm.step(stepper=stp, step_size=step)
    step = r.next_step
    if not r.success:
        dm.reset_stepper()
    acc += r.success
    trials += r.trials
dt = (time.time() - t) / args.steps
print(f"step: {dt*1e6:.1f} us/step  -> {1/dt:.1f} steps/s  accepted {acc}/{args.steps} trials {trials} E={r.energy:.12f}", flush=True)
