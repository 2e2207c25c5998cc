"""Measurement of the tilt rows at the headline size: surface + tilt +
bending_tilt on the 2 048 000-facet icosphere with a nested Jacobi-CG tilt relaxation --
the single-field counterpart of BASELINE config 5.  Prints one JSON line (not the bench
contract: bench.py stays the headline metric)."""
import argparse
import json
import time

import numpy as np

from membrane_solver_amd import meshgen
from membrane_solver_amd.geometry.mesh import ArrayMesh
from membrane_solver_amd.runtime.constraint_manager import ConstraintModuleManager
from membrane_solver_amd.runtime.energy_manager import EnergyModuleManager
from membrane_solver_amd.runtime.minimizer import Minimizer
from membrane_solver_amd.runtime.steppers import GradientDescent

ap = argparse.ArgumentParser()
ap.add_argument("--freq", type=int, default=320)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--inner", type=int, default=5)
ap.add_argument("--events-first", action="store_true",
                help="round 3's measurement: the window right after the warm-up WITH an event pair around every launch "
                     "(steps_per_s_with_launch_events is then that window; steps_per_s the one after it)")
ap.add_argument("--leaflet", action="store_true",
                help="two-leaflet family instead: surface + tilt_in/out + bending_tilt_in/out + tilt_smoothness_in/out")
args = ap.parse_args()

P, T = meshgen.icosphere(args.freq)
P = meshgen.smooth_displace(P, 0.05)
nv, nf = P.shape[0], T.shape[0]
rng = np.random.default_rng(1)
tl = 0.05 * rng.normal(size=P.shape)
gp = {"surface_tension": 1.0, "bending_modulus": 1.0, "spontaneous_curvature": 0.0, "tilt_rigidity": 2.0,
      "bending_gradient_mode": "analytic", "volume_constraint_mode": "lagrange",
      "volume_projection_during_minimization": False, "tilt_solve_mode": "nested", "tilt_solver": "cg",
      "tilt_step_size": 0.2, "tilt_inner_steps": args.inner}
mods = ["surface", "tilt", "bending_tilt"]
if args.leaflet:
    gp.update({"tilt_modulus_in": 2.0, "tilt_modulus_out": 1.5, "bending_modulus_in": 1.0, "bending_modulus_out": 0.8})
    mods = ["surface", "tilt_in", "tilt_out", "bending_tilt_in", "bending_tilt_out", "tilt_smoothness_in",
            "tilt_smoothness_out"]
    mesh = ArrayMesh(P, T, tilts_in=tl, tilts_out=0.8 * tl[::-1].copy(), global_parameters=gp, energy_modules=mods,
                     constraint_modules=[])
else:
    mesh = ArrayMesh(P, T, tilts=tl, global_parameters=gp, energy_modules=mods, constraint_modules=[])
mz = Minimizer(mesh, mesh.global_parameters, GradientDescent(), EnergyModuleManager(mods),
               ConstraintModuleManager([]), quiet=True, step_size=1e-6)
E0 = mz.compute_energy()
mz.minimize(5, sync_mesh=False)
dm = mesh._hip_mirror.dm
def plain_window():
    ts0 = dm.tsearch_stats()
    t0 = time.perf_counter()
    r = mz.minimize(args.steps, sync_mesh=False)
    return r, time.perf_counter() - t0, ts0, dm.tsearch_stats()


def event_window():
    # an event pair around every launch (per-kernel averages; slower)
    dm.profile_enable(True)
    dm.profile_read()
    t0 = time.perf_counter()
    r = mz.minimize(args.steps, sync_mesh=False)
    d = time.perf_counter() - t0
    pr = dm.profile_read()
    dm.profile_enable(False)
    return r, d, pr


if args.events_first:
    _, dtp, prof = event_window()
    res, dt, ts0, ts1 = plain_window()
else:
    _, dt, ts0, ts1 = plain_window()
    res, dtp, prof = event_window()
# relaxation alone
t1 = time.perf_counter()
relax = dm.relax_leaflet_tilts if args.leaflet else dm.relax_tilts
it, ev = relax(solver="cg", max_iters=args.inner, step_size=0.2)
dtr = time.perf_counter() - t1
out = {"workload": f"icosphere f={args.freq} (nv={nv}, nf={nf}), {' + '.join(mods)}, GD shape stepper, "
                   f"nested Jacobi-CG tilt relaxation ({args.inner} inner steps)",
       "steps_per_s": args.steps / dt, "ms_per_step": 1e3 * dt / args.steps,
       "steps_per_s_with_launch_events": args.steps / dtp,
       "search_passes_per_step": (ts1["passes"] - ts0["passes"]) / args.steps,
       "step_sizes_per_step": (ts1["step_sizes"] - ts0["step_sizes"]) / args.steps,
       "relax_ms": 1e3 * dtr, "relax_iters": it, "relax_evals": ev,
       "energy_start": E0, "energy_end": res["energy"],
       "kernels_per_step": {k: {"avg_us": 1e3 * ms / n, "launches_per_step": n / args.steps}
                            for k, (ms, n) in prof.items() if n}}
print(json.dumps(out))
