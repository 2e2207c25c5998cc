"""Where does a step's wall time go: Minimizer (Python bookkeeping) vs a bare ms_step loop."""
import ctypes
import time

import numpy as np

from membrane_solver_amd import _lib as L
from membrane_solver_amd import meshgen
from membrane_solver_amd.device import DeviceMesh

P, T = meshgen.icosphere(320)
P = meshgen.smooth_displace(P, 0.05)
nv, nf = P.shape[0], T.shape[0]
dm = DeviceMesh(P, T)
dm.set_surface_tension(np.ones(nf))
dm.set_bending_params(np.ones(nv), np.zeros(nv))
dm.set_params(modules=L.MS_MOD_SURFACE | L.MS_MOD_BENDING)
lib = L.lib()
sp = L.ms_stepper_params(L.MS_STEPPER_CG, 10, 0.7, 1e-4, 1.5, 10.0, 10, 0.0, 2)
r = L.ms_step_result()
step = 1e-6


def run(n):
    global step
    for _ in range(n):
        lib.ms_step(dm._h, ctypes.byref(sp), step, 1e-6, ctypes.byref(r))
        step = r.next_step
        if not r.success:
            lib.ms_reset_stepper(dm._h)


run(40)
t0 = time.perf_counter()
run(200)
dt = time.perf_counter() - t0
print("bare ms_step loop: %.1f us/step" % (1e6 * dt / 200))
dm.profile_enable(True)
dm.profile_read()
run(100)
prof = dm.profile_read()
tot = sum(ms for ms, n in prof.values())
print("kernel time per step (HIP events): %.1f us" % (1e3 * tot / 100), {k: (round(1e3 * ms / max(n, 1), 1), n) for k, (ms, n) in prof.items()})

# the same loop through the reference-shaped Python Minimizer
from membrane_solver_amd.geometry.mesh import ArrayMesh  # noqa: E402
from membrane_solver_amd.runtime.constraint_manager import ConstraintModuleManager  # noqa: E402
from membrane_solver_amd.runtime.energy_manager import EnergyModuleManager  # noqa: E402
from membrane_solver_amd.runtime.minimizer import Minimizer  # noqa: E402
from membrane_solver_amd.runtime.steppers import ConjugateGradient  # noqa: E402

dm.close()
gp = {"surface_tension": 1.0, "bending_modulus": 1.0, "spontaneous_curvature": 0.0,
      "bending_gradient_mode": "analytic", "volume_constraint_mode": "lagrange",
      "volume_projection_during_minimization": False}
mods = ["surface", "bending"]
mesh = ArrayMesh(P, T, global_parameters=gp, energy_modules=mods, constraint_modules=[])
mz = Minimizer(mesh, mesh.global_parameters, ConjugateGradient(), EnergyModuleManager(mods),
               ConstraintModuleManager([]), quiet=True, step_size=1e-6)
mz.minimize(40, sync_mesh=False)
t0 = time.perf_counter()
mz.minimize(200, sync_mesh=False)
dt = time.perf_counter() - t0
print("Minimizer.minimize: %.1f us/step" % (1e6 * dt / 200))
for n in (1, 2, 200, 1000):
    t0 = time.perf_counter()
    mz.minimize(n, sync_mesh=False)
    dt = time.perf_counter() - t0
    print("Minimizer.minimize(%d): %.1f us total, %.1f us/step" % (n, 1e6 * dt, 1e6 * dt / n))
import cProfile  # noqa: E402
import pstats  # noqa: E402

pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    mz.minimize(2, sync_mesh=False)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
