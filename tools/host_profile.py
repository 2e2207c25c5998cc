"""cProfile of Minimizer.minimize on the bench workload (host-side overhead per step)."""
import cProfile
import pstats

import numpy as np

from membrane_solver_amd import meshgen
from membrane_solver_amd.geometry.mesh import ArrayMesh
from membrane_solver_amd.runtime.constraint_manager import ConstraintModuleManager
from membrane_solver_amd.runtime.energy_manager import EnergyModuleManager
from membrane_solver_amd.runtime.minimizer import Minimizer
from membrane_solver_amd.runtime.steppers import ConjugateGradient

P, T = meshgen.icosphere(320)
P = meshgen.smooth_displace(P, 0.05)
gp = {"surface_tension": 1.0, "bending_modulus": 1.0, "bending_energy_model": "helfrich",
      "spontaneous_curvature": 0.0, "bending_gradient_mode": "analytic", "volume_constraint_mode": "lagrange",
      "volume_projection_during_minimization": False}
mods = ["surface", "bending"]
mesh = ArrayMesh(P, T, global_parameters=gp, energy_modules=mods, constraint_modules=[])
mz = Minimizer(mesh, mesh.global_parameters, ConjugateGradient(), EnergyModuleManager(mods),
               ConstraintModuleManager([]), quiet=True, step_size=1e-6)
mz.minimize(30, sync_mesh=False)
pr = cProfile.Profile()
pr.enable()
mz.minimize(300, sync_mesh=False)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
