#!/usr/bin/env python3
"""Per-call overhead of Minimizer.minimize on the headline problem: n calls of one step against one call of n steps,
and a cProfile of the Python side of the one-step calls."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench

P, T = bench.bench_mesh(int(os.environ.get("FREQ", "320")))
from membrane_solver_amd.geometry.mesh import ArrayMesh
from membrane_solver_amd.runtime.constraint_manager import ConstraintModuleManager
from membrane_solver_amd.runtime.energy_manager import EnergyModuleManager
from membrane_solver_amd.runtime.minimizer import Minimizer
from membrane_solver_amd.runtime.steppers import ConjugateGradient
mesh = ArrayMesh(P, T, global_parameters=dict(bench.GP), energy_modules=["surface", "bending"], constraint_modules=[], bodies=[])
mz = Minimizer(mesh, mesh.global_parameters, ConjugateGradient(), EnergyModuleManager(["surface", "bending"]),
               ConstraintModuleManager([]), quiet=True, step_size=1e-6)
mz.minimize(60, sync_mesh=False)
torch.cuda.synchronize()
n = 100
t0 = time.perf_counter(); mz.minimize(n, sync_mesh=False); torch.cuda.synchronize(); t_one = time.perf_counter() - t0
t0 = time.perf_counter()
for _ in range(n):
    mz.minimize(1, sync_mesh=False)
torch.cuda.synchronize(); t_many = time.perf_counter() - t0
print(f"one call of {n} steps: {1e6*t_one/n:.1f} us/step; {n} calls of one step: {1e6*t_many/n:.1f} us/step; per-call overhead {1e6*(t_many-t_one)/n:.1f} us")
pr = cProfile.Profile(); pr.enable()
for _ in range(200):
    mz.minimize(1, sync_mesh=False)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:3500])
