import time, numpy as np, sys
sys.path.insert(0, '/root/repo')
import bench
from membrane_solver_amd import _lib as L
from membrane_solver_amd.geometry.mesh import ArrayBody, ArrayMesh
from membrane_solver_amd.runtime.constraint_manager import ConstraintModuleManager
from membrane_solver_amd.runtime.energy_manager import EnergyModuleManager
from membrane_solver_amd.runtime.minimizer import Minimizer
from membrane_solver_amd.runtime.steppers import GradientDescent
import torch
P, T = bench.bench_mesh(81)
v0, v1, v2 = P[T[:, 0]], P[T[:, 1]], P[T[:, 2]]
bodies = [ArrayBody(0, None, float(np.einsum("ij,ij->i", np.cross(v1, v2), v0).sum() / 6.0))]
mesh = ArrayMesh(P, T, global_parameters=dict(bench.GP), energy_modules=["surface"], constraint_modules=["volume"], bodies=bodies)
mz = Minimizer(mesh, mesh.global_parameters, GradientDescent(), EnergyModuleManager(["surface"]), ConstraintModuleManager(["volume"]), quiet=True, step_size=1e-3, device=0)
mz.compute_energy(); mz.minimize(30, sync_mesh=False)
for n in (10, 50, 200, 1000, 2000):
    torch.cuda.synchronize(); t0 = time.perf_counter(); mz.minimize(n, sync_mesh=False); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(n, "steps:", round(1e3 * dt, 3), "ms ->", round(1e6 * dt / n, 2), "us/step")
