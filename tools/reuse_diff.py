"""Diagnostic: which ms_step outputs differ between evaluation-reuse levels."""
import numpy as np
from membrane_solver_amd import _lib as L
from membrane_solver_amd import meshgen
from membrane_solver_amd.device import DeviceMesh

pos, tri = meshgen.icosphere(24)
pos = meshgen.smooth_displace(pos, 0.05)
pos = pos + 2.0e-3 * np.random.default_rng(3).standard_normal(pos.shape)
nv = pos.shape[0]
outs = []
names = ["success", "trials", "energy", "energy_eval", "grad_norm", "g_dot_d", "alpha"]
for level in (0, 1, 2):
    dm = DeviceMesh(pos, tri)
    dm.set_surface_tension(np.full(tri.shape[0], 1.0))
    dm.set_bending_params(np.full(nv, 1.0), np.full(nv, 0.2))
    dm.set_params(modules=L.MS_MOD_SURFACE | L.MS_MOD_BENDING)
    step, rows = 5.0e-2, []
    for _ in range(25):
        r = dm.step(stepper=L.MS_STEPPER_CG, step_size=step, reuse_energy0=level)
        rows.append((r.success, r.trials, r.energy, r.energy_eval, r.grad_norm, r.g_dot_d, r.alpha))
        step = r.next_step
        if not r.success:
            dm.reset_stepper()
    outs.append(np.array(rows, dtype=np.float64))
    dm.close()
for lv in (1, 2):
    d = outs[lv] != outs[0]
    print("level", lv, "differs:", d.any())
    for i, j in zip(*np.nonzero(d)):
        print("  step", i, names[j], repr(outs[0][i, j]), repr(outs[lv][i, j]))
        if i > 3:
            break
print(outs[0][:8])
