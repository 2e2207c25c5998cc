#!/usr/bin/env python3
"""One-off: HIP vs CPU oracle at the full BASELINE size (f=320), plus run-to-run
reproducibility of the HIP gradient.  Development aid (takes ~10 s of CPU)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from membrane_solver_amd import _lib as L, meshgen  # noqa: E402
from membrane_solver_amd.device import DeviceMesh  # noqa: E402
from oracle import ms_oracle as orc  # noqa: E402

f = int(sys.argv[1]) if len(sys.argv) > 1 else 320
P, T = meshgen.icosphere(f)
P = meshgen.smooth_displace(P, 0.05)
nv, nf = len(P), len(T)
dm = DeviceMesh(P, T)
dm.set_surface_tension(np.ones(nf))
dm.set_bending_params(np.ones(nv), np.zeros(nv))
dm.set_params(modules=L.MS_MOD_SURFACE | L.MS_MOD_BENDING)
e1, g1 = dm.energy_and_gradient()
e2, g2 = dm.energy_and_gradient()
rel = lambda a, b: float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
print("run-to-run: dE", abs(e1.sum() - e2.sum()) / abs(e1.sum()), "dgrad", rel(g1, g2))
t = time.time()
gref = np.zeros_like(P)
Es = orc.surface_energy_and_gradient(P, T, np.ones(nf), gref)
Eb = orc.bending_energy_and_gradient(P, T, np.ones(nv), np.zeros(nv), np.zeros(nv, bool), grad=gref)
print(f"oracle {time.time()-t:.2f}s  Es rel {abs(e1[0]-Es)/Es:.3e}  Eb rel {abs(e1[1]-Eb)/Eb:.3e}  grad rel(max-norm) {rel(g1, gref):.3e}")
gs = np.zeros_like(P)
orc.surface_energy_and_gradient(P, T, np.ones(nf), gs)
dm.set_params(modules=L.MS_MOD_SURFACE)
_, g3 = dm.energy_and_gradient()
print("surface-only grad rel", rel(g3, gs))
