#!/usr/bin/env python3
"""Gate read-path probe (build_variants/lib_gateprobe.so, -DMS_GATE_PROBE=1): run the headline problem for a while and
print how often a workgroup of a gated gradient pass read a value other than the agent-scope one through the scalar
data cache / through the vector L1-L2 path, per XCC.  One run, counters only (tools/gate_probe.sh)."""
import ctypes, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from membrane_solver_amd import _lib as L, meshgen
from membrane_solver_amd.device import DeviceMesh

P, T = meshgen.icosphere(int(os.environ.get("PROBE_FREQ", "320")))
P = meshgen.smooth_displace(P, 0.05)
nv, nf = len(P), len(T)
dm = DeviceMesh(P, T)
dm.set_surface_tension(np.ones(nf))
dm.set_bending_params(np.ones(nv), np.zeros(nv))
dm.set_params(modules=L.MS_MOD_SURFACE | L.MS_MOD_BENDING)
step, acc, err = 1e-6, 0, None
try:
    for i in range(int(os.environ.get("PROBE_STEPS", "120"))):
        r = dm.step(stepper=L.MS_STEPPER_CG, step_size=step, reuse_energy0=2)
        step = r.next_step
        acc += int(r.success)
        if not r.success:
            dm.reset_stepper()
except L.MembraneHipError as e:
    err = str(e)[:300]
buf = (ctypes.c_ulonglong * 32)()
L.lib().ms_debug_read_probe(buf)
v = [int(x) for x in buf]
print(json.dumps({"workgroups_probed": v[0], "gated_launches_probed": v[3], "scalar_path_differs": v[1],
                  "vector_path_differs": v[2], "scalar_differs_by_xcc": v[8:16], "vector_differs_by_xcc": v[16:24],
                  "accepted_steps": acc, "queue": dm.queue_stats(), "error": err}))
