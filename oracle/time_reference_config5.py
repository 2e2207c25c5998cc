#!/usr/bin/env python3
"""Test infrastructure, not product: times the REFERENCE itself (imported from /root/reference, Fortran kernels as
oracle/gen_golden.py loads them) on BASELINE config 5's deck exactly as gen_config5 sets it up -- one
relax_leaflet_tilts call and the deck's `g` steps -- on this machine's host cores.  The GPU counterpart is
tools/bench_config5.py.  usage: PYTHONDONTWRITEBYTECODE=1 python3 oracle/time_reference_config5.py [STEPS=8]"""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8
sys.argv = [sys.argv[0]]
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  (sets up sys.path for the reference and loads its Fortran kernels)

deck = os.path.join(gg.args.reference, "meshes", "caveolin", "kozlov_1disk_3d_tensionless_bilayer_profile.yaml")
m = gg.parse_geometry(gg.load_data(deck))
mods = list(m.energy_modules)
m.constraint_modules = []
m.global_parameters.set("mesh_quality_auto_repair_enabled", False)
mz = gg.Minimizer(m, m.global_parameters, gg.GradientDescent(), gg.EnergyModuleManager(mods),
                  gg.ConstraintModuleManager([]), quiet=True, step_size=float(m.global_parameters.get("step_size")))
mz._relax_leaflet_tilts(positions=m.positions_view(), mode="coupled")  # warm
t0 = time.perf_counter()
n_rel = 3
for _ in range(n_rel):
    m.increment_version()
    mz._relax_leaflet_tilts(positions=m.positions_view(), mode="coupled")
t_rel = (time.perf_counter() - t0) / n_rel
mz.minimize(2)
t0 = time.perf_counter()
mz.minimize(steps)
dt = time.perf_counter() - t0
print(f"reference on this host, config 5 deck ({len(m.vertex_ids)} vertices): relaxation {1e3 * t_rel:.1f} ms, "
      f"{1e3 * dt / steps:.1f} ms per step ({steps / dt:.2f} steps/s)")
