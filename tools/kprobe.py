#!/usr/bin/env python3
"""Times the two tile kernels alone on the headline mesh (HIP events inside the library):
K_A as the trial pass with factor write, K_C with the fused per-row-PR direction pass.
Used for A/B and phase-ablation builds (MEMBRANE_HIP_LIB=build_variants/lib_x.so)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from membrane_solver_amd import _lib as L  # noqa: E402
from membrane_solver_amd import meshgen  # noqa: E402
from membrane_solver_amd.device import DeviceMesh  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--freq", type=int, default=320)
ap.add_argument("--tile", type=int, default=256)
ap.add_argument("--reps", type=int, default=60)
ap.add_argument("--stamps", default="", help="write the diagnostic build's per-workgroup stamps of the LAST K_C launch here (.npz)")
ap.add_argument("--last", default="gradient", choices=("gradient", "energy"), help="kernel whose stamps are read")
ap.add_argument("--tag", default=os.path.basename(os.environ.get("MEMBRANE_HIP_LIB", "default")))
args = ap.parse_args()

P, T = meshgen.icosphere(args.freq)
P = meshgen.smooth_displace(P, 0.05)
nv, nf = len(P), len(T)
dm = DeviceMesh(P, T, tile_vertices=args.tile)
dm.set_surface_tension(np.ones(nf))
dm.set_bending_params(np.ones(nv), np.zeros(nv))
dm.set_params(modules=L.MS_MOD_SURFACE | L.MS_MOD_BENDING)
dm.phase_energy(write_bending_factors=True)
dm.phase_gradient_direction(L.MS_STEPPER_CG, False)
dm.phase_energy(use_direction=True, alpha=1e-7, write_trial=True, write_bending_factors=True)
dm.phase_accept(True)  # x <- trial point, CG history <- (g, d)
for _ in range(5):
    dm.phase_energy(use_direction=True, alpha=1e-9, write_trial=True, write_bending_factors=True)
    dm.phase_set_factors_valid(True)
    dm.phase_gradient_direction(L.MS_STEPPER_CG, True)
dm.fetch_scalars()
dm.profile_enable(True)
dm.profile_read()
def time_energy(**kw):
    dm.profile_read()
    for _ in range(args.reps):
        dm.phase_energy(**kw)
    dm.fetch_scalars()
    pr = dm.profile_read()
    return 1e3 * pr["energy"][0] / max(1, pr["energy"][1])


ka_x = time_energy(write_bending_factors=True)                                    # pass at x with factors
ka_e = time_energy()                                                              # energy only at x
ka_t = time_energy(use_direction=True, alpha=1e-9, write_trial=True)              # trial, xt write, no factors
dm.profile_read()
for _ in range(args.reps):
    dm.phase_set_factors_valid(True)
    dm.phase_gradient_direction(L.MS_STEPPER_CG, True)
dm.fetch_scalars()
prof = dm.profile_read()
kc = 1e3 * (prof["gradient"][0] + prof["gradient_lean"][0]) / max(1, prof["gradient"][1] + prof["gradient_lean"][1])
ka = time_energy(use_direction=True, alpha=1e-9, write_trial=True, write_bending_factors=True)
if args.last == "gradient":
    dm.phase_set_factors_valid(True)
    dm.phase_gradient_direction(L.MS_STEPPER_CG, True)
    dm.fetch_scalars()
sc = dm.fetch_scalars()
assert sc[0] > 12.0 and sc[2] > 25.0, sc[:4]  # a sphere-like surface, not a degenerate state
print(f"{args.tag:24s} K_A trial+factors {ka:6.2f}  at-x+factors {ka_x:6.2f}  energy-only {ka_e:6.2f}  trial {ka_t:6.2f} | "
      f"K_C {kc:6.2f} us  (nf={nf}, {args.reps} launches each)", flush=True)

if args.stamps:
    import ctypes
    lib = L.lib()
    n_tiles = dm.tile_stats()["n_tiles"]
    buf = np.zeros((n_tiles, 8), dtype=np.uint64)
    rc = lib.ms_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(n_tiles))
    assert rc == 0, rc
    np.savez_compressed(args.stamps, stamps=buf)
    cnt = np.zeros(1, dtype=np.uint64)
    lib.ms_debug_read_stamps(cnt.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(-1))
    print("  rare vertex-normal path entered by", int(cnt[0]), "lanes in all launches so far; energies", dm.fetch_scalars()[:4])
    t = buf[:, :4].astype(np.int64)
    t0 = t[:, 0].min()
    d = (t - t0) * 0.01  # us (100 MHz)
    print("  kernel span %.1f us; per workgroup: stage-in %.2f  loop %.2f  epilogue %.2f  total %.2f us (means)" % (
        d[:, 3].max(), (d[:, 1] - d[:, 0]).mean(), (d[:, 2] - d[:, 1]).mean(), (d[:, 3] - d[:, 2]).mean(),
        (d[:, 3] - d[:, 0]).mean()))
