#!/usr/bin/env python3
"""How far the headline number depends on the specialised kernel instances: the headline problem (2 048 000 facets,
surface + bending, CG, ms_minimize 200 steps after 60) with the parameters the lean instances ask for -- uniform
tension / modulus / spontaneous curvature, closed surface, nothing pinned -- and with each of them broken in turn.
usage: python3 tools/variant_bench.py [FREQ=320]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from membrane_solver_amd import _lib as L
from membrane_solver_amd.device import DeviceMesh

freq = int(sys.argv[1]) if len(sys.argv) > 1 else 320
P, T = bench.bench_mesh(freq)
nv, nf = len(P), len(T)
rng = np.random.default_rng(3)
cases = {
    "uniform (headline)": dict(),
    "tension varies per facet (+-5 %)": dict(gamma=1.0 + 0.05 * rng.uniform(-1, 1, nf)),
    "modulus varies per vertex (+-5 %)": dict(kappa=1.0 + 0.05 * rng.uniform(-1, 1, nv)),
    "c0 = 0.1 uniform": dict(c0=np.full(nv, 0.1)),
    "1 % of the vertices pinned": dict(fixed=(rng.uniform(size=nv) < 0.01)),
}


def open_cap():
    """the same sphere with a polar cap cut off: a boundary ring (pinned, as a rim usually is)"""
    keep = (P[T].max(axis=1)[:, 2] < 0.97)
    T2 = T[keep]
    used = np.zeros(nv, bool)
    used[T2.ravel()] = True
    remap = np.cumsum(used) - 1
    P2, T2 = P[used], remap[T2].astype(np.int32)
    e = np.sort(np.concatenate([T2[:, [0, 1]], T2[:, [1, 2]], T2[:, [2, 0]]]), axis=1)
    key = e[:, 0].astype(np.int64) * len(P2) + e[:, 1]
    uniq, cnt = np.unique(key, return_counts=True)
    b = uniq[cnt == 1]
    isb = np.zeros(len(P2), bool)
    isb[b // len(P2)] = True
    isb[b % len(P2)] = True
    return np.ascontiguousarray(P2), np.ascontiguousarray(T2), isb


cases["open surface (polar cap cut off, rim pinned)"] = dict(open=True)
for name, kw in cases.items():
    if kw.get("open"):
        P, T, isb = open_cap()
        nv, nf = len(P), len(T)
        dm = DeviceMesh(P, T, fixed=isb, boundary=isb)
    else:
        dm = DeviceMesh(P, T, fixed=kw.get("fixed"))
    dm.set_surface_tension(kw.get("gamma", np.ones(nf)))
    dm.set_bending_params(kw.get("kappa", np.ones(nv)), kw.get("c0", np.zeros(nv)))
    dm.set_params(modules=L.MS_MOD_SURFACE | L.MS_MOD_BENDING)
    mp = L.ms_minimize_params()
    mp.stepper = L.ms_stepper_params(L.MS_STEPPER_CG, 10, 0.7, 1e-4, 1.5, 10.0, 10, 0.0, 2)
    mp.step_size, mp.tol = 1e-6 * (320.0 / freq) ** 2, 1e-9
    mp.fixed_step_mode, mp.fixed_step, mp.max_zero_steps, mp.step_size_floor = 0, 0.0, 10, 1e-12
    o, _ = dm.minimize(mp, 60)
    mp.step_size = float(o.step_size)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    o, _ = dm.minimize(mp, 200)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dm.profile_enable(True)
    mp.step_size = float(o.step_size)
    o2, _ = dm.minimize(mp, 40)
    prof = dm.profile_read()
    dm.profile_enable(False)
    ks = ", ".join(f"{k} {1e3 * ms / n:.1f} us x{n}" for k, (ms, n) in prof.items() if n)
    print(f"{name:36s} {200 / dt:8.0f} steps/s  accepted {int(o.accepted):3d} trials {int(o.trials):3d}  mismatches "
          f"{dm.queue_stats()['mismatches']}  | {ks}")
    dm.close()
