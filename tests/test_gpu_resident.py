"""The resident step kernel (csrc/ms_resident.inc): many steps of the surface (+ volume row) / gradient-descent lane in
one launch.  It restates k_energy / k_gradient / k_direction / k_reduce for that lane, so with fixed-order vertex sums
its step logs and positions must be those of the kernel-per-phase path (MS_RESIDENT=0)."""
from __future__ import annotations

import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _params(L, *, step_size, drift, target, tol=1e-9, project=0, vol_tol=1e-3, max_iter=10):
    mp = L.ms_minimize_params()
    mp.stepper = L.ms_stepper_params(int(L.MS_STEPPER_GD), max_iter, 0.7, 1e-4, 1.5, 10.0, 10, 0.0, 2, 0, 0)
    mp.step_size = step_size
    mp.tol = tol
    mp.fixed_step_mode = 0
    mp.fixed_step = step_size
    mp.max_zero_steps = 10
    mp.step_size_floor = 1e-8
    mp.drift_check = 1 if drift else 0
    mp.target_volume = float(target)
    mp.volume_tolerance = vol_tol
    mp.project_on_drift = project
    mp.relax_tilts = 0
    return mp


def _run(monkeypatch, resident, *, freq, volume, n_steps, step_size, fixed_every=0, noise=0.0, vol_tol=1e-3, project=0,
         tol=1e-9):
    from membrane_solver_amd import _lib as L
    from membrane_solver_amd import meshgen
    from membrane_solver_amd.device import DeviceMesh

    monkeypatch.setenv("MS_RESIDENT", "1" if resident else "0")
    pos, tri = meshgen.icosphere(freq)
    pos = meshgen.smooth_displace(pos, 0.05)
    if noise:
        pos = pos + noise * np.random.default_rng(5).standard_normal(pos.shape)
    fixed = None
    if fixed_every:
        fixed = np.zeros(len(pos), dtype=np.uint8)
        fixed[::fixed_every] = 1
    dm = DeviceMesh(pos, tri, fixed=fixed, body_facets=np.ones(len(tri), dtype=np.uint8) if volume else None)
    dm.set_surface_tension(np.full(len(tri), 1.0))
    mods = L.MS_MOD_SURFACE | (L.MS_CON_VOLUME if volume else 0)
    V0 = 0.0
    dm.set_params(modules=mods)
    if volume:
        dm.energy()
        V0 = float(dm.fetch_scalars()[L.MS_S_VOL])
        dm.set_params(modules=mods, target_volume=V0)
    mp = _params(L, step_size=step_size, drift=volume, target=V0, vol_tol=vol_tol, project=project, tol=tol)
    out, log = dm.minimize(mp, n_steps, want_log=True)
    res = {"log": log.copy(), "x": dm.get_positions(), "stats": dm.resident_stats(), "accepted": out.accepted,
           "trials": out.trials, "iterations": out.iterations, "step_size": out.step_size,
           "energy_current": (out.energy_current if out.energy_current_valid else None), "converged": out.converged}
    dm.close()
    return res


CASES = {
    "surface_volume_row": dict(freq=20, volume=True, n_steps=60, step_size=1e-3),
    "surface_only": dict(freq=20, volume=False, n_steps=40, step_size=1e-3),
    "fixed_rows": dict(freq=16, volume=True, n_steps=40, step_size=2e-3, fixed_every=7),
    # an over-long first step: the first searches backtrack, some run into the guard range (declined steps)
    "backtracking_and_guard": dict(freq=16, volume=True, n_steps=40, step_size=0.3, noise=2e-3),
    # a tolerance the volume drifts past: the kernel stops after the step, the host projects and resets
    "volume_drift": dict(freq=12, volume=True, n_steps=40, step_size=5e-3, vol_tol=1e-7, project=1),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_resident_steps_equal_the_kernel_per_phase_path(name, deterministic, monkeypatch):
    ref = _run(monkeypatch, False, **CASES[name])
    got = _run(monkeypatch, True, **CASES[name])
    assert ref["stats"]["launches"] == 0
    assert got["stats"]["co_resident"] == 1 and got["stats"]["steps"] > 0, got["stats"]
    assert got["iterations"] == ref["iterations"] and got["accepted"] == ref["accepted"] and got["trials"] == ref["trials"]
    assert np.array_equal(got["log"], ref["log"])
    assert np.array_equal(got["x"], ref["x"])
    assert got["step_size"] == ref["step_size"]
    if got["energy_current"] is not None and ref["energy_current"] is not None:
        assert got["energy_current"] == ref["energy_current"]


def test_resident_default_mode_agrees_to_rounding(monkeypatch):
    """LDS-atomic vertex sums (the default): the order of a vertex's additions varies from run to run in both paths."""
    ref = _run(monkeypatch, False, **CASES["surface_volume_row"])
    got = _run(monkeypatch, True, **CASES["surface_volume_row"])
    assert got["stats"]["steps"] > 0
    assert np.array_equal(got["log"][:, 0], ref["log"][:, 0]) and np.array_equal(got["log"][:, 7], ref["log"][:, 7])
    assert np.allclose(got["log"][:, 2], ref["log"][:, 2], rtol=1e-12, atol=0)
    assert np.abs(got["x"] - ref["x"]).max() <= 1e-11


def test_resident_convergence_goes_through_the_ordinary_path(deterministic, monkeypatch):
    ref = _run(monkeypatch, False, freq=12, volume=True, n_steps=30, step_size=1e-3, tol=0.5)
    got = _run(monkeypatch, True, freq=12, volume=True, n_steps=30, step_size=1e-3, tol=0.5)
    assert ref["converged"] == got["converged"] and got["iterations"] == ref["iterations"]
    assert np.array_equal(got["log"], ref["log"]) and np.array_equal(got["x"], ref["x"])
