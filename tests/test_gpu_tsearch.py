"""The search pass of the tilt relaxations (csrc/ms_tsearch.inc: several step sizes of the backtracking ladder of
runtime/steppers/tilt_relaxation.py:326-347, 380-398, 918-973, 1150-1230 in one launch, every tilt module of both
leaflets) against the launch-per-module-and-trial path it replaces on multi-tile meshes (MS_TSEARCH=0): with fixed-order
sums the relaxed fields, iteration and evaluation counts are EQUAL bit for bit -- the pass forms the same trial rows and
the same per-tile partial sums.  (Parity with the reference / the oracle port is what test_gpu_leaflet.py and
test_gpu_bending_tilt.py check through the same entry points, which now run this pass.)"""

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mesh(freq=14):
    from membrane_solver_amd import meshgen

    P, T = meshgen.icosphere(freq)
    return meshgen.smooth_displace(P, 0.05), T, None


def _tangent(P, T, seed, amp):
    from oracle import minimizer_port as mp

    rng = np.random.default_rng(seed)
    nrm = mp.unit_vertex_normals(P, T)
    t = amp * rng.normal(size=P.shape)
    return t - np.einsum("ij,ij->i", t, nrm)[:, None] * nrm


class _Env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        for k, v in self.kv.items():
            os.environ[k] = v

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _single(P, T, tl, fixed, smooth, tsearch, step, solver, iters, tile, det=True, calls=2):
    from membrane_solver_amd import _lib as L
    from membrane_solver_amd.device import DeviceMesh

    with _Env(MS_TSEARCH="1" if tsearch else "0", MS_DETERMINISTIC="1" if det else "0"):
        dm = DeviceMesh(P, T, tile_vertices=tile)
    nv = len(P)
    dm.set_surface_tension(np.ones(len(T)))
    dm.set_bending_params(np.full(nv, 1.3), np.full(nv, 0.2))
    dm.set_tilts(tl, 2.0)
    dm.set_tilt_fixed(fixed)
    mods = L.MS_MOD_SURFACE | L.MS_MOD_TILT | L.MS_MOD_BENDING_TILT
    if smooth:
        dm.set_tilt_smoothness(0.6)
        mods |= L.MS_MOD_TILT_SMOOTH
    dm.set_params(modules=mods)
    out = []
    for _ in range(calls):  # (a later call starts where the search pattern before it left the predictor)
        it, ev = dm.relax_tilts(solver=solver, max_iters=iters, step_size=step, jacobi=True)
        out.append((it, ev, dm.get_tilts().copy()))
    st = dm.tsearch_stats()
    dm.close()
    return out, st


@pytest.mark.parametrize("tile", [64, 256])
@pytest.mark.parametrize("solver,step,smooth", [("cg", 0.05, False), ("gd", 0.05, True), ("cg", 40.0, True), ("gd", 1.0e3, False)])
def test_single_field_search_pass_equals_trial_launches(tile, solver, step, smooth):
    """step 0.05: the first step size passes; 40 / 1e3: the ladder halves several times (8-trial passes, a second
    pass behind them)."""
    P, T, _ = _mesh(14)
    tl = _tangent(P, T, 5, 0.2)
    fixed = np.zeros(len(P), bool)
    fixed[::11] = True
    a, st_a = _single(P, T, tl, fixed, smooth, True, step, solver, 4, tile)
    b, st_b = _single(P, T, tl, fixed, smooth, False, step, solver, 4, tile)
    assert st_a["passes"] > 0 and st_b["passes"] == 0
    if step > 1.0:
        assert st_a["step_sizes"] > st_a["passes"]  # multi-trial passes ran
    for (ia, ea, ta), (ib, eb, tb) in zip(a, b):
        assert (ia, ea) == (ib, eb)
        assert np.array_equal(ta, tb)
    assert np.allclose(a[-1][2][fixed], tl[fixed], rtol=0, atol=1e-12)  # clamped rows: projected once, never moved


def test_whole_ladder_rejected():
    """A step size far too large: every one of the twelve halvings is rejected (13 evaluations, nothing moves).  From
    the second such search on both passes are queued back to back."""
    P, T, _ = _mesh(14)
    tl = _tangent(P, T, 5, 0.2)
    fixed = np.zeros(len(P), bool)
    fixed[::11] = True
    a, st_a = _single(P, T, tl, fixed, True, True, 1.0e30, "gd", 3, 256, calls=3)
    b, _ = _single(P, T, tl, fixed, True, False, 1.0e30, "gd", 3, 256, calls=3)
    for (ia, ea, ta), (ib, eb, tb) in zip(a, b):
        assert (ia, ea) == (ib, eb) == (1, 13)
        assert np.array_equal(ta, tb)
    assert np.allclose(a[0][2], a[-1][2], rtol=0, atol=1e-14)  # (every call projects once more)
    assert st_a == {"passes": 3 + 3 + 2 + 2, "step_sizes": 3 * 12}  # gradient passes; 1 + 8 + 3, then 8 + 4 twice


def _leaflets(P, T, tin, tout, fin, mods_bt, tsearch, step, solver, iters, tile, det=True):
    from membrane_solver_amd import _lib as L
    from membrane_solver_amd.device import DeviceMesh

    with _Env(MS_TSEARCH="1" if tsearch else "0", MS_DETERMINISTIC="1" if det else "0"):
        dm = DeviceMesh(P, T, tile_vertices=tile)
    dm.set_surface_tension(np.ones(len(T)))
    dm.set_leaflet_tilts("in", tin, tilt_fixed=fin, tilt_modulus=1.3, smoothness=0.4)
    dm.set_leaflet_tilts("out", tout, tilt_modulus=0.7, mass_mode="consistent", smoothness=0.25)
    mods = L.MS_MOD_SURFACE | L.MS_MOD_TILT_IN | L.MS_MOD_TILT_OUT | L.MS_MOD_TILT_SMOOTH_IN | L.MS_MOD_TILT_SMOOTH_OUT
    if mods_bt:
        dm.set_leaflet_bending("in", 1.2, 0.05)
        dm.set_leaflet_bending("out", 0.8, -0.1)
        mods |= L.MS_MOD_BENDING_TILT_IN | L.MS_MOD_BENDING_TILT_OUT
    dm.set_params(modules=mods)
    out = []
    for _ in range(2):
        it, ev = dm.relax_leaflet_tilts(solver=solver, max_iters=iters, step_size=step, jacobi=True)
        out.append((it, ev, dm.get_leaflet_tilts("in").copy(), dm.get_leaflet_tilts("out").copy()))
    st = dm.tsearch_stats()
    dm.close()
    return out, st


@pytest.mark.parametrize("tile", [64, 256])
@pytest.mark.parametrize("solver,step,bt", [("cg", 0.05, True), ("gd", 0.05, False), ("cg", 60.0, True), ("gd", 500.0, True)])
def test_leaflet_search_pass_equals_trial_launches(tile, solver, step, bt):
    P, T, _ = _mesh(14)
    tin, tout = _tangent(P, T, 7, 0.2), _tangent(P, T, 8, 0.15)
    fin = np.zeros(len(P), bool)
    fin[::13] = True
    a, st_a = _leaflets(P, T, tin, tout, fin, bt, True, step, solver, 3, tile)
    b, st_b = _leaflets(P, T, tin, tout, fin, bt, False, step, solver, 3, tile)
    assert st_a["passes"] > 0 and st_b["passes"] == 0
    for (ia, ea, xa, ya), (ib, eb, xb, yb) in zip(a, b):
        assert (ia, ea) == (ib, eb)
        assert np.array_equal(xa, xb) and np.array_equal(ya, yb)


@pytest.mark.parametrize("step", [0.05, 60.0])
def test_default_mode_agrees_with_trial_launches(step):
    """Default mode (per-vertex sums by LDS atomics, as in the shape kernels): the gradient pass adds a corner's module
    terms and accumulates them per owned vertex with ds_add_f64 -- same counts, fields equal to rounding."""
    from conftest import relerr

    P, T, _ = _mesh(14)
    tin, tout = _tangent(P, T, 7, 0.2), _tangent(P, T, 8, 0.15)
    fin = np.zeros(len(P), bool)
    fin[::13] = True
    a, st_a = _leaflets(P, T, tin, tout, fin, True, True, step, "cg", 3, 256, det=False)
    b, _ = _leaflets(P, T, tin, tout, fin, True, False, step, "cg", 3, 256, det=False)
    assert st_a["passes"] > 0
    for (ia, ea, xa, ya), (ib, eb, xb, yb) in zip(a, b):
        assert (ia, ea) == (ib, eb)
        assert relerr(xa, xb) < 1e-11 and relerr(ya, yb) < 1e-11
    tl = _tangent(P, T, 5, 0.2)
    fixed = np.zeros(len(P), bool)
    fixed[::11] = True
    a, _ = _single(P, T, tl, fixed, True, True, step, "cg", 4, 256, det=False)
    b, _ = _single(P, T, tl, fixed, True, False, step, "cg", 4, 256, det=False)
    for (ia, ea, ta), (ib, eb, tb) in zip(a, b):
        assert (ia, ea) == (ib, eb)
        assert relerr(ta, tb) < 1e-11


def test_search_pass_declines_what_it_does_not_cover():
    """A lone tilt module on the single field (k_tilt keeps its per-facet form) and one-tile meshes (the interpreter's
    relaxation program) do not go through the pass."""
    from membrane_solver_amd import _lib as L
    from membrane_solver_amd.device import DeviceMesh

    P, T, _ = _mesh(14)
    tl = _tangent(P, T, 9, 0.2)
    dm = DeviceMesh(P, T, tile_vertices=64)
    dm.set_surface_tension(np.ones(len(T)))
    dm.set_tilts(tl, 2.0)
    dm.set_tilt_smoothness(0.5)
    dm.set_params(modules=L.MS_MOD_SURFACE | L.MS_MOD_TILT | L.MS_MOD_TILT_SMOOTH)
    dm.relax_tilts(solver="cg", max_iters=2, step_size=0.05, jacobi=True)
    assert dm.tsearch_stats()["passes"] == 0
    dm.close()
    P1, T1, _ = _mesh(4)
    dm = DeviceMesh(P1, T1)
    assert dm.tile_stats()["n_tiles"] == 1
    dm.set_surface_tension(np.ones(len(T1)))
    dm.set_bending_params(np.full(len(P1), 1.3), np.full(len(P1), 0.2))
    dm.set_tilts(_tangent(P1, T1, 2, 0.2), 2.0)
    dm.set_params(modules=L.MS_MOD_SURFACE | L.MS_MOD_TILT | L.MS_MOD_BENDING_TILT)
    dm.relax_tilts(solver="cg", max_iters=2, step_size=0.05, jacobi=True)
    assert dm.tsearch_stats()["passes"] == 0
    dm.close()


def _leaflet_ctx(P, T, tin, tout, fin, tsearch, tile):
    from membrane_solver_amd import _lib as L
    from membrane_solver_amd.device import DeviceMesh

    with _Env(MS_TSEARCH="1" if tsearch else "0", MS_DETERMINISTIC="1"):
        dm = DeviceMesh(P, T, tile_vertices=tile)
    dm.set_surface_tension(np.ones(len(T)))
    dm.set_leaflet_tilts("in", tin, tilt_fixed=fin, tilt_modulus=1.3, smoothness=0.4)
    dm.set_leaflet_tilts("out", tout, tilt_modulus=0.7, smoothness=0.25)
    dm.set_leaflet_bending("in", 1.2, 0.05)
    dm.set_leaflet_bending("out", 0.8, -0.1)
    dm.set_params(modules=L.MS_MOD_SURFACE | L.MS_MOD_TILT_IN | L.MS_MOD_TILT_OUT | L.MS_MOD_TILT_SMOOTH_IN |
                  L.MS_MOD_TILT_SMOOTH_OUT | L.MS_MOD_BENDING_TILT_IN | L.MS_MOD_BENDING_TILT_OUT)
    return dm


@pytest.mark.parametrize("tile", [64, 256])
def test_shape_trial_energies_through_the_pass(tile):
    """Energy-only evaluations (the energy at x, the trials of the shape line search) take the tilt modules' facet
    passes as ONE launch; the energies and the steps they decide agree with the launch-per-module path (energies 1e-13,
    accepted step sizes and trial counts equal)."""
    from membrane_solver_amd import _lib as L

    P, T, _ = _mesh(14)
    tin, tout = _tangent(P, T, 7, 0.2), _tangent(P, T, 8, 0.15)
    fin = np.zeros(len(P), bool)
    fin[::13] = True
    logs = []
    for on in (True, False):
        dm = _leaflet_ctx(P, T, tin, tout, fin, on, tile)
        log = [dm.energy().copy()]
        for _ in range(3):
            r = dm.step(stepper=L.MS_STEPPER_GD, step_size=1e-3, tol=0.0)
            log.append((r.success, r.alpha, r.trials, r.energy))
            dm.relax_leaflet_tilts(solver="cg", max_iters=2, step_size=0.05, jacobi=True)
        log.append(dm.energy().copy())
        logs.append((log, dm.tsearch_stats(), dm.get_positions().copy()))
        dm.close()
    (la, sa, xa), (lb, sb, xb) = logs
    assert sa["passes"] > sb["passes"] == 0
    assert np.allclose(la[0], lb[0], rtol=1e-13, atol=0) and np.allclose(la[-1], lb[-1], rtol=1e-12, atol=0)
    for ra, rb in zip(la[1:-1], lb[1:-1]):
        assert ra[:3] == rb[:3]
        assert abs(ra[3] - rb[3]) <= 1e-12 * abs(rb[3])
    assert np.allclose(xa, xb, rtol=0, atol=1e-13)


def test_full_size_relaxations_equal_trial_launches():
    """BASELINE's headline size (2 048 000 facets, 4001 tiles): one two-leaflet relaxation and one single-field
    relaxation with a ladder that halves, through the passes and through the launch-per-module-and-trial path, fixed-order
    sums: counts and fields equal bit for bit."""
    from membrane_solver_amd import meshgen

    P, T = meshgen.icosphere(320)
    P = meshgen.smooth_displace(P, 0.05)
    rng = np.random.default_rng(11)
    r = P / np.linalg.norm(P, axis=1)[:, None]  # (close enough to the vertex normals for a tangent start)
    def tangent(amp):
        t = amp * rng.normal(size=P.shape)
        return t - np.einsum("ij,ij->i", t, r)[:, None] * r
    tin, tout, tl = tangent(0.2), tangent(0.15), tangent(0.2)
    fin = np.zeros(len(P), bool)
    fin[::13] = True
    a, st = _leaflets(P, T, tin, tout, fin, True, True, 0.05, "cg", 2, 256)
    b, _ = _leaflets(P, T, tin, tout, fin, True, False, 0.05, "cg", 2, 256)
    assert st["passes"] > 0
    for (ia, ea, xa, ya), (ib, eb, xb, yb) in zip(a, b):
        assert (ia, ea) == (ib, eb)
        assert np.array_equal(xa, xb) and np.array_equal(ya, yb)
    a, st = _single(P, T, tl, fin, True, True, 40.0, "cg", 2, 256)
    b, _ = _single(P, T, tl, fin, True, False, 40.0, "cg", 2, 256)
    assert st["step_sizes"] > st["passes"]
    for (ia, ea, ta), (ib, eb, tb) in zip(a, b):
        assert (ia, ea) == (ib, eb)
        assert np.array_equal(ta, tb)


def test_full_size_leaflet_relaxation_matches_the_oracle_port():
    """2 048 000 facets: one two-leaflet relaxation (Jacobi CG, two inner steps) through the gradient and search
    passes against the CPU oracle port of relax_leaflet_tilts (runtime/steppers/tilt_relaxation.py:426-1478): the same
    iteration and evaluation counts, fields to 1e-9 of their norm."""
    from conftest import relerr
    from membrane_solver_amd import _lib as L
    from membrane_solver_amd import meshgen
    from membrane_solver_amd.device import DeviceMesh
    from oracle import minimizer_port as mp

    P, T = meshgen.icosphere(320)
    P = meshgen.smooth_displace(P, 0.05)
    rng = np.random.default_rng(5)
    nrm = mp.unit_vertex_normals(P, T)
    tin = 0.2 * rng.normal(size=P.shape)
    tin -= np.einsum("ij,ij->i", tin, nrm)[:, None] * nrm
    tout = 0.15 * rng.normal(size=P.shape)
    tout -= np.einsum("ij,ij->i", tout, nrm)[:, None] * nrm
    fin = np.zeros(len(P), bool)
    fin[::13] = True
    gp = {"tilt_modulus_in": 1.3, "tilt_modulus_out": 0.7, "bending_modulus": 0.4, "tilt_solve_mode": "nested",
          "tilt_solver": "cg", "tilt_step_size": 0.05, "tilt_inner_steps": 2}
    mods = ["tilt_in", "tilt_out", "tilt_smoothness_in", "tilt_smoothness_out"]
    p = mp.Problem(positions=P, tri=T, tilts_in=tin, tilts_out=tout, tilt_fixed_in=fin, energy_modules=mods, gp=gp)
    dm = DeviceMesh(P, T)
    dm.set_leaflet_tilts("in", tin, tilt_fixed=fin, tilt_modulus=1.3, smoothness=0.4)
    dm.set_leaflet_tilts("out", tout, tilt_modulus=0.7, smoothness=0.4)
    dm.set_params(modules=L.MS_MOD_TILT_IN | L.MS_MOD_TILT_OUT | L.MS_MOD_TILT_SMOOTH_IN | L.MS_MOD_TILT_SMOOTH_OUT)
    stats = mp.relax_leaflet_tilts(p, P)
    it, ev = dm.relax_leaflet_tilts(solver="cg", max_iters=2, step_size=0.05, jacobi=True)
    assert dm.tsearch_stats()["passes"] > 0
    assert (it, ev) == (stats["iters"], stats["evals"])
    assert relerr(dm.get_leaflet_tilts("in"), p.tilts_in) < 1e-9
    assert relerr(dm.get_leaflet_tilts("out"), p.tilts_out) < 1e-9
    dm.close()


def test_full_size_single_field_relaxation_with_halvings_matches_the_oracle_port():
    """2 048 000 facets, tilt + bending_tilt + tilt_smoothness, a step size that makes the ladder halve several times
    (multi-trial passes): counts equal the oracle port's relax_tilts (tilt_relaxation.py:237-424), field to 1e-9."""
    from conftest import relerr
    from membrane_solver_amd import _lib as L
    from membrane_solver_amd import meshgen
    from membrane_solver_amd.device import DeviceMesh
    from oracle import minimizer_port as mp

    P, T = meshgen.icosphere(320)
    P = meshgen.smooth_displace(P, 0.05)
    nv = len(P)
    rng = np.random.default_rng(6)
    nrm = mp.unit_vertex_normals(P, T)
    tl = 0.2 * rng.normal(size=P.shape)
    tl -= np.einsum("ij,ij->i", tl, nrm)[:, None] * nrm
    tfix = np.zeros(nv, bool)
    tfix[::11] = True
    gp = {"bending_modulus": 1.3, "spontaneous_curvature": 0.2, "tilt_rigidity": 2.0, "tilt_smoothness_rigidity": 0.7,
          "tilt_solve_mode": "nested", "tilt_solver": "cg", "tilt_step_size": 40.0, "tilt_inner_steps": 2}
    p = mp.Problem(positions=P, tri=T, tilts=tl, tilt_fixed=tfix, energy_modules=["tilt", "tilt_smoothness", "bending_tilt"],
                   gp=gp)
    st = mp.relax_tilts(p, p.positions)
    dm = DeviceMesh(P, T)
    dm.set_surface_tension(np.ones(len(T)))
    dm.set_bending_params(np.full(nv, 1.3), np.full(nv, 0.2))
    dm.set_tilts(tl, 2.0)
    dm.set_tilt_smoothness(0.7)
    dm.set_tilt_fixed(tfix)
    dm.set_params(modules=L.MS_MOD_TILT | L.MS_MOD_BENDING_TILT | L.MS_MOD_TILT_SMOOTH)
    iters, evals = dm.relax_tilts(solver="cg", max_iters=2, step_size=40.0, jacobi=True)
    ts = dm.tsearch_stats()
    assert ts["step_sizes"] > ts["passes"] > 0  # (passes with several step sizes ran)
    assert (iters, evals) == (st["iters"], st["evals"])
    assert relerr(dm.get_tilts(), p.tilts) < 1e-9
    dm.close()
