"""Pins the CPU oracle (oracle/ms_oracle.c + oracle/minimizer_port.py) to the
reference's own outputs (tests/golden/*.npz, made by oracle/gen_golden.py from
the reference with its Fortran kernels enabled), and to the reference's Fortran
kernels compiled in place (oracle/_ref) when those are present.

Tolerances: 1e-12 relative (max-norm) on energies and per-array outputs -- the
oracle differs from the reference only in summation order.
"""

import numpy as np
import pytest

from conftest import load_golden, relerr
from oracle import minimizer_port as mp
from oracle import ms_oracle as orc
from oracle import ref_fortran as rf

TOL = 1e-12


def test_kernel_cases_match_reference():
    g = load_golden("kernel_cases.npz")
    for n in ("4", "17", "deg"):
        gu, gv = orc.grad_cotan_batch(g[f"gc{n}_u"], g[f"gc{n}_v"])
        assert np.allclose(gu, g[f"gc{n}_gu"], atol=1e-10, rtol=1e-10)
        assert np.allclose(gv, g[f"gc{n}_gv"], atol=1e-10, rtol=1e-10)
    assert np.all(orc.grad_cotan_batch(g["gcdeg_u"], g["gcdeg_v"])[0][0] == 0.0)
    out = orc.apply_beltrami_laplacian(g["lap_weights"], g["lap_tri"], g["lap_field"])
    assert np.allclose(out, g["lap_out"], atol=1e-10, rtol=1e-10)
    div, area, g0, g1, g2 = orc.p1_triangle_divergence(g["div_pos"], g["div_tilts"], g["div_tri"])
    for a, k in ((div, "div_div"), (area, "div_area"), (g0, "div_g0"), (g1, "div_g1"), (g2, "div_g2")):
        assert np.allclose(a, g[k], atol=1e-10, rtol=1e-10), k
    k, A, w = orc.compute_curvature_data(g["curv_pos"], g["curv_tri"])
    assert np.allclose(k, g["curv_k"], atol=1e-10, rtol=1e-10)
    assert np.allclose(A, g["curv_A"], atol=1e-10, rtol=1e-10)
    assert np.allclose(w, g["curv_w"], atol=1e-10, rtol=1e-10)
    grad = np.zeros((3, 3))
    E = orc.surface_energy_and_gradient(g["surf_rt_pos"], g["surf_rt_tri"], g["surf_rt_gamma"], grad)
    assert abs(E - 1.0) < 1e-12  # reference tests/test_surface.py:61-93
    assert np.all(np.isfinite(grad))


@pytest.mark.parametrize("name", ["ico4", "ico8", "disk5", "ico5_noisy"])
def test_mesh_energies_and_gradients_match_reference(name):
    g = load_golden(f"mesh_{name}.npz")
    pos, tri, isb = g["positions"], g["tri"], g["is_boundary"]
    nv = pos.shape[0]
    grad = np.zeros_like(pos)
    E = orc.surface_energy_and_gradient(pos, tri, g["gamma"], grad)
    assert abs(E - g["E_surface"]) <= TOL * abs(g["E_surface"])
    assert relerr(grad, g["grad_surface"]) < TOL

    k, A, w = orc.compute_curvature_data(pos, tri)
    assert relerr(k, g["k_vecs"]) < TOL and relerr(A, g["A_vor"]) < TOL and relerr(w, g["weights"]) < TOL
    Aeff, va = orc.effective_areas(pos, tri, w, isb)
    assert relerr(Aeff, g["A_eff"]) < TOL and relerr(va, g["va_eff"]) < TOL
    assert relerr(orc.vertex_normals(pos, tri), g["normals"]) < TOL

    kappa = g["kappa"]
    for model, c0v in (("helfrich", 0.0), ("helfrich", 0.5), ("willmore", 0.0)):
        c0 = np.full(nv, c0v)
        for mode in ("analytic", "approx"):
            tag = f"{model}_c{int(c0v * 10)}_{mode}"
            grad = np.zeros_like(pos)
            E = orc.bending_energy_and_gradient(pos, tri, kappa, c0, isb, model=model, mode=mode, grad=grad)
            Eref = float(g[f"E_bend_{tag}"])
            assert abs(E - Eref) <= TOL * max(abs(Eref), 1.0), tag
            assert relerr(grad, g[f"grad_bend_{tag}"]) < 1e-11, tag
        Ea, pv = orc.bending_energy(pos, tri, kappa, c0, isb, model=model, per_vertex=True)
        assert relerr(pv, g[f"Earr_bend_{model}_c{int(c0v * 10)}"]) < TOL

    assert abs(orc.volume(pos, tri) - g["volume"]) <= TOL * abs(g["volume"])
    gC = np.zeros_like(pos)
    orc.volume_gradient(pos, tri, gC)
    assert relerr(gC, g["grad_volume"]) < TOL
    # penalty mode (modules/energy/volume.py:94-128)
    k_pen, V0 = float(g["volpen_k"]), float(g["volpen_target"])
    V = orc.volume(pos, tri)
    assert abs(0.5 * k_pen * (V - V0) ** 2 - g["E_volpen"]) <= TOL * abs(g["E_volpen"])
    gp = np.zeros_like(pos)
    orc.volume_gradient(pos, tri, gp, factor=k_pen * (V - V0))
    assert relerr(gp, g["grad_volpen"]) < TOL
    assert abs(mp.min_edge_length(pos, tri) - g["min_edge"]) <= TOL * g["min_edge"]

    if "tilts" in g:
        gs, gt = np.zeros_like(pos), np.zeros_like(pos)
        E = orc.tilt_energy_and_gradient(pos, g["tilts"], tri, float(g["k_tilt"]), gs, gt)
        assert abs(E - g["E_tilt"]) <= TOL * abs(g["E_tilt"])
        assert relerr(gs, g["grad_tilt_shape"]) < TOL and relerr(gt, g["grad_tilt_tilt"]) < TOL


def test_analytic_anchors():
    """E_surface(unit sphere) -> 4 pi, E_helfrich(c0=0, kappa=1) -> 8 pi (SURVEY 8c anchors)."""
    from membrane_solver_amd import meshgen

    P, T = meshgen.icosphere(24)
    nv = P.shape[0]
    E = orc.surface_energy_and_gradient(P, T, np.ones(T.shape[0]), None)
    assert abs(E - 4 * np.pi) / (4 * np.pi) < 2e-3
    Eb = orc.bending_energy(P, T, np.ones(nv), np.zeros(nv), np.zeros(nv, bool))
    assert abs(Eb - 8 * np.pi) / (8 * np.pi) < 2e-3


@pytest.mark.skipif(not rf.available(), reason="oracle/_ref not built (no reference checkout here)")
def test_c_restatement_matches_reference_fortran_objects():
    rng = np.random.default_rng(5)
    nv, nf = 40, 90
    pos = rng.normal(size=(nv, 3))
    tri = rng.integers(0, nv, size=(nf, 3), dtype=np.int32)
    gamma = rng.random(nf) + 0.5
    g1, g2 = np.zeros((nv, 3)), np.zeros((nv, 3))
    E1 = orc.surface_energy_and_gradient(pos, tri, gamma, g1)
    E2 = rf.surface_energy_and_gradient(pos, tri, gamma, g2)
    assert abs(E1 - E2) <= 1e-13 * abs(E2) and relerr(g1, g2) < 1e-13
    u, v = rng.normal(size=(nf, 3)), rng.normal(size=(nf, 3))
    for a, b in zip(orc.grad_cotan_batch(u, v), rf.grad_cotan_batch(u, v)):
        assert relerr(a, b) < 1e-13
    w = rng.normal(size=(nf, 3))
    fld = rng.normal(size=(nv, 3))
    assert relerr(orc.apply_beltrami_laplacian(w, tri, fld), rf.apply_beltrami_laplacian(w, tri, fld)) < 1e-13
    tl = rng.normal(size=(nv, 3))
    for a, b in zip(orc.p1_triangle_divergence(pos, tl, tri), rf.p1_triangle_divergence(pos, tl, tri)):
        assert relerr(a, b) < 1e-12
    for a, b in zip(orc.compute_curvature_data(pos, tri, True), rf.compute_curvature_data(pos, tri)):
        assert relerr(a, b) < 1e-12


def _problem_from_traj(g, mods, cons, gp):
    return mp.Problem(positions=g["positions0"], tri=g["tri"], gamma=g["gamma"],
                      tilts=g["tilts0"] if "tilts0" in g else None,
                      is_boundary=g["is_boundary"], fixed=g["fixed"], energy_modules=mods,
                      constraint_modules=cons,
                      target_volume=float(g["target_volume"]) if "target_volume" in g else None, gp=gp)


TRAJ = {
    "traj_cube_gd.npz": (["surface", "volume"], [], "gd",
                         {"volume_constraint_mode": "penalty", "volume_projection_during_minimization": True}),
    "traj_ico8_gd_surface_volume.npz": (["surface"], ["volume"], "gd",
                                        {"volume_constraint_mode": "lagrange",
                                         "volume_projection_during_minimization": False}),
    # a body with a target volume is present but no volume module is loaded: the
    # post-step drift check (minimizer.py:1478-1513) still resets the CG history.
    "traj_ico8_cg_surface_bending.npz": (["surface", "bending"], [], "cg",
                                         {"bending_modulus": 1.0, "volume_constraint_mode": "lagrange",
                                          "volume_projection_during_minimization": False}),
    "traj_ico8_cg_surface_bending_volume.npz": (["surface", "bending"], ["volume"], "cg",
                                                {"bending_modulus": 1.0, "spontaneous_curvature": 0.3,
                                                 "volume_constraint_mode": "lagrange",
                                                 "volume_projection_during_minimization": False}),
    "traj_ico4_gd_surface_tilt.npz": (["surface", "tilt"], [], "gd",
                                      {"tilt_rigidity": 2.5, "volume_constraint_mode": "lagrange",
                                       "volume_projection_during_minimization": False}),
    "traj_disk5_gd_surface_bending_fixed.npz": (["surface", "bending"], [], "gd",
                                                {"bending_modulus": 1.0, "volume_constraint_mode": "lagrange",
                                                 "volume_projection_during_minimization": False}),
    # the normal-rotation guard decides trials (topology.py:13-48; 72 of the reference's 73 guard calls reject) and
    # searches run out of their ten trials (line_search.py:425-426)
    "traj_ico6_cg_guard.npz": (["surface", "bending"], [], "cg",
                               {"bending_modulus": 0.5, "volume_constraint_mode": "lagrange",
                                "volume_projection_during_minimization": False}),
    # ten exhausted searches in a row, then the step size has shrunk enough to pass the guard and the Armijo test
    "traj_ico6_gd_exhaust.npz": (["surface", "bending"], [], "gd",
                                 {"bending_modulus": 0.5, "volume_constraint_mode": "lagrange",
                                  "volume_projection_during_minimization": False}),
    # volume_tolerance 1e-11: the Lagrange drift check (minimizer.py:1478-1513) fires after every accepted step and
    # volume.enforce_constraint (volume.py:69-149) re-projects the positions (8 calls: start, 6 steps, finalize)
    "traj_ico8_gd_volume_drift.npz": (["surface"], ["volume"], "gd",
                                      {"volume_constraint_mode": "lagrange", "volume_tolerance": 1.0e-11,
                                       "volume_projection_during_minimization": False}),
    # the same with a tilt module: every enforce is followed by project_tilts_to_tangent (minimizer.py:1224, :1506)
    "traj_ico4_gd_tilt_volume_drift.npz": (["surface", "tilt"], ["volume"], "gd",
                                           {"tilt_rigidity": 2.5, "volume_constraint_mode": "lagrange",
                                            "volume_tolerance": 1.0e-11,
                                            "volume_projection_during_minimization": False}),
    # the enforcer lane of the line search with the programmatic defaults of GlobalParameters() (Lagrange row AND
    # volume_projection_during_minimization on): every trial is projected onto the target volume before its energy is
    # taken (line_search.py:428-487, minimizer.py:1379)
    "traj_ico8_gd_volume_enforcer.npz": (["surface"], ["volume"], "gd",
                                         {"volume_constraint_mode": "lagrange",
                                          "volume_projection_during_minimization": True}),
    "traj_ico8_cg_bending_volume_enforcer.npz": (["surface", "bending"], ["volume"], "cg",
                                                 {"bending_modulus": 1.0, "spontaneous_curvature": 0.3,
                                                  "volume_constraint_mode": "lagrange",
                                                  "volume_projection_during_minimization": True}),
    # ConjugateGradient(precondition=True) (conjugate_gradient.py:74-76): backtracking, a restart, exhausted searches;
    # with the Lagrange row: non-descent directions (line_search.py:325-328) every other step
    "traj_ico8_cg_precondition.npz": (["surface", "bending"], [], "cgp",
                                      {"bending_modulus": 1.0, "volume_constraint_mode": "lagrange",
                                       "volume_projection_during_minimization": False}),
    "traj_ico8_cg_precondition_volume.npz": (["surface", "bending"], ["volume"], "cgp",
                                             {"bending_modulus": 1.0, "spontaneous_curvature": 0.3,
                                              "volume_constraint_mode": "lagrange",
                                              "volume_projection_during_minimization": False}),
}


@pytest.mark.parametrize("fname", sorted(TRAJ))
def test_minimizer_port_reproduces_reference_trajectory(fname):
    mods, cons, kind, gp = TRAJ[fname]
    g = load_golden(fname)
    gp = dict(gp)
    if "gp_volume_stiffness" in g:
        gp["volume_stiffness"] = float(g["gp_volume_stiffness"])
        gp["surface_tension"] = float(g["gp_surface_tension"])
    p = _problem_from_traj(g, mods, cons, gp)
    E0, grad0 = mp.energy_and_gradient(p, p.positions)
    assert abs(E0 - g["E0"]) <= 1e-12 * abs(g["E0"])
    assert relerr(grad0, g["grad0"]) < 1e-11
    stepper = mp.GradientDescent() if kind == "gd" else mp.ConjugateGradient(precondition=(kind == "cgp"))
    n = int(g["n_steps"])
    res = mp.minimize(p, stepper, n, step_size=float(g["step_size0"]))
    log = g["step_log"]
    got = np.array([[float(t["success"]), t["next_step"], t["E_accepted"]] for t in res["trace"]])
    assert got.shape == log.shape
    assert np.array_equal(got[:, 0], log[:, 0]), "accept/reject sequence differs"
    assert np.allclose(got[:, 1], log[:, 1], rtol=1e-12, atol=0)
    assert np.allclose(got[:, 2], log[:, 2], rtol=1e-10, atol=0)
    assert relerr(p.positions, g["positions_final"]) < 1e-9
    assert abs(res["energy"] - g["E_final"]) <= 1e-10 * abs(g["E_final"])
    assert abs(res["step_size"] - g["step_size_final"]) <= 1e-12 * g["step_size_final"]
    if "tilts_final" in g:
        assert relerr(p.tilts, g["tilts_final"]) < 1e-9


# ---- bending_tilt + nested tilt relaxation (single tilt field) ----------------------
@pytest.mark.parametrize("name", ["ico5", "disk5"])
@pytest.mark.parametrize("mode", ["analytic", "approx"])
def test_bending_tilt_matches_reference(name, mode):
    """modules/energy/bending_tilt.py on a closed and an open (boundary, obtuse) mesh."""
    g = load_golden("bending_tilt_cases.npz")
    pos, tri, isb, tl = g[name + "_positions"], g[name + "_tri"], g[name + "_is_boundary"], g[name + "_tilts"]
    nv = pos.shape[0]
    grad, tg = np.zeros_like(pos), np.zeros_like(pos)
    E = orc.bending_tilt_energy_and_gradient(pos, tl, tri, np.full(nv, 1.3), np.full(nv, 0.2), isb,
                                             mode=mode, grad=grad, tilt_grad=tg)
    k = f"{name}_{mode}"
    assert abs(E - g[k + "_E"]) <= 1e-12 * abs(g[k + "_E"])
    assert relerr(grad, g[k + "_grad"]) < 1e-11
    assert relerr(tg, g[k + "_tilt_grad"]) < 1e-11
    tg2 = np.zeros_like(pos)
    E2 = orc.bending_tilt_energy_and_gradient(pos, tl, tri, np.full(nv, 1.3), np.full(nv, 0.2), isb,
                                              tilt_grad=tg2)
    assert abs(E2 - g[k + "_E_tiltonly"]) <= 1e-12 * abs(E2)
    assert relerr(tg2, g[k + "_tilt_grad_tiltonly"]) < 1e-11
    p = mp.Problem(positions=pos, tri=tri, is_boundary=isb, tilts=tl, gp={"tilt_rigidity": 2.0})
    assert relerr(mp.tilt_cg_preconditioner(p, pos, np.zeros(nv, bool)), g[name + "_jacobi_Minv"]) < 1e-12


BT_BASE = {"surface_tension": 1.0, "bending_modulus": 1.0, "spontaneous_curvature": 0.1,
           "bending_energy_model": "helfrich", "bending_gradient_mode": "analytic", "tilt_rigidity": 2.5,
           "volume_constraint_mode": "lagrange", "volume_projection_during_minimization": False}
BT_TRAJ = {
    "traj_ico4_gd_bt_fixed.npz": ("gd", dict(BT_BASE, tilt_solve_mode="fixed")),
    "traj_ico4_gd_bt_nested_cg.npz": ("gd", dict(BT_BASE, tilt_solve_mode="nested", tilt_solver="cg",
                                                 tilt_step_size=0.1, tilt_inner_steps=6, tilt_tol=1e-10)),
    "traj_ico4_cg_bt_nested_gd.npz": ("cg", dict(BT_BASE, tilt_solve_mode="nested", tilt_solver="gd",
                                                 tilt_step_size=0.08, tilt_inner_steps=4)),
    "traj_disk5_gd_bt_coupled.npz": ("gd", dict(BT_BASE, tilt_solve_mode="coupled", tilt_solver="cg",
                                                tilt_cg_preconditioner="none", tilt_step_size=0.1,
                                                tilt_coupled_steps=3, bending_modulus=0.8)),
}


@pytest.mark.parametrize("fname", sorted(BT_TRAJ))
def test_port_reproduces_bending_tilt_trajectory(fname):
    """surface + tilt + bending_tilt with the tilt field fixed / relaxed by the nested GD or
    (Jacobi-preconditioned) CG inner solve of runtime/steppers/tilt_relaxation.py:237-424."""
    kind, gp = BT_TRAJ[fname]
    g = load_golden(fname)
    p = mp.Problem(positions=g["positions0"], tri=g["tri"], gamma=g["gamma"], is_boundary=g["is_boundary"],
                   fixed=g["fixed"], tilts=g["tilts0"], tilt_fixed=g["tilt_fixed"],
                   energy_modules=["surface", "tilt", "bending_tilt"], constraint_modules=[], gp=dict(gp))
    E0, grad0 = mp.energy_and_gradient(p, p.positions)
    assert abs(E0 - g["E0"]) <= 1e-12 * abs(g["E0"])
    assert relerr(grad0, g["grad0"]) < 1e-11
    stepper = mp.GradientDescent() if kind == "gd" else mp.ConjugateGradient()
    res = mp.minimize(p, stepper, int(g["n_steps"]), step_size=float(g["step_size0"]))
    log = g["step_log"]
    got = np.array([[float(t["success"]), t["next_step"], t["E_accepted"]] for t in res["trace"]])
    assert got.shape == log.shape
    assert np.array_equal(got[:, 0], log[:, 0])
    assert np.allclose(got[:, 1], log[:, 1], rtol=1e-12, atol=0)
    assert np.allclose(got[:, 2], log[:, 2], rtol=1e-9, atol=0)
    assert relerr(p.positions, g["positions_final"]) < 1e-8
    assert relerr(p.tilts, g["tilts_final"]) < 1e-8
    assert abs(res["energy"] - g["E_final"]) <= 1e-9 * abs(g["E_final"])


# ---- tilt_smoothness (cotangent Dirichlet energy of the tilt field) ---------------------
@pytest.mark.parametrize("name", ["ico5", "disk5"])
def test_tilt_smoothness_matches_reference(name):
    g = load_golden("tilt_smoothness_cases.npz")
    pos, tri, tl = g[name + "_positions"], g[name + "_tri"], g[name + "_tilts"]
    tg = np.zeros_like(pos)
    E = orc.tilt_smoothness_energy_and_gradient(pos, tl, tri, 0.7, tg)
    assert abs(E - g[name + "_E"]) <= 1e-12 * abs(g[name + "_E"])
    assert relerr(tg, g[name + "_tilt_grad"]) < 1e-11
    assert not np.any(g[name + "_grad"])  # the reference module has no shape gradient
    p = mp.Problem(positions=pos, tri=tri, is_boundary=g[name + "_is_boundary"], tilts=tl,
                   gp={"tilt_rigidity": 2.0, "tilt_smoothness_rigidity": 0.7})
    assert relerr(mp.tilt_cg_preconditioner(p, pos, np.zeros(len(pos), bool)), g[name + "_jacobi_Minv"]) < 1e-12


TS_BASE = dict(BT_BASE, tilt_smoothness_rigidity=0.6)
TS_TRAJ = {
    "traj_ico4_gd_ts_nested_cg.npz": ("gd", ["surface", "tilt", "tilt_smoothness", "bending_tilt"],
                                      dict(TS_BASE, tilt_solve_mode="nested", tilt_solver="cg",
                                           tilt_step_size=0.1, tilt_inner_steps=6)),
    "traj_ico4_cg_ts_fixed.npz": ("cg", ["surface", "tilt", "tilt_smoothness"],
                                  dict(TS_BASE, tilt_solve_mode="fixed")),
    # rejected trials: the CG stepper's mesh-mutating line search compounds the tilt projections,
    # the GD stepper's array line search does not
    "traj_ico4_cg_ts_backtrack.npz": ("cg", ["surface", "tilt", "tilt_smoothness", "bending_tilt"],
                                      dict(TS_BASE, tilt_solve_mode="fixed")),
    "traj_ico4_gd_ts_backtrack.npz": ("gd", ["surface", "tilt", "tilt_smoothness", "bending_tilt"],
                                      dict(TS_BASE, tilt_solve_mode="fixed")),
}


@pytest.mark.parametrize("fname", sorted(TS_TRAJ))
def test_port_reproduces_tilt_smoothness_trajectory(fname):
    kind, mods, gp = TS_TRAJ[fname]
    g = load_golden(fname)
    p = mp.Problem(positions=g["positions0"], tri=g["tri"], gamma=g["gamma"], is_boundary=g["is_boundary"],
                   fixed=g["fixed"], tilts=g["tilts0"], tilt_fixed=g["tilt_fixed"], energy_modules=list(mods),
                   constraint_modules=[], gp=dict(gp))
    E0, grad0 = mp.energy_and_gradient(p, p.positions)
    assert abs(E0 - g["E0"]) <= 1e-12 * abs(g["E0"])
    assert relerr(grad0, g["grad0"]) < 1e-11
    stepper = mp.GradientDescent() if kind == "gd" else mp.ConjugateGradient()
    res = mp.minimize(p, stepper, int(g["n_steps"]), step_size=float(g["step_size0"]))
    log = g["step_log"]
    got = np.array([[float(t["success"]), t["next_step"], t["E_accepted"]] for t in res["trace"]])
    assert got.shape == log.shape
    assert np.array_equal(got[:, 0], log[:, 0])
    assert np.allclose(got[:, 1], log[:, 1], rtol=1e-12, atol=0)
    assert np.allclose(got[:, 2], log[:, 2], rtol=1e-9, atol=0)
    assert relerr(p.positions, g["positions_final"]) < 1e-8
    assert relerr(p.tilts, g["tilts_final"]) < 1e-8


# ---------------------------------------------------------------------------
# two-leaflet tilt fields: tilt_in / tilt_out (lumped + consistent), tilt_smoothness_in / _out,
# leaflet Jacobi preconditioner, relax_leaflet_tilts (oracle/gen_golden.py: gen_leaflet)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["ico5", "disk5"])
@pytest.mark.parametrize("mass", ["lumped", "consistent"])
def test_leaflet_modules_match_reference(name, mass):
    import json

    g = load_golden("tilt_leaflet_cases.npz")
    key = f"{name}_{mass}"
    gp = json.loads(str(g[key + "_gp_json"]))
    pos, tri = g[name + "_positions"], g[name + "_tri"]
    tin, tout = g[name + "_tilts_in"], g[name + "_tilts_out"]
    p = mp.Problem(positions=pos, tri=tri, is_boundary=g[name + "_is_boundary"], tilts_in=tin, tilts_out=tout,
                   energy_modules=[], gp=gp)
    for lf, tl in (("in", tin), ("out", tout)):
        grad, tg = np.zeros_like(pos), np.zeros_like(pos)
        E = orc.tilt_leaflet_energy_and_gradient(pos, tl, tri, mp.tilt_modulus(p, lf), mp.tilt_mass_mode(p, lf), grad, tg)
        assert abs(E - g[f"{key}_tilt_{lf}_E"]) <= 1e-13 * abs(g[f"{key}_tilt_{lf}_E"])
        assert relerr(grad, g[f"{key}_tilt_{lf}_grad"]) < 1e-12
        assert relerr(tg, g[f"{key}_tilt_{lf}_tilt_grad"]) < 1e-12
        tg = np.zeros_like(pos)
        E = mp._smoothness_leaflet(p, pos, tl, lf, tg)
        assert abs(E - g[f"{key}_tilt_smoothness_{lf}_E"]) <= 1e-12 * abs(g[f"{key}_tilt_smoothness_{lf}_E"])
        assert relerr(tg, g[f"{key}_tilt_smoothness_{lf}_tilt_grad"]) < 1e-11
        assert not np.any(g[f"{key}_tilt_smoothness_{lf}_grad"])  # no shape gradient
    # the mass mode really is per leaflet: "in" follows tilt_mass_mode_in, "out" the global key
    assert mp.tilt_mass_mode(p, "in") == mass and mp.tilt_mass_mode(p, "out") != mass
    va = orc.barycentric_vertex_areas(pos, tri)
    fin = g[key + "_jacobi_fixed_in"]
    Mi, Mo = mp.leaflet_tilt_cg_preconditioner(p, pos, fin, np.zeros(len(pos), bool), va)
    assert relerr(Mi, g[key + "_jacobi_Minv_in"]) < 1e-11 and relerr(Mo, g[key + "_jacobi_Minv_out"]) < 1e-11
    p.energy_modules = ["tilt_in", "tilt_out", "tilt_smoothness_in", "tilt_smoothness_out"]
    E, gi, go = mp.energy_and_leaflet_tilt_gradients(p, pos, tin, tout, va)
    assert abs(E - g[key + "_relax_E"]) <= 1e-12 * abs(g[key + "_relax_E"])
    assert relerr(gi, g[key + "_relax_grad_in"]) < 1e-11 and relerr(go, g[key + "_relax_grad_out"]) < 1e-11


LEAFLET_TRAJ = {
    "traj_ico4_gd_leaflet_nested_cg.npz": "gd",
    "traj_ico4_cg_leaflet_coupled_gd.npz": "cg",
    "traj_ico4_cg_leaflet_plaincg.npz": "cg",
    "traj_disk5_gd_leaflet_consistent_backtrack.npz": "gd",
}


def leaflet_problem(g):
    import json

    return mp.Problem(positions=g["positions0"], tri=g["tri"], gamma=g["gamma"], is_boundary=g["is_boundary"],
                      fixed=g["fixed"], tilts_in=g["tilts_in0"], tilts_out=g["tilts_out0"],
                      tilt_fixed_in=g["tilt_fixed_in"], tilt_fixed_out=g["tilt_fixed_out"],
                      energy_modules=[str(m) for m in g["modules"]], constraint_modules=[],
                      gp=json.loads(str(g["gp_json"])))


@pytest.mark.parametrize("fname", sorted(LEAFLET_TRAJ))
def test_port_reproduces_leaflet_trajectory(fname):
    g = load_golden(fname)
    p = leaflet_problem(g)
    E0, grad0 = mp.energy_and_gradient(p, p.positions)
    assert abs(E0 - g["E0"]) <= 1e-12 * abs(g["E0"])
    assert relerr(grad0, g["grad0"]) < 1e-11
    stepper = mp.GradientDescent() if LEAFLET_TRAJ[fname] == "gd" else mp.ConjugateGradient()
    res = mp.minimize(p, stepper, int(g["n_steps"]), step_size=float(g["step_size0"]))
    log = g["step_log"]
    got = np.array([[float(t["success"]), t["next_step"], t["E_accepted"]] for t in res["trace"]])
    assert got.shape == log.shape
    assert np.array_equal(got[:, 0], log[:, 0])
    assert np.allclose(got[:, 1], log[:, 1], rtol=1e-12, atol=0)
    assert np.allclose(got[:, 2], log[:, 2], rtol=1e-9, atol=0)
    assert relerr(p.positions, g["positions_final"]) < 1e-8
    assert relerr(p.tilts_in, g["tilts_in_final"]) < 1e-8
    assert relerr(p.tilts_out, g["tilts_out_final"]) < 1e-8


# ---------------------------------------------------------------------------
# bending_tilt_in / bending_tilt_out (oracle/gen_golden.py: gen_bending_tilt_leaflet)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["ico5", "disk5"])
def test_bending_tilt_leaflet_matches_reference(name):
    import json

    g = load_golden("bending_tilt_leaflet_cases.npz")
    gp = json.loads(str(g["gp_json"]))
    pos, tri = g[name + "_positions"], g[name + "_tri"]
    p = mp.Problem(positions=pos, tri=tri, is_boundary=g[name + "_is_boundary"], tilts_in=g[name + "_tilts_in"],
                   tilts_out=g[name + "_tilts_out"], energy_modules=[], gp=gp)
    for lf in ("in", "out"):
        grad, tg = np.zeros_like(pos), np.zeros_like(pos)
        E = mp._bending_tilt_leaflet(p, pos, mp.leaflet_tilts(p, lf), lf, grad=grad, tilt_grad=tg)
        ref = g[f"{name}_bending_tilt_{lf}_E"]
        assert abs(E - ref) <= 1e-12 * abs(ref)
        assert relerr(grad, g[f"{name}_bending_tilt_{lf}_grad"]) < 1e-11
        assert relerr(tg, g[f"{name}_bending_tilt_{lf}_tilt_grad"]) < 1e-11
        assert abs(mp._bending_tilt_leaflet(p, pos, mp.leaflet_tilts(p, lf), lf) - ref) <= 1e-12 * abs(ref)


BTL_TRAJ = {"traj_ico4_gd_btl_nested_cg.npz": "gd", "traj_ico4_cg_btl_coupled_gd.npz": "cg",
            "traj_disk5_gd_btl_backtrack.npz": "gd"}


@pytest.mark.parametrize("fname", sorted(BTL_TRAJ))
def test_port_reproduces_bending_tilt_leaflet_trajectory(fname):
    g = load_golden(fname)
    p = leaflet_problem(g)
    E0, grad0 = mp.energy_and_gradient(p, p.positions)
    assert abs(E0 - g["E0"]) <= 1e-12 * abs(g["E0"])
    assert relerr(grad0, g["grad0"]) < 1e-11
    stepper = mp.GradientDescent() if BTL_TRAJ[fname] == "gd" else mp.ConjugateGradient()
    res = mp.minimize(p, stepper, int(g["n_steps"]), step_size=float(g["step_size0"]))
    log = g["step_log"]
    got = np.array([[float(t["success"]), t["next_step"], t["E_accepted"]] for t in res["trace"]])
    assert got.shape == log.shape
    assert np.array_equal(got[:, 0], log[:, 0])
    assert np.allclose(got[:, 1], log[:, 1], rtol=1e-12, atol=0)
    assert np.allclose(got[:, 2], log[:, 2], rtol=1e-9, atol=0)
    assert relerr(p.positions, g["positions_final"]) < 1e-8
    assert relerr(p.tilts_in, g["tilts_in_final"]) < 1e-8
    assert relerr(p.tilts_out, g["tilts_out_final"]) < 1e-8


# ---------------------------------------------------------------------------
# tilt_disk_target_in / _out (oracle/gen_golden.py: gen_disk_target)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["disk6", "ico5"])
@pytest.mark.parametrize("tag", ["bessel", "linear", "moduli"])
def test_disk_target_matches_reference(name, tag):
    import json

    g = load_golden("tilt_disk_target_cases.npz")
    key = f"{name}_{tag}"
    gp = json.loads(str(g[key + "_gp_json"]))
    pos, tri, rows = g[name + "_positions"], g[name + "_tri"], g[name + "_disk_rows"]
    p = mp.Problem(positions=pos, tri=tri, is_boundary=g[name + "_is_boundary"], tilts_in=g[name + "_tilts_in"],
                   tilts_out=g[name + "_tilts_out"], disk_rows_in=rows, disk_rows_out=rows, energy_modules=[], gp=gp)
    for lf in ("in", "out"):
        grad, tg = np.zeros_like(pos), np.zeros_like(pos)
        E = mp._disk_target(p, pos, mp.leaflet_tilts(p, lf), lf, grad=grad, tilt_grad=tg)
        ref = g[f"{key}_tilt_disk_target_{lf}_E"]
        assert abs(E - ref) <= 1e-12 * abs(ref)
        assert relerr(grad, g[f"{key}_tilt_disk_target_{lf}_grad"]) < 1e-11
        assert relerr(tg, g[f"{key}_tilt_disk_target_{lf}_tilt_grad"]) < 1e-11


DISK_TRAJ = {"traj_disk6_gd_disktarget_nested_cg.npz": "gd", "traj_disk6_cg_disktarget_coupled_gd.npz": "cg"}


@pytest.mark.parametrize("fname", sorted(DISK_TRAJ))
def test_port_reproduces_disk_target_trajectory(fname):
    g = load_golden(fname)
    p = leaflet_problem(g)
    p.disk_rows_in = p.disk_rows_out = g["disk_rows"]
    E0, grad0 = mp.energy_and_gradient(p, p.positions)
    assert abs(E0 - g["E0"]) <= 1e-12 * abs(g["E0"])
    assert relerr(grad0, g["grad0"]) < 1e-11
    stepper = mp.GradientDescent() if DISK_TRAJ[fname] == "gd" else mp.ConjugateGradient()
    res = mp.minimize(p, stepper, int(g["n_steps"]), step_size=float(g["step_size0"]))
    log = g["step_log"]
    got = np.array([[float(t["success"]), t["next_step"], t["E_accepted"]] for t in res["trace"]])
    assert got.shape == log.shape
    assert np.array_equal(got[:, 0], log[:, 0])
    assert np.allclose(got[:, 1], log[:, 1], rtol=1e-12, atol=0)
    assert np.allclose(got[:, 2], log[:, 2], rtol=1e-9, atol=0)
    assert relerr(p.positions, g["positions_final"]) < 1e-8
    assert relerr(p.tilts_in, g["tilts_in_final"]) < 1e-8
    assert relerr(p.tilts_out, g["tilts_out_final"]) < 1e-8


def test_curvature_fields_and_open_surface_gauss_bonnet_match_reference():
    """compute_curvature_fields (geometry/curvature.py:404-448) on a closed noisy sphere and an open bulged disk, and
    the Gaussian-modulus energy of the surface WITH boundary (gaussian_curvature.py:128-143)."""
    g = load_golden("angle_defect_cases.npz")
    for name in ("ico5", "disk5"):
        cf = orc.curvature_fields(g[name + "_positions"], g[name + "_tri"], g[name + "_is_boundary"])
        for key, tol in (("mean_curvature_normal", 1e-12), ("mean_curvature", 1e-12), ("mixed_area", 1e-13),
                         ("angle_defect", 1e-12), ("gaussian_curvature", 1e-11), ("principal_curvatures", 1e-11)):
            assert relerr(cf[key], g[f"{name}_cf_{key}"]) < tol, (name, key)
    G, k_int, b_tot = orc.gauss_bonnet_invariant(g["disk5_positions"], g["disk5_tri"], g["disk5_is_boundary"])
    assert abs(G - g["disk5_gauss_bonnet_G"]) < 1e-12 and abs(G - 2.0 * np.pi) < 1e-12  # a disk: chi = 1
    assert abs(k_int - g["disk5_gauss_bonnet_interior"]) < 1e-12 and abs(b_tot - g["disk5_gauss_bonnet_boundary"]) < 1e-12
    assert abs(-0.7 * G - g["disk5_gaussian_E"]) < 1e-12


def test_angle_defects_match_reference():
    g = load_golden("angle_defect_cases.npz")
    for name in ("ico5", "disk5"):
        d = orc.angle_defects(g[name + "_positions"], g[name + "_tri"], g[name + "_is_boundary"])
        assert np.max(np.abs(d - g[name + "_defects"])) <= 1e-14
    chi = orc.euler_characteristic(len(g["ico5_positions"]), g["ico5_tri"])
    assert chi == 2 and abs(np.sum(g["ico5_defects"]) - 2.0 * np.pi * chi) < 1e-10  # Gauss-Bonnet
    assert abs(float(g["ico5_gaussian_E"]) - (-0.7) * 2.0 * np.pi * chi) <= 1e-14


# ---------------------------------------------------------------------------
# BASELINE config 5 on its own deck (oracle/gen_golden.py: gen_config5): the caveolin one-disk bilayer deck's mesh,
# flags, parameters and energy-module list (its pin / rim constraints are out of scope and were not kept)
# ---------------------------------------------------------------------------
def test_port_reproduces_config5_deck():
    g = load_golden("traj_config5_deck_gd.npz")
    assert len(g["positions0"]) == 109 and len(g["tri"]) == 204 and int(g["is_boundary"].sum()) == 12
    p = leaflet_problem(g)
    p.disk_rows_in = p.disk_rows_out = g["disk_rows"]
    # the deck's own state: zero tilt fields, the disk-target modules carry all the energy
    E0, grad0 = mp.energy_and_gradient(p, p.positions)
    assert abs(E0 - g["E0"]) <= 1e-12 * abs(g["E0"])
    assert relerr(grad0, g["grad0"]) < 1e-11
    # every module's share with seeded tangent fields: sum of the reference's per-module energies and gradients
    pb = leaflet_problem(g)
    pb.disk_rows_in = pb.disk_rows_out = g["disk_rows"]
    pb.tilts_in, pb.tilts_out = g["state_b_tilts_in"].copy(), g["state_b_tilts_out"].copy()
    Eb, gb = mp.energy_and_gradient(pb, pb.positions)
    mods = [str(m) for m in g["modules"]]
    assert abs(Eb - sum(float(g[f"mod_{m}_E"]) for m in mods)) <= 1e-12 * abs(Eb)
    assert abs(Eb - g["state_b_E"]) <= 1e-12 * abs(Eb)
    fixed = g["fixed"].astype(bool)

    def movable(a):  # Minimizer.compute_energy_and_gradient_array zeroes the fixed rows (minimizer.py:988-990)
        a = np.array(a, copy=True)
        a[fixed] = 0.0
        return a

    assert relerr(gb, movable(sum(g[f"mod_{m}_grad"] for m in mods))) < 1e-11
    assert relerr(gb, g["state_b_grad"]) < 1e-11
    for m in mods:
        pm = leaflet_problem(g)
        pm.disk_rows_in = pm.disk_rows_out = g["disk_rows"]
        pm.tilts_in, pm.tilts_out = g["state_b_tilts_in"].copy(), g["state_b_tilts_out"].copy()
        pm.energy_modules = [m]
        Em, gm = mp.energy_and_gradient(pm, pm.positions)
        ref = float(g[f"mod_{m}_E"])
        assert abs(Em - ref) <= 1e-12 * max(abs(ref), 1e-300), m
        if np.any(g[f"mod_{m}_grad"]):
            assert relerr(gm, movable(g[f"mod_{m}_grad"])) < 1e-11, m
        else:
            assert not np.any(gm), m
    # ONE relax_leaflet_tilts call as the deck configures it (coupled, jacobi CG, 40 inner steps, step 0.15)
    pr = leaflet_problem(g)
    pr.disk_rows_in = pr.disk_rows_out = g["disk_rows"]
    mp.relax_leaflet_tilts(pr, pr.positions)
    assert relerr(pr.tilts_in, g["relax_tilts_in"]) < 1e-9
    assert relerr(pr.tilts_out, g["relax_tilts_out"]) < 1e-9
    assert abs(mp.energy_total(pr, pr.positions) - g["relax_E"]) <= 1e-10 * abs(g["relax_E"])
    # the deck's `g` steps: fixed step size 0.01, coupled tilt relaxation at the top of every iteration
    res = mp.minimize(p, mp.GradientDescent(), int(g["n_steps"]), step_size=float(g["step_size0"]))
    log = g["step_log"]
    got = np.array([[float(t["success"]), t["next_step"], t["E_accepted"]] for t in res["trace"]])
    assert got.shape == log.shape
    assert np.array_equal(got[:, 0], log[:, 0])
    assert np.allclose(got[:, 1], log[:, 1], rtol=1e-12, atol=0)
    assert np.allclose(got[:, 2], log[:, 2], rtol=1e-9, atol=0)
    assert relerr(p.positions, g["positions_final"]) < 1e-8
    assert relerr(p.tilts_in, g["tilts_in_final"]) < 1e-8
    assert relerr(p.tilts_out, g["tilts_out_final"]) < 1e-8
    assert abs(res["energy"] - g["E_final"]) <= 1e-9 * abs(g["E_final"])
