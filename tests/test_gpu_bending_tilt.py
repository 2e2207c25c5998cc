"""GPU parity of the bending_tilt module (modules/energy/bending_tilt.py) and of the nested /
coupled tilt relaxation (runtime/steppers/tilt_relaxation.py:237-424) against the reference's
golden vectors and the CPU oracle.  Tolerances: energies 1e-12, gradients 1e-10 relative;
trajectories are compared per step (accept sequence identical, energies 1e-9, final state 1e-8)."""

import numpy as np
import pytest

from conftest import load_golden, relerr

pytestmark = pytest.mark.gpu


def _dm(pos, tri, isb, tl, mode, tile, k_tilt=2.0, kappa=1.3, c0=0.2, tilt_module=False):
    from membrane_solver_amd import _lib as L
    from membrane_solver_amd.device import DeviceMesh

    nv = pos.shape[0]
    dm = DeviceMesh(pos, tri, boundary=isb, tile_vertices=tile)
    dm.set_surface_tension(np.ones(tri.shape[0]))
    dm.set_bending_params(np.full(nv, kappa), np.full(nv, c0))
    dm.set_tilts(tl, k_tilt)
    mods = L.MS_MOD_BENDING_TILT | (L.MS_MOD_TILT if tilt_module else 0)
    dm.set_params(modules=mods, bending_grad_mode=L.MS_GRAD_ANALYTIC if mode == "analytic" else L.MS_GRAD_APPROX)
    return dm


@pytest.mark.parametrize("tile", [64, 256])
@pytest.mark.parametrize("mode", ["analytic", "approx"])
@pytest.mark.parametrize("name", ["ico5", "disk5"])
def test_bending_tilt_kernel_cases(name, mode, tile):
    g = load_golden("bending_tilt_cases.npz")
    pos, tri, isb, tl = g[name + "_positions"], g[name + "_tri"], g[name + "_is_boundary"], g[name + "_tilts"]
    dm = _dm(pos, tri, isb, tl, mode, tile)
    k = f"{name}_{mode}"
    e, grad = dm.energy_and_gradient()
    assert abs(e[1] - g[k + "_E"]) <= 1e-12 * abs(g[k + "_E"])
    assert relerr(grad, g[k + "_grad"]) < 1e-10
    E2, tg = dm.tilt_energy_and_gradient()
    assert abs(E2 - g[k + "_E"]) <= 1e-12 * abs(g[k + "_E"])
    assert relerr(tg, g[k + "_tilt_grad"]) < 1e-10
    assert abs(dm.energy()[1] - g[k + "_E"]) <= 1e-12 * abs(g[k + "_E"])
    dm.close()


def test_bending_tilt_plugin_accumulates():
    from membrane_solver_amd.core.parameters import GlobalParameters, ParameterResolver
    from membrane_solver_amd.geometry.mesh import ArrayMesh
    from membrane_solver_amd.modules.energy import bending_tilt

    g = load_golden("bending_tilt_cases.npz")
    pos, tri, tl = g["ico5_positions"], g["ico5_tri"], g["ico5_tilts"]
    gp = GlobalParameters({"bending_modulus": 1.3, "spontaneous_curvature": 0.2, "tilt_rigidity": 2.0,
                           "bending_gradient_mode": "analytic"})
    mesh = ArrayMesh(pos, tri, tilts=tl, global_parameters=gp)
    grad = np.full_like(pos, 0.5)
    tg = np.full_like(pos, -0.25)
    E = bending_tilt.compute_energy_and_gradient_array(mesh, gp, ParameterResolver(gp), positions=pos,
                                                       index_map=mesh.vertex_index_to_row, grad_arr=grad,
                                                       tilt_grad_arr=tg)
    assert abs(E - g["ico5_analytic_E"]) <= 1e-12 * abs(E)
    assert relerr(grad - 0.5, g["ico5_analytic_grad"]) < 1e-10
    assert relerr(tg + 0.25, g["ico5_analytic_tilt_grad"]) < 1e-10
    tg2 = np.zeros_like(pos)
    E2 = bending_tilt.compute_energy_and_gradient_array(mesh, gp, ParameterResolver(gp), positions=pos,
                                                        index_map=mesh.vertex_index_to_row, grad_arr=None,
                                                        tilt_grad_arr=tg2)
    assert abs(E2 - g["ico5_analytic_E_tiltonly"]) <= 1e-12 * abs(E2)
    assert relerr(tg2, g["ico5_analytic_tilt_grad_tiltonly"]) < 1e-10


@pytest.mark.parametrize("solver,jacobi,pin", [("cg", True, False), ("cg", False, True), ("gd", True, True)])
def test_relax_tilts_matches_oracle(solver, jacobi, pin):
    """ms_relax_tilts vs oracle/minimizer_port.relax_tilts on frozen positions (tilt + bending_tilt)."""
    from oracle import minimizer_port as mp

    g = load_golden("bending_tilt_cases.npz")
    pos, tri, isb, tl = g["disk5_positions"], g["disk5_tri"], g["disk5_is_boundary"], g["disk5_tilts"]
    nv = pos.shape[0]
    tfix = np.zeros(nv, bool)
    if pin:
        tfix[::7] = True
    gp = {"bending_modulus": 1.3, "spontaneous_curvature": 0.2, "tilt_rigidity": 2.0, "tilt_solve_mode": "nested",
          "tilt_solver": solver, "tilt_step_size": 0.12, "tilt_inner_steps": 7,
          "tilt_cg_preconditioner": "jacobi" if jacobi else "none"}
    p = mp.Problem(positions=pos, tri=tri, is_boundary=isb, tilts=tl, tilt_fixed=tfix,
                   energy_modules=["tilt", "bending_tilt"], gp=gp)
    st = mp.relax_tilts(p, p.positions)
    dm = _dm(pos, tri, isb, tl, "analytic", 64, tilt_module=True)
    dm.set_tilt_fixed(tfix)
    iters, evals = dm.relax_tilts(solver=solver, max_iters=7, step_size=0.12, jacobi=jacobi)
    assert (iters, evals) == (st["iters"], st["evals"])
    assert relerr(dm.get_tilts(), p.tilts) < 1e-10
    E_dev, _ = dm.tilt_energy_and_gradient(want_gradient=False)
    assert abs(E_dev - mp.tilt_dependent_energy(p, pos, p.tilts)) <= 1e-11 * abs(E_dev)
    dm.close()


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
def test_minimizer_reproduces_bending_tilt_trajectory(fname):
    from membrane_solver_amd.geometry.mesh import ArrayMesh
    from membrane_solver_amd.runtime.constraint_manager import ConstraintModuleManager
    from membrane_solver_amd.runtime.energy_manager import EnergyModuleManager
    from membrane_solver_amd.runtime.minimizer import Minimizer
    from membrane_solver_amd.runtime.steppers import ConjugateGradient, GradientDescent

    kind, gp = BT_TRAJ[fname]
    g = load_golden(fname)
    mods = ["surface", "tilt", "bending_tilt"]
    mesh = ArrayMesh(g["positions0"], g["tri"], fixed=g["fixed"], surface_tension=g["gamma"], tilts=g["tilts0"],
                     tilt_fixed=g["tilt_fixed"], global_parameters=dict(gp), energy_modules=mods,
                     constraint_modules=[])
    stepper = GradientDescent() if kind == "gd" else ConjugateGradient()
    log = []
    orig = stepper.device_step

    def logged(dm, m, step_size, tol=0.0):
        r = orig(dm, m, step_size, tol=tol)
        log.append((float(r.success), r.next_step, r.energy))
        return r

    stepper.device_step = logged
    mz = Minimizer(mesh, mesh.global_parameters, stepper, EnergyModuleManager(mods), ConstraintModuleManager([]),
                   quiet=True, step_size=float(g["step_size0"]))
    E0, grad0 = mz.compute_energy_and_gradient_array()
    assert abs(E0 - g["E0"]) <= 1e-12 * abs(g["E0"])
    assert relerr(grad0, g["grad0"]) < 1e-10
    res = mz.minimize(int(g["n_steps"]))
    got, ref = np.array(log), g["step_log"]
    assert got.shape == ref.shape
    assert np.array_equal(got[:, 0], ref[:, 0]), "accept/reject sequence differs from the reference"
    assert np.allclose(got[:, 1], ref[:, 1], rtol=1e-12, atol=0)
    assert np.allclose(got[:, 2], ref[:, 2], rtol=1e-9, atol=0)
    assert relerr(mesh.positions_view(), g["positions_final"]) < 1e-8
    assert relerr(mesh.tilts_view(), g["tilts_final"]) < 1e-8
    assert abs(res["energy"] - g["E_final"]) <= 1e-9 * abs(g["E_final"])


def test_bending_tilt_midsize_matches_oracle():
    """131 220 facets (config C2 size), many tiles: HIP vs the CPU oracle for the energy, the shape
    gradient and the tilt gradient of tilt + bending_tilt; plus size-independent properties
    (quadratic scaling in the tilt amplitude of the pure tilt part, translation invariance)."""
    from membrane_solver_amd import _lib as L
    from membrane_solver_amd import meshgen
    from membrane_solver_amd.device import DeviceMesh
    from oracle import minimizer_port as mp
    from oracle import ms_oracle as orc

    P, T = meshgen.icosphere(81)
    P = meshgen.smooth_displace(P, 0.05)
    nv, nf = P.shape[0], T.shape[0]
    rng = np.random.default_rng(11)
    nrm = mp.unit_vertex_normals(P, T)
    tl = 0.02 * rng.normal(size=P.shape)
    tl -= np.einsum("ij,ij->i", tl, nrm)[:, None] * nrm
    kappa, c0 = np.full(nv, 1.1), np.full(nv, 0.3)
    dm = DeviceMesh(P, T)
    dm.set_surface_tension(np.ones(nf))
    dm.set_bending_params(kappa, c0)
    dm.set_tilts(tl, 1.7)
    dm.set_params(modules=L.MS_MOD_BENDING_TILT | L.MS_MOD_TILT)
    e, grad = dm.energy_and_gradient()
    E_t, tg = dm.tilt_energy_and_gradient()
    g_ref, tg_ref = np.zeros_like(P), np.zeros_like(P)
    isb = np.zeros(nv, bool)
    E_bt = orc.bending_tilt_energy_and_gradient(P, tl, T, kappa, c0, isb, grad=g_ref, tilt_grad=tg_ref)
    E_tilt = orc.tilt_energy_and_gradient(P, tl, T, 1.7, g_ref, tg_ref)
    assert abs(e[1] - E_bt) <= 1e-11 * abs(E_bt)
    assert abs(e[3] - E_tilt) <= 1e-11 * abs(E_tilt)
    assert abs(E_t - (E_bt + E_tilt)) <= 1e-11 * abs(E_t)
    assert relerr(grad, g_ref) < 1e-9  # same conditioning caveat as the bending gradient at this size
    assert relerr(tg, tg_ref) < 1e-10
    # pure tilt energy is quadratic in the amplitude
    dm.set_tilts(2.0 * tl, 1.7)
    assert abs(dm.energy()[3] - 4.0 * e[3]) <= 1e-12 * abs(4.0 * e[3])
    # rigid translation leaves both energies unchanged
    dm.set_tilts(tl, 1.7)
    dm.set_positions(P + np.array([0.3, -0.2, 0.1]))
    e2 = dm.energy()
    assert abs(e2[1] - e[1]) <= 1e-9 * abs(e[1]) and abs(e2[3] - e[3]) <= 1e-11 * abs(e[3])
    dm.close()


# ---- tilt_smoothness ------------------------------------------------------------------------
@pytest.mark.parametrize("tile", [64, 256])
@pytest.mark.parametrize("name", ["ico5", "disk5"])
def test_tilt_smoothness_kernel_cases(name, tile):
    from membrane_solver_amd import _lib as L
    from membrane_solver_amd.device import DeviceMesh

    g = load_golden("tilt_smoothness_cases.npz")
    pos, tri, isb, tl = g[name + "_positions"], g[name + "_tri"], g[name + "_is_boundary"], g[name + "_tilts"]
    dm = DeviceMesh(pos, tri, boundary=isb, tile_vertices=tile)
    dm.set_tilts(tl, 2.0)
    dm.set_tilt_smoothness(0.7)
    dm.set_params(modules=L.MS_MOD_TILT_SMOOTH)
    E, tg = dm.tilt_energy_and_gradient()
    assert abs(E - g[name + "_E"]) <= 1e-12 * abs(g[name + "_E"])
    assert relerr(tg, g[name + "_tilt_grad"]) < 1e-10
    assert abs(dm.energy()[3] - g[name + "_E"]) <= 1e-12 * abs(g[name + "_E"])
    e, grad = dm.energy_and_gradient()
    assert not np.any(grad)  # no shape gradient (tilt_smoothness.py:21-23)
    dm.close()


def test_relax_with_smoothness_matches_oracle():
    """Jacobi diagonal with both terms (k_t A_v + k_s/2 sum(c_a + c_b)) and the three tilt modules."""
    from membrane_solver_amd import _lib as L
    from oracle import minimizer_port as mp

    g = load_golden("tilt_smoothness_cases.npz")
    pos, tri, isb, tl = g["disk5_positions"], g["disk5_tri"], g["disk5_is_boundary"], g["disk5_tilts"]
    nv = pos.shape[0]
    tfix = np.zeros(nv, bool)
    tfix[::11] = True
    gp = {"bending_modulus": 1.3, "spontaneous_curvature": 0.2, "tilt_rigidity": 2.0, "tilt_smoothness_rigidity": 0.7,
          "tilt_solve_mode": "nested", "tilt_solver": "cg", "tilt_step_size": 0.15, "tilt_inner_steps": 8}
    p = mp.Problem(positions=pos, tri=tri, is_boundary=isb, tilts=tl, tilt_fixed=tfix,
                   energy_modules=["tilt", "tilt_smoothness", "bending_tilt"], gp=gp)
    st = mp.relax_tilts(p, p.positions)
    dm = _dm(pos, tri, isb, tl, "analytic", 64, tilt_module=True)
    dm.set_tilt_smoothness(0.7)
    dm.set_params(modules=L.MS_MOD_TILT | L.MS_MOD_BENDING_TILT | L.MS_MOD_TILT_SMOOTH)
    dm.set_tilt_fixed(tfix)
    iters, evals = dm.relax_tilts(solver="cg", max_iters=8, step_size=0.15, jacobi=True)
    assert (iters, evals) == (st["iters"], st["evals"])
    assert relerr(dm.get_tilts(), p.tilts) < 1e-10
    dm.close()


TS_BASE = dict(BT_BASE, tilt_smoothness_rigidity=0.6)
TS_TRAJ = {
    "traj_ico4_gd_ts_nested_cg.npz": ("gd", ["surface", "tilt", "tilt_smoothness", "bending_tilt"],
                                      dict(TS_BASE, tilt_solve_mode="nested", tilt_solver="cg",
                                           tilt_step_size=0.1, tilt_inner_steps=6)),
    "traj_ico4_cg_ts_fixed.npz": ("cg", ["surface", "tilt", "tilt_smoothness"],
                                  dict(TS_BASE, tilt_solve_mode="fixed")),
    # rejected trials: a rejected trial keeps the tilts projected onto its surface (line search's
    # mesh-mutating path, the one the tilt goldens pin)
    "traj_ico4_cg_ts_backtrack.npz": ("cg", ["surface", "tilt", "tilt_smoothness", "bending_tilt"],
                                      dict(TS_BASE, tilt_solve_mode="fixed")),
    "traj_ico4_gd_ts_backtrack.npz": ("gd", ["surface", "tilt", "tilt_smoothness", "bending_tilt"],
                                      dict(TS_BASE, tilt_solve_mode="fixed")),
}


@pytest.mark.parametrize("fname", sorted(TS_TRAJ))
def test_minimizer_reproduces_tilt_smoothness_trajectory(fname):
    from membrane_solver_amd.geometry.mesh import ArrayMesh
    from membrane_solver_amd.runtime.constraint_manager import ConstraintModuleManager
    from membrane_solver_amd.runtime.energy_manager import EnergyModuleManager
    from membrane_solver_amd.runtime.minimizer import Minimizer
    from membrane_solver_amd.runtime.steppers import ConjugateGradient, GradientDescent

    kind, mods, gp = TS_TRAJ[fname]
    g = load_golden(fname)
    mesh = ArrayMesh(g["positions0"], g["tri"], fixed=g["fixed"], surface_tension=g["gamma"], tilts=g["tilts0"],
                     tilt_fixed=g["tilt_fixed"], global_parameters=dict(gp), energy_modules=list(mods),
                     constraint_modules=[])
    stepper = GradientDescent() if kind == "gd" else ConjugateGradient()
    log = []
    orig = stepper.device_step

    def logged(dm, m, step_size, tol=0.0):
        r = orig(dm, m, step_size, tol=tol)
        log.append((float(r.success), r.next_step, r.energy))
        return r

    stepper.device_step = logged
    mz = Minimizer(mesh, mesh.global_parameters, stepper, EnergyModuleManager(mods), ConstraintModuleManager([]),
                   quiet=True, step_size=float(g["step_size0"]))
    E0, grad0 = mz.compute_energy_and_gradient_array()
    assert abs(E0 - g["E0"]) <= 1e-12 * abs(g["E0"])
    assert relerr(grad0, g["grad0"]) < 1e-10
    res = mz.minimize(int(g["n_steps"]))
    got, ref = np.array(log), g["step_log"]
    assert got.shape == ref.shape
    assert np.array_equal(got[:, 0], ref[:, 0]), "accept/reject sequence differs from the reference"
    assert np.allclose(got[:, 1], ref[:, 1], rtol=1e-12, atol=0)
    assert np.allclose(got[:, 2], ref[:, 2], rtol=1e-9, atol=0)
    assert relerr(mesh.positions_view(), g["positions_final"]) < 1e-8
    assert relerr(mesh.tilts_view(), g["tilts_final"]) < 1e-8
    assert abs(res["energy"] - g["E_final"]) <= 1e-9 * abs(g["E_final"])
    bd = mz.compute_energy_breakdown()
    assert abs(sum(bd.values()) - g["E_final"]) <= 1e-9 * abs(g["E_final"])
