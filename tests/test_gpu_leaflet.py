"""Two-leaflet tilt fields on the HIP path: tilt_in / tilt_out (lumped + consistent mass), tilt_smoothness_in /
_out, the leaflet relaxation (relax_leaflet_tilts) and the minimizer loop around them, against the reference's
golden vectors (oracle/gen_golden.py: gen_leaflet) and the CPU oracle at a size with many tiles."""

import json

import numpy as np
import pytest

from conftest import load_golden, relerr

pytestmark = pytest.mark.gpu


def _gp(g, key):
    return json.loads(str(g[key]))


@pytest.mark.parametrize("name", ["ico5", "disk5"])
@pytest.mark.parametrize("mass", ["lumped", "consistent"])
def test_leaflet_plugins_match_reference(name, mass):
    """Every leaflet module through the reference's plugin signature on host arrays."""
    from membrane_solver_amd import _lib as L
    from membrane_solver_amd.core.parameters import GlobalParameters, ParameterResolver
    from membrane_solver_amd.geometry.mesh import ArrayMesh
    from membrane_solver_amd.runtime.energy_manager import EnergyModuleManager

    g = load_golden("tilt_leaflet_cases.npz")
    key = f"{name}_{mass}"
    gp = GlobalParameters(_gp(g, key + "_gp_json"))
    pos, tri = g[name + "_positions"], g[name + "_tri"]
    tin, tout = g[name + "_tilts_in"], g[name + "_tilts_out"]
    mesh = ArrayMesh(pos, tri, global_parameters=gp, tilts_in=tin, tilts_out=tout)
    res = ParameterResolver(gp)
    em = EnergyModuleManager(["tilt_in", "tilt_out", "tilt_smoothness_in", "tilt_smoothness_out"])
    for mod in ("tilt_in", "tilt_out", "tilt_smoothness_in", "tilt_smoothness_out"):
        lf = mod.rsplit("_", 1)[1]
        module = em.get_module(mod)
        assert module.USES_TILT_LEAFLETS
        grad = np.zeros_like(pos)
        tg = np.zeros_like(pos)
        kw = {"tilt_in_grad_arr": tg} if lf == "in" else {"tilt_out_grad_arr": tg}
        # (tilt_in / tilt_out with consistent mass: the tilt gradient k A/12 (2 t_k + t_a + t_b) of
        # tilt_leaflet.py:124-150 is k_tilt's second gather pass -- SURVEY row a15 in full)
        E = module.compute_energy_and_gradient_array(mesh, gp, res, positions=pos, index_map=mesh.vertex_index_to_row,
                                                     grad_arr=grad, tilts_in=tin, tilts_out=tout, **kw)
        assert abs(E - g[f"{key}_{mod}_E"]) <= 1e-12 * abs(g[f"{key}_{mod}_E"])
        assert relerr(grad, g[f"{key}_{mod}_grad"]) < 1e-10 or not np.any(g[f"{key}_{mod}_grad"])
        if not np.any(g[f"{key}_{mod}_grad"]):
            assert not np.any(grad)
        if kw:
            assert relerr(tg, g[f"{key}_{mod}_tilt_grad"]) < 1e-10
        E2 = module.compute_energy_array(mesh, gp, res, positions=pos, index_map=mesh.vertex_index_to_row,
                                         tilts_in=tin, tilts_out=tout)
        assert abs(E2 - E) <= 1e-13 * abs(E)


@pytest.mark.parametrize("name", ["ico5", "disk5"])
@pytest.mark.parametrize("mass", ["lumped", "consistent"])
def test_leaflet_relaxation_evaluation_matches_reference(name, mass):
    """compute_energy_and_leaflet_tilt_gradients_array with vertex areas (what relax_leaflet_tilts calls):
    all four modules at once, vertex-area form of the magnitude modules whatever the mass mode."""
    from membrane_solver_amd import _lib as L
    from membrane_solver_amd.device import DeviceMesh

    g = load_golden("tilt_leaflet_cases.npz")
    key = f"{name}_{mass}"
    gp = _gp(g, key + "_gp_json")
    pos, tri = g[name + "_positions"], g[name + "_tri"]
    dm = DeviceMesh(pos, tri, boundary=g[name + "_is_boundary"])
    mass_in = gp["tilt_mass_mode_in"]
    mass_out = gp["tilt_mass_mode"]
    dm.set_leaflet_tilts("in", g[name + "_tilts_in"], tilt_modulus=gp["tilt_modulus_in"], mass_mode=mass_in,
                         smoothness=gp["bending_modulus"])
    dm.set_leaflet_tilts("out", g[name + "_tilts_out"], tilt_modulus=gp["tilt_modulus_out"], mass_mode=mass_out,
                         smoothness=gp["bending_modulus_out"])
    dm.set_params(modules=L.MS_MOD_TILT_IN | L.MS_MOD_TILT_OUT | L.MS_MOD_TILT_SMOOTH_IN | L.MS_MOD_TILT_SMOOTH_OUT)
    E, gi, go = dm.leaflet_tilt_energy_and_gradient()
    assert abs(E - g[key + "_relax_E"]) <= 1e-12 * abs(g[key + "_relax_E"])
    assert relerr(gi, g[key + "_relax_grad_in"]) < 1e-10
    assert relerr(go, g[key + "_relax_grad_out"]) < 1e-10
    # the ordinary evaluation sums the modules' own (mass-mode dependent) energies
    want = sum(float(g[f"{key}_{m}_E"]) for m in ("tilt_in", "tilt_out", "tilt_smoothness_in", "tilt_smoothness_out"))
    e = dm.energy()
    assert abs(e[3] - want) <= 1e-12 * abs(want)
    sc = dm.fetch_scalars()
    assert abs(sc[L.MS_S_ETILT_IN] - g[f"{key}_tilt_in_E"]) <= 1e-12 * abs(g[f"{key}_tilt_in_E"])
    assert abs(sc[L.MS_S_ETS_OUT] - g[f"{key}_tilt_smoothness_out_E"]) <= 1e-12 * abs(g[f"{key}_tilt_smoothness_out_E"])
    _e, grad = dm.energy_and_gradient()
    want_g = g[f"{key}_tilt_in_grad"] + g[f"{key}_tilt_out_grad"]
    assert relerr(grad, want_g) < 1e-10
    dm.close()


LEAFLET_TRAJ = {
    "traj_ico4_gd_leaflet_nested_cg.npz": "gd",
    "traj_ico4_cg_leaflet_coupled_gd.npz": "cg",
    "traj_ico4_cg_leaflet_plaincg.npz": "cg",
    "traj_disk5_gd_leaflet_consistent_backtrack.npz": "gd",
}


def _leaflet_minimizer(g, kind, observe, tile=0):
    from membrane_solver_amd.geometry.mesh import ArrayMesh
    from membrane_solver_amd.runtime.constraint_manager import ConstraintModuleManager
    from membrane_solver_amd.runtime.energy_manager import EnergyModuleManager
    from membrane_solver_amd.runtime.minimizer import Minimizer
    from membrane_solver_amd.runtime.steppers import ConjugateGradient, GradientDescent

    mods = [str(m) for m in g["modules"]]
    mesh = ArrayMesh(g["positions0"], g["tri"], fixed=g["fixed"], surface_tension=g["gamma"],
                     tilts_in=g["tilts_in0"], tilts_out=g["tilts_out0"], tilt_fixed_in=g["tilt_fixed_in"],
                     tilt_fixed_out=g["tilt_fixed_out"], global_parameters=_gp(g, "gp_json"), energy_modules=mods,
                     constraint_modules=[])
    stepper = GradientDescent() if kind == "gd" else ConjugateGradient()
    log = []
    if observe:
        orig = stepper.device_step

        def logged(dm, m, step_size, tol=0.0):
            r = orig(dm, m, step_size, tol=tol)
            log.append((float(r.success), r.next_step, r.energy))
            return r

        stepper.device_step = logged
    mz = Minimizer(mesh, mesh.global_parameters, stepper, EnergyModuleManager(mods), ConstraintModuleManager([]),
                   quiet=True, step_size=float(g["step_size0"]), tile_vertices=tile)
    return mesh, mz, log


@pytest.mark.parametrize("fname", sorted(LEAFLET_TRAJ))
def test_minimizer_reproduces_leaflet_trajectory(fname):
    g = load_golden(fname)
    mesh, mz, log = _leaflet_minimizer(g, LEAFLET_TRAJ[fname], observe=True)
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
    assert relerr(mesh.tilts_in_view(), g["tilts_in_final"]) < 1e-8
    assert relerr(mesh.tilts_out_view(), g["tilts_out_final"]) < 1e-8
    assert abs(res["energy"] - g["E_final"]) <= 1e-9 * abs(g["E_final"])
    bd = mz.compute_energy_breakdown()
    assert abs(sum(bd.values()) - res["energy"]) <= 1e-12 * abs(res["energy"])
    # the same run with the whole loop inside the library (ms_minimize)
    mesh2, mz2, _ = _leaflet_minimizer(g, LEAFLET_TRAJ[fname], observe=False)
    res2 = mz2.minimize(int(g["n_steps"]))
    assert relerr(mesh2.positions_view(), g["positions_final"]) < 1e-8
    assert relerr(mesh2.tilts_in_view(), g["tilts_in_final"]) < 1e-8
    assert relerr(mesh2.tilts_out_view(), g["tilts_out_final"]) < 1e-8
    assert abs(res2["energy"] - g["E_final"]) <= 1e-9 * abs(g["E_final"])


def test_leaflet_midsize_matches_oracle():
    """131 220 facets, many tiles: device vs the CPU oracle for the leaflet energies, the shape gradient and the
    relaxation's tilt gradients; then one relaxation against the oracle port's (iterations, evaluations, result)."""
    from membrane_solver_amd import _lib as L
    from membrane_solver_amd import meshgen
    from membrane_solver_amd.device import DeviceMesh
    from oracle import minimizer_port as mp
    from oracle import ms_oracle as orc

    P, T = meshgen.icosphere(81)
    P = meshgen.smooth_displace(P, 0.05)
    rng = np.random.default_rng(3)
    nrm = mp.unit_vertex_normals(P, T)
    tin = 0.2 * rng.normal(size=P.shape)
    tin -= np.einsum("ij,ij->i", tin, nrm)[:, None] * nrm
    tout = 0.15 * rng.normal(size=P.shape)
    tout -= np.einsum("ij,ij->i", tout, nrm)[:, None] * nrm
    fin = np.zeros(len(P), bool)
    fin[::13] = True
    gp = {"tilt_modulus_in": 1.3, "tilt_modulus_out": 0.7, "tilt_mass_mode_out": "consistent", "bending_modulus": 0.4,
          "tilt_solve_mode": "nested", "tilt_solver": "cg", "tilt_step_size": 0.05, "tilt_inner_steps": 3}
    mods = ["tilt_in", "tilt_out", "tilt_smoothness_in", "tilt_smoothness_out"]
    p = mp.Problem(positions=P, tri=T, tilts_in=tin, tilts_out=tout, tilt_fixed_in=fin, energy_modules=mods, gp=gp)
    dm = DeviceMesh(P, T)
    dm.set_leaflet_tilts("in", tin, tilt_fixed=fin, tilt_modulus=1.3, smoothness=0.4)
    dm.set_leaflet_tilts("out", tout, tilt_modulus=0.7, mass_mode="consistent", smoothness=0.4)
    dm.set_params(modules=L.MS_MOD_TILT_IN | L.MS_MOD_TILT_OUT | L.MS_MOD_TILT_SMOOTH_IN | L.MS_MOD_TILT_SMOOTH_OUT)
    E_ref, g_ref = mp.energy_and_gradient(p, P)
    e, grad = dm.energy_and_gradient()
    assert abs(e.sum() - E_ref) <= 1e-11 * abs(E_ref)
    assert relerr(grad, g_ref) < 1e-10
    va = orc.barycentric_vertex_areas(P, T)
    Er, gi_ref, go_ref = mp.energy_and_leaflet_tilt_gradients(p, P, tin, tout, va)
    E, gi, go = dm.leaflet_tilt_energy_and_gradient()
    assert abs(E - Er) <= 1e-11 * abs(Er)
    assert relerr(gi, gi_ref) < 1e-10 and relerr(go, go_ref) < 1e-10
    stats = mp.relax_leaflet_tilts(p, P)
    it, ev = dm.relax_leaflet_tilts(solver="cg", max_iters=3, step_size=0.05, jacobi=True)
    assert (it, ev) == (stats["iters"], stats["evals"])
    assert relerr(dm.get_leaflet_tilts("in"), p.tilts_in) < 1e-9
    assert relerr(dm.get_leaflet_tilts("out"), p.tilts_out) < 1e-9
    assert np.allclose(dm.get_leaflet_tilts("in")[fin], (tin - 0)[fin], rtol=0, atol=1e-12)  # clamped rows
    dm.close()


def test_leaflet_guards():
    from membrane_solver_amd import _lib as L
    from membrane_solver_amd import meshgen
    from membrane_solver_amd.device import DeviceMesh
    from membrane_solver_amd.geometry.mesh import ArrayMesh
    from membrane_solver_amd.runtime.constraint_manager import ConstraintModuleManager
    from membrane_solver_amd.runtime.energy_manager import EnergyModuleManager
    from membrane_solver_amd.runtime.minimizer import Minimizer
    from membrane_solver_amd.runtime.steppers import GradientDescent

    P, T = meshgen.icosphere(3)
    dm = DeviceMesh(P, T)
    dm.set_params(modules=L.MS_MOD_TILT_IN)
    with pytest.raises(L.MembraneHipError, match="never set|not called"):
        dm.energy()
    with pytest.raises(L.MembraneHipError, match="no leaflet module"):
        dm.set_params(modules=L.MS_MOD_SURFACE)
        dm.relax_leaflet_tilts(max_iters=2, step_size=0.1)
    with pytest.raises(ValueError):
        dm.set_leaflet_tilts("in", np.zeros((len(P), 3)), mass_mode="diagonal")
    dm.close()
    for bad in ({"tilt_cg_rejection_fallback": "gd"}, {"leaflet_out_absent_presets": ["disk"]},
                {"tilt_transport_model": "connection_v1"}):
        gp = dict({"tilt_modulus_in": 1.0}, **bad)
        mods = ["surface", "tilt_in"]
        mesh = ArrayMesh(P, T, global_parameters=gp, energy_modules=mods)
        mz = Minimizer(mesh, mesh.global_parameters, GradientDescent(), EnergyModuleManager(mods),
                       ConstraintModuleManager([]), quiet=True)
        with pytest.raises(L.MembraneHipError, match="outside the HIP hot path"):
            mz.compute_energy()
    mods = ["surface", "tilt", "tilt_in"]
    mesh = ArrayMesh(P, T, global_parameters={"tilt_modulus_in": 1.0, "tilt_rigidity": 1.0}, energy_modules=mods)
    mz = Minimizer(mesh, mesh.global_parameters, GradientDescent(), EnergyModuleManager(mods),
                   ConstraintModuleManager([]), quiet=True)
    with pytest.raises(L.MembraneHipError, match="together"):
        mz.compute_energy()


# ---------------------------------------------------------------------------
# bending_tilt_in / bending_tilt_out (bending_tilt_leaflet.py, default options, analytic gradient)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["ico5", "disk5"])
def test_bending_tilt_leaflet_plugins_match_reference(name):
    from membrane_solver_amd import _lib as L
    from membrane_solver_amd.core.parameters import GlobalParameters, ParameterResolver
    from membrane_solver_amd.geometry.mesh import ArrayMesh
    from membrane_solver_amd.runtime.energy_manager import EnergyModuleManager

    g = load_golden("bending_tilt_leaflet_cases.npz")
    gp = GlobalParameters(_gp(g, "gp_json"))
    pos, tri = g[name + "_positions"], g[name + "_tri"]
    tin, tout = g[name + "_tilts_in"], g[name + "_tilts_out"]
    mesh = ArrayMesh(pos, tri, global_parameters=gp, tilts_in=tin, tilts_out=tout)
    res = ParameterResolver(gp)
    em = EnergyModuleManager(["bending_tilt_in", "bending_tilt_out"])
    for lf in ("in", "out"):
        module = em.get_module(f"bending_tilt_{lf}")
        grad, tg = np.zeros_like(pos), np.zeros_like(pos)
        kw = {"tilt_in_grad_arr": tg} if lf == "in" else {"tilt_out_grad_arr": tg}
        E = module.compute_energy_and_gradient_array(mesh, gp, res, positions=pos, index_map=mesh.vertex_index_to_row,
                                                     grad_arr=grad, tilts_in=tin, tilts_out=tout, **kw)
        ref = g[f"{name}_bending_tilt_{lf}_E"]
        assert abs(E - ref) <= 1e-12 * abs(ref)
        assert relerr(grad, g[f"{name}_bending_tilt_{lf}_grad"]) < 1e-10
        assert relerr(tg, g[f"{name}_bending_tilt_{lf}_tilt_grad"]) < 1e-10
        E2 = module.compute_energy_and_gradient_array(mesh, gp, res, positions=pos, index_map=mesh.vertex_index_to_row,
                                                      grad_arr=None, tilts_in=tin, tilts_out=tout)
        assert abs(E2 - ref) <= 1e-12 * abs(ref)
    gp.set("bending_gradient_mode", "approx")
    with pytest.raises(L.MembraneHipError, match="analytic"):
        em.get_module("bending_tilt_in").compute_energy_and_gradient_array(
            mesh, gp, res, positions=pos, index_map=mesh.vertex_index_to_row, grad_arr=np.zeros_like(pos),
            tilts_in=tin, tilts_out=tout)


BTL_TRAJ = {"traj_ico4_gd_btl_nested_cg.npz": "gd", "traj_ico4_cg_btl_coupled_gd.npz": "cg",
            "traj_disk5_gd_btl_backtrack.npz": "gd"}


@pytest.mark.parametrize("fname", sorted(BTL_TRAJ))
def test_minimizer_reproduces_bending_tilt_leaflet_trajectory(fname):
    g = load_golden(fname)
    mesh, mz, log = _leaflet_minimizer(g, BTL_TRAJ[fname], observe=True)
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
    assert relerr(mesh.tilts_in_view(), g["tilts_in_final"]) < 1e-8
    assert relerr(mesh.tilts_out_view(), g["tilts_out_final"]) < 1e-8
    assert abs(res["energy"] - g["E_final"]) <= 1e-9 * abs(g["E_final"])
    bd = mz.compute_energy_breakdown()
    assert abs(sum(bd.values()) - res["energy"]) <= 1e-12 * abs(res["energy"])
    mesh2, mz2, _ = _leaflet_minimizer(g, BTL_TRAJ[fname], observe=False)
    res2 = mz2.minimize(int(g["n_steps"]))
    assert relerr(mesh2.positions_view(), g["positions_final"]) < 1e-8
    assert relerr(mesh2.tilts_in_view(), g["tilts_in_final"]) < 1e-8
    assert abs(res2["energy"] - g["E_final"]) <= 1e-9 * abs(g["E_final"])


def test_bending_tilt_leaflet_midsize_matches_oracle():
    """131 220 facets (many tiles, default tile size): both leaflets at once against the CPU oracle."""
    from membrane_solver_amd import _lib as L
    from membrane_solver_amd import meshgen
    from membrane_solver_amd.device import DeviceMesh
    from oracle import minimizer_port as mp

    P, T = meshgen.icosphere(81)
    P = meshgen.smooth_displace(P, 0.05)
    rng = np.random.default_rng(4)
    nrm = mp.unit_vertex_normals(P, T)
    tin = 0.2 * rng.normal(size=P.shape)
    tin -= np.einsum("ij,ij->i", tin, nrm)[:, None] * nrm
    tout = 0.15 * rng.normal(size=P.shape)
    tout -= np.einsum("ij,ij->i", tout, nrm)[:, None] * nrm
    gp = {"bending_modulus": 0.8, "bending_modulus_in": 1.2, "spontaneous_curvature": 0.05,
          "spontaneous_curvature_out": -0.1, "tilt_modulus_in": 1.3, "surface_tension": 1.0}
    mods = ["surface", "tilt_in", "bending_tilt_in", "bending_tilt_out"]
    p = mp.Problem(positions=P, tri=T, tilts_in=tin, tilts_out=tout, energy_modules=mods, gp=gp)
    dm = DeviceMesh(P, T)
    dm.set_leaflet_tilts("in", tin, tilt_modulus=1.3)
    dm.set_leaflet_tilts("out", tout)
    dm.set_leaflet_bending("in", 1.2, 0.05)
    dm.set_leaflet_bending("out", 0.8, -0.1)
    dm.set_params(modules=L.MS_MOD_SURFACE | L.MS_MOD_TILT_IN | L.MS_MOD_BENDING_TILT_IN | L.MS_MOD_BENDING_TILT_OUT)
    E_ref, g_ref = mp.energy_and_gradient(p, P)
    e, grad = dm.energy_and_gradient()
    assert abs(e.sum() - E_ref) <= 1e-11 * abs(E_ref)
    assert relerr(grad, g_ref) < 1e-10
    from oracle import ms_oracle as orc
    va = orc.barycentric_vertex_areas(P, T)
    Er, gi_ref, go_ref = mp.energy_and_leaflet_tilt_gradients(p, P, tin, tout, va)
    E, gi, go = dm.leaflet_tilt_energy_and_gradient()
    assert abs(E - Er) <= 1e-11 * abs(Er)
    assert relerr(gi, gi_ref) < 1e-10 and relerr(go, go_ref) < 1e-10
    dm.close()


# ---------------------------------------------------------------------------
# tilt_disk_target_in / tilt_disk_target_out (tilt_disk_target_in.py:160-286)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["disk6", "ico5"])
@pytest.mark.parametrize("tag", ["bessel", "linear", "moduli"])
def test_disk_target_plugins_match_reference(name, tag):
    from membrane_solver_amd.core.parameters import GlobalParameters, ParameterResolver
    from membrane_solver_amd.geometry.mesh import ArrayMesh
    from membrane_solver_amd.runtime.energy_manager import EnergyModuleManager

    g = load_golden("tilt_disk_target_cases.npz")
    key = f"{name}_{tag}"
    gp = GlobalParameters(_gp(g, key + "_gp_json"))
    pos, tri, rows = g[name + "_positions"], g[name + "_tri"], g[name + "_disk_rows"]
    tin, tout = g[name + "_tilts_in"], g[name + "_tilts_out"]
    mesh = ArrayMesh(pos, tri, global_parameters=gp, tilts_in=tin, tilts_out=tout, disk_rows_in=rows, disk_rows_out=rows)
    res = ParameterResolver(gp)
    em = EnergyModuleManager(["tilt_disk_target_in", "tilt_disk_target_out"])
    for lf in ("in", "out"):
        module = em.get_module(f"tilt_disk_target_{lf}")
        grad, tg = np.zeros_like(pos), np.zeros_like(pos)
        kw = {"tilt_in_grad_arr": tg} if lf == "in" else {"tilt_out_grad_arr": tg}
        E = module.compute_energy_and_gradient_array(mesh, gp, res, positions=pos, index_map=mesh.vertex_index_to_row,
                                                     grad_arr=grad, tilts_in=tin, tilts_out=tout, **kw)
        ref = g[f"{key}_tilt_disk_target_{lf}_E"]
        assert abs(E - ref) <= 1e-12 * abs(ref)
        assert relerr(grad, g[f"{key}_tilt_disk_target_{lf}_grad"]) < 1e-10
        assert relerr(tg, g[f"{key}_tilt_disk_target_{lf}_tilt_grad"]) < 1e-10


DISK_TRAJ = {"traj_disk6_gd_disktarget_nested_cg.npz": "gd", "traj_disk6_cg_disktarget_coupled_gd.npz": "cg"}


@pytest.mark.parametrize("fname", sorted(DISK_TRAJ))
def test_minimizer_reproduces_disk_target_trajectory(fname):
    g = load_golden(fname)
    for observe in (True, False):
        mesh, mz, log = _leaflet_minimizer(g, DISK_TRAJ[fname], observe=observe)
        mesh.disk_rows_in = mesh.disk_rows_out = g["disk_rows"]
        if observe:
            E0, grad0 = mz.compute_energy_and_gradient_array()
            assert abs(E0 - g["E0"]) <= 1e-12 * abs(g["E0"])
            assert relerr(grad0, g["grad0"]) < 1e-10
        res = mz.minimize(int(g["n_steps"]))
        if observe:
            got, ref = np.array(log), g["step_log"]
            assert got.shape == ref.shape
            assert np.array_equal(got[:, 0], ref[:, 0]), "accept/reject sequence differs from the reference"
            assert np.allclose(got[:, 1], ref[:, 1], rtol=1e-12, atol=0)
            assert np.allclose(got[:, 2], ref[:, 2], rtol=1e-9, atol=0)
            bd = mz.compute_energy_breakdown()
            assert abs(sum(bd.values()) - res["energy"]) <= 1e-12 * abs(res["energy"])
        assert relerr(mesh.positions_view(), g["positions_final"]) < 1e-8
        assert relerr(mesh.tilts_in_view(), g["tilts_in_final"]) < 1e-8
        assert relerr(mesh.tilts_out_view(), g["tilts_out_final"]) < 1e-8
        assert abs(res["energy"] - g["E_final"]) <= 1e-9 * abs(g["E_final"])


def test_disk_target_needs_a_normal():
    from membrane_solver_amd import _lib as L
    from membrane_solver_amd import meshgen
    from membrane_solver_amd.geometry.mesh import ArrayMesh
    from membrane_solver_amd.runtime.constraint_manager import ConstraintModuleManager
    from membrane_solver_amd.runtime.energy_manager import EnergyModuleManager
    from membrane_solver_amd.runtime.minimizer import Minimizer
    from membrane_solver_amd.runtime.steppers import GradientDescent

    P, T = meshgen.icosphere(3)
    gp = {"tilt_disk_target_group_in": "disk", "tilt_disk_target_strength_in": 5.0, "tilt_disk_target_theta_B": 0.3}
    mods = ["surface", "tilt_disk_target_in"]
    mesh = ArrayMesh(P, T, global_parameters=gp, energy_modules=mods, disk_rows_in=np.arange(10))
    mz = Minimizer(mesh, mesh.global_parameters, GradientDescent(), EnergyModuleManager(mods), ConstraintModuleManager([]),
                   quiet=True)
    with pytest.raises(L.MembraneHipError, match="tilt_disk_target_normal"):
        mz.compute_energy()


@pytest.mark.parametrize("fname", ["traj_ico4_gd_leaflet_nested_cg.npz", "traj_ico4_cg_btl_coupled_gd.npz",
                                   "traj_disk5_gd_btl_backtrack.npz", "traj_disk6_cg_disktarget_coupled_gd.npz"])
def test_leaflet_trajectories_with_small_tiles(fname):
    """The same reference trajectories with 64-vertex tiles: several tiles, halos and the generic-size kernel
    instances on the small golden meshes (the default 256-vertex tiling covers them with one tile)."""
    g = load_golden(fname)
    kind = "cg" if "_cg_" in fname else "gd"
    mesh, mz, _ = _leaflet_minimizer(g, kind, observe=False, tile=64)
    if "disk_rows" in g and len(g["disk_rows"]):
        mesh.disk_rows_in = mesh.disk_rows_out = g["disk_rows"]
    E0, grad0 = mz.compute_energy_and_gradient_array()
    assert mesh._hip_mirror.dm.tile_stats()["n_tiles"] > 1
    assert abs(E0 - g["E0"]) <= 1e-12 * abs(g["E0"])
    assert relerr(grad0, g["grad0"]) < 1e-10
    res = mz.minimize(int(g["n_steps"]))
    assert relerr(mesh.positions_view(), g["positions_final"]) < 1e-8
    assert relerr(mesh.tilts_in_view(), g["tilts_in_final"]) < 1e-8
    assert relerr(mesh.tilts_out_view(), g["tilts_out_final"]) < 1e-8
    assert abs(res["energy"] - g["E_final"]) <= 1e-9 * abs(g["E_final"])


@pytest.mark.parametrize("fname", ["traj_ico4_gd_leaflet_nested_cg.npz", "traj_ico4_gd_btl_nested_cg.npz"])
def test_callback_sees_reference_states(fname):
    """minimize(callback=...) hands the callback the mesh of the reference at the top of every iteration: positions and
    both leaflet tilt fields (relaxed and projected so far) equal the reference's per-iteration snapshots."""
    g = load_golden(fname)
    mesh, mz, _ = _leaflet_minimizer(g, "gd", observe=False)
    snaps = []

    def cb(m, i):
        snaps.append((m.positions_view().copy(), m.tilts_in_view().copy(), m.tilts_out_view().copy()))

    mz.minimize(int(g["n_steps"]), callback=cb)
    assert len(snaps) == len(g["positions_iter"])
    for k, (x, tin, tout) in enumerate(snaps):
        assert relerr(x, g["positions_iter"][k]) < 1e-8, k
        assert relerr(tin, g["tilts_in_iter"][k]) < 1e-8, k
        assert relerr(tout, g["tilts_out_iter"][k]) < 1e-8, k


def test_rim_slope_match_out_only_switched_off():
    """The caveolin decks list rim_slope_match_out among the energy modules with strength 0: accepted as a zero,
    refused loudly with a non-zero strength."""
    from membrane_solver_amd import _lib as L
    from membrane_solver_amd import meshgen
    from membrane_solver_amd.geometry.mesh import ArrayMesh
    from membrane_solver_amd.runtime.constraint_manager import ConstraintModuleManager
    from membrane_solver_amd.runtime.energy_manager import EnergyModuleManager
    from membrane_solver_amd.runtime.minimizer import Minimizer
    from membrane_solver_amd.runtime.steppers import GradientDescent

    P, T = meshgen.icosphere(3)
    mods = ["surface", "rim_slope_match_out"]
    for strength, ok in ((0.0, True), (2.0, False)):
        gp = {"surface_tension": 1.0, "rim_slope_match_strength": strength, "rim_slope_match_group": "rim",
              "rim_slope_match_outer_group": "outer"}
        mesh = ArrayMesh(P, T, global_parameters=gp, energy_modules=mods)
        mz = Minimizer(mesh, mesh.global_parameters, GradientDescent(), EnergyModuleManager(mods),
                       ConstraintModuleManager([]), quiet=True)
        if ok:
            bd = mz.compute_energy_breakdown()
            assert bd["rim_slope_match_out"] == 0.0 and bd["surface"] > 0.0
            assert mz.minimize(2)["energy"] > 0.0
        else:
            with pytest.raises(L.MembraneHipError, match="rim_slope_match_strength"):
                mz.compute_energy()


# ---------------------------------------------------------------------------
# BASELINE config 5 on its own deck: meshes/caveolin/kozlov_1disk_3d_tensionless_bilayer_profile.yaml as the
# reference parses it (oracle/gen_golden.py: gen_config5) -- 109 vertices, 204 facets, 12 boundary vertices, the
# deck's tilt_fixed_in/out flags, "disk" group rows, parameters and energy-module list.  Its pin / rim constraint
# modules are outside the hot path and were not kept when the vectors were generated.
# ---------------------------------------------------------------------------
def _config5_mesh(g, tilts_in, tilts_out, modules):
    from membrane_solver_amd.core.parameters import GlobalParameters
    from membrane_solver_amd.geometry.mesh import ArrayMesh

    gp = GlobalParameters(_gp(g, "gp_json"))
    rows = g["disk_rows"]
    mesh = ArrayMesh(g["positions0"], g["tri"], fixed=g["fixed"], surface_tension=g["gamma"], tilts_in=tilts_in,
                     tilts_out=tilts_out, tilt_fixed_in=g["tilt_fixed_in"], tilt_fixed_out=g["tilt_fixed_out"],
                     global_parameters=gp, energy_modules=list(modules), constraint_modules=[],
                     disk_rows_in=rows, disk_rows_out=rows)
    return mesh, gp


def test_config5_deck_plugins_match_reference():
    """Every energy module of the deck through the reference's plugin signature, on the deck's surface with seeded
    tangent tilt fields (the deck's own fields are zero)."""
    from membrane_solver_amd.core.parameters import ParameterResolver
    from membrane_solver_amd.runtime.energy_manager import EnergyModuleManager

    g = load_golden("traj_config5_deck_gd.npz")
    mods = [str(m) for m in g["modules"]]
    assert len(g["positions0"]) == 109 and len(g["tri"]) == 204 and int(g["is_boundary"].sum()) == 12
    assert "rim_slope_match_out" in mods and "bending_tilt_in" in mods and "tilt_disk_target_out" in mods
    tin, tout = g["state_b_tilts_in"], g["state_b_tilts_out"]
    pos = g["positions0"]
    mesh, gp = _config5_mesh(g, tin, tout, mods)
    res = ParameterResolver(gp)
    em = EnergyModuleManager(mods)
    total_E, total_g = 0.0, np.zeros_like(pos)
    for mod in mods:
        module = em.get_module(mod)
        grad, tgi, tgo = np.zeros_like(pos), np.zeros_like(pos), np.zeros_like(pos)
        E = module.compute_energy_and_gradient_array(mesh, gp, res, positions=pos, index_map=mesh.vertex_index_to_row,
                                                     grad_arr=grad, tilts_in=tin, tilts_out=tout,
                                                     tilt_in_grad_arr=tgi, tilt_out_grad_arr=tgo)
        ref = float(g[f"mod_{mod}_E"])
        assert abs(E - ref) <= 1e-12 * max(abs(ref), 1e-300), mod
        for got, key in ((grad, "grad"), (tgi, "tilt_grad_in"), (tgo, "tilt_grad_out")):
            want = g[f"mod_{mod}_{key}"]
            if np.any(want):
                assert relerr(got, want) < 1e-10, (mod, key)
            else:
                assert not np.any(got), (mod, key)
        total_E += E
        total_g += grad
    assert abs(total_E - g["state_b_E"]) <= 1e-12 * abs(g["state_b_E"])


def test_config5_deck_relaxation_and_steps_match_reference():
    """ONE relax_leaflet_tilts call as the deck configures it (coupled mode, Jacobi CG, 40 inner steps, step 0.15)
    from the deck's zero fields, then the deck's `g` steps (fixed step size 0.01, the relaxation at the top of every
    iteration) -- through Minimizer / ms_relax_leaflet_tilts / ms_minimize."""
    g = load_golden("traj_config5_deck_gd.npz")
    mesh, mz, _ = _leaflet_minimizer(g, "gd", observe=True)
    mesh.disk_rows_in = mesh.disk_rows_out = g["disk_rows"]
    _mir, dm = mz._device()
    assert mz._relax_tilts(dm)
    mz._write_back_tilts(dm, _mir)
    assert relerr(mesh.tilts_in_view(), g["relax_tilts_in"]) < 1e-8
    assert relerr(mesh.tilts_out_view(), g["relax_tilts_out"]) < 1e-8
    assert abs(mz.compute_energy() - g["relax_E"]) <= 1e-9 * abs(g["relax_E"])
    for observe in (True, False):
        mesh, mz, log = _leaflet_minimizer(g, "gd", observe=observe)
        mesh.disk_rows_in = mesh.disk_rows_out = g["disk_rows"]
        if observe:
            E0, grad0 = mz.compute_energy_and_gradient_array()
            assert abs(E0 - g["E0"]) <= 1e-12 * abs(g["E0"])
            assert relerr(grad0, g["grad0"]) < 1e-10
        res = mz.minimize(int(g["n_steps"]))
        if observe:
            got, ref = np.array(log), g["step_log"]
            assert got.shape == ref.shape
            assert np.array_equal(got[:, 0], ref[:, 0]), "accept/reject sequence differs from the reference"
            assert np.allclose(got[:, 1], ref[:, 1], rtol=1e-12, atol=0)
            assert np.allclose(got[:, 2], ref[:, 2], rtol=1e-9, atol=0)
        assert relerr(mesh.positions_view(), g["positions_final"]) < 1e-8
        assert relerr(mesh.tilts_in_view(), g["tilts_in_final"]) < 1e-8
        assert relerr(mesh.tilts_out_view(), g["tilts_out_final"]) < 1e-8
        assert abs(res["energy"] - g["E_final"]) <= 1e-9 * abs(g["E_final"])
