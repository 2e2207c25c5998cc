"""Edge cases of the domain, HIP path vs CPU oracle: degenerate and repeated-index
triangles, out-of-range indices, isolated vertices, empty facet list, everything fixed,
non-uniform per-facet / per-vertex parameters, a body that covers only some facets,
oversized valence (error path), and state round trips."""

import numpy as np
import pytest

from conftest import relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from membrane_solver_amd import _lib

    _lib.lib()
    return _lib


def _oracle_eg(P, T, gamma, kappa, c0, isb):
    from oracle import ms_oracle as orc

    g = np.zeros_like(P)
    Es = orc.surface_energy_and_gradient(P, T, gamma, g)
    Eb = orc.bending_energy_and_gradient(P, T, kappa, c0, isb, grad=g)
    return Es, Eb, g


def test_degenerate_repeated_and_out_of_range_triangles(L):
    from membrane_solver_amd import meshgen
    from membrane_solver_amd.device import DeviceMesh
    from oracle import ms_oracle as orc

    P, T = meshgen.icosphere(3)
    P = meshgen.smooth_displace(P, 0.1)
    T = T.copy()
    nv = len(P)
    P = np.vstack([P, [[5.0, 5.0, 5.0], [6.0, 5.0, 5.0]]])  # two isolated vertices
    extra = np.array([[0, 0, 1],          # repeated index (zero area)
                      [2, 3, 2],          # repeated index
                      [4, 5, nv + 7],     # out of range -> skipped by the reference kernels
                      [-1, 2, 3]], dtype=np.int32)
    T2 = np.vstack([T, extra]).astype(np.int32)
    # collapse one real triangle to zero area (A2 < 1e-12 branch)
    P[T[10, 2]] = P[T[10, 1]]
    gamma = np.linspace(0.5, 1.5, len(T2))
    gref = np.zeros_like(P)
    Eref = orc.surface_energy_and_gradient(P, T2, gamma, gref)
    dm = DeviceMesh(P, T2, tile_vertices=64)
    dm.set_surface_tension(gamma)
    dm.set_params(modules=L.MS_MOD_SURFACE)
    e, g = dm.energy_and_gradient()
    assert abs(e[0] - Eref) <= 1e-12 * abs(Eref)
    assert relerr(g, gref) < 1e-10
    assert np.all(g[-2:] == 0.0)
    dm.close()


def test_empty_facet_list_and_single_triangle(L):
    from membrane_solver_amd.device import DeviceMesh

    P = np.array([[0.0, 0, 0], [1, 0, 0], [0, 1, 0]])
    dm = DeviceMesh(P, np.zeros((0, 3), np.int32))
    dm.set_params(modules=L.MS_MOD_SURFACE | L.MS_MOD_BENDING)
    e, g = dm.energy_and_gradient()
    assert np.all(e == 0.0) and np.all(g == 0.0)
    r = dm.step(stepper=L.MS_STEPPER_GD, step_size=1e-3)
    assert r.converged and r.success and r.grad_norm == 0.0
    dm.close()
    # tests/test_surface.py:61-93 of the reference: right triangle, gamma = 2 -> E = 1
    dm = DeviceMesh(P, np.array([[0, 1, 2]], np.int32))
    dm.set_surface_tension(np.array([2.0]))
    dm.set_params(modules=L.MS_MOD_SURFACE)
    e, g = dm.energy_and_gradient()
    assert abs(e[0] - 1.0) < 1e-15 and np.all(np.isfinite(g))
    assert np.allclose(g.sum(axis=0), 0.0, atol=1e-15)
    dm.close()


def test_all_fixed_and_partially_fixed(L):
    from membrane_solver_amd import meshgen
    from membrane_solver_amd.device import DeviceMesh

    P, T = meshgen.icosphere(4)
    dm = DeviceMesh(P, T, fixed=np.ones(len(P), bool))
    dm.set_params(modules=L.MS_MOD_SURFACE)
    e, g = dm.energy_and_gradient()
    assert e[0] > 0 and np.all(g == 0.0)
    r = dm.step(stepper=L.MS_STEPPER_CG, step_size=1e-2)
    assert r.converged
    assert np.array_equal(dm.get_positions(), P)
    dm.close()


def test_nonuniform_parameters_boundary_and_body_subset(L):
    from membrane_solver_amd import meshgen
    from membrane_solver_amd.device import DeviceMesh
    from oracle import ms_oracle as orc

    P, T, B = meshgen.disk_patch(8, jitter=0.25, seed=9)
    rng = np.random.default_rng(4)
    nv, nf = len(P), len(T)
    gamma = 0.5 + rng.random(nf)
    kappa = 0.2 + rng.random(nv)
    c0 = rng.normal(scale=0.4, size=nv)
    Es, Eb, gref = _oracle_eg(P, T, gamma, kappa, c0, B)
    body = (rng.random(nf) < 0.6)
    rows = np.flatnonzero(body).astype(np.int32)
    Vref = orc.volume(P, T, rows)
    gCref = np.zeros_like(P)
    orc.volume_gradient(P, T, gCref, body_rows=rows)
    for tile in (64, 256):
        dm = DeviceMesh(P, T, boundary=B, body_facets=body, tile_vertices=tile)
        dm.set_surface_tension(gamma)
        dm.set_bending_params(kappa, c0)
        dm.set_params(modules=L.MS_MOD_SURFACE | L.MS_MOD_BENDING)
        e, g = dm.energy_and_gradient()
        assert abs(e[0] - Es) <= 1e-12 * abs(Es) and abs(e[1] - Eb) <= 1e-12 * abs(Eb)
        assert relerr(g, gref) < 1e-10
        dm.set_params(modules=L.MS_CON_VOLUME)
        dm.energy_and_gradient(want_grad=False)
        assert abs(dm.fetch_scalars()[L.MS_S_VOL] - Vref) <= 1e-12 * abs(Vref)
        assert relerr(dm.get_vertex_buffer(L.MS_BUF_GC), gCref) < 1e-12
        dm.close()


def test_willmore_and_approx_on_open_noisy_mesh(L):
    from membrane_solver_amd import meshgen
    from membrane_solver_amd.device import DeviceMesh
    from oracle import ms_oracle as orc

    P, T, B = meshgen.disk_patch(6, jitter=0.3, seed=1)
    nv = len(P)
    kappa, c0 = np.full(nv, 1.3), np.zeros(nv)
    dm = DeviceMesh(P, T, boundary=B)
    dm.set_bending_params(kappa, c0)
    for model, mid in (("willmore", L.MS_BEND_WILLMORE), ("helfrich", L.MS_BEND_HELFRICH)):
        for mode, gid in (("analytic", L.MS_GRAD_ANALYTIC), ("approx", L.MS_GRAD_APPROX)):
            gref = np.zeros_like(P)
            Eref = orc.bending_energy_and_gradient(P, T, kappa, c0, B, model=model, mode=mode, grad=gref)
            dm.set_params(modules=L.MS_MOD_BENDING, bending_model=mid, bending_grad_mode=gid)
            e, g = dm.energy_and_gradient()
            assert abs(e[1] - Eref) <= 1e-12 * abs(Eref), (model, mode)
            assert relerr(g, gref) < 1e-10, (model, mode)
    dm.close()


def test_tile_size_and_instance_choice_do_not_change_the_results(L):
    """The tile size decides which workgroup sums what and which kernel instances run (T = 256: packed facet records
    and the lean gradient instance, also with fewer rows per tile than threads; 128 / 64: the runtime-size instances),
    never what is summed: energies agree to
    1e-12 and gradients to a bound that is MEASURED here -- five times the larger of (a) the run-to-run difference of
    the default LDS-atomic sums on one tiling and (b) the CPU oracle's own summation-order noise (serial against OpenMP
    build of the same source).  The bending back-propagation amplifies last-bit differences of the vertex sums by
    ~1/h^2, on the CPU exactly as on the GPU, which is what both measurements show."""
    from membrane_solver_amd import meshgen
    from membrane_solver_amd.device import DeviceMesh
    from oracle import ms_oracle as orc

    P, T = meshgen.icosphere(40)  # 32 000 facets, 63 tiles of 256 vertices
    P = meshgen.smooth_displace(P, 0.08)
    nv, nf = len(P), len(T)
    kappa, c0 = np.ones(nv), np.full(nv, 0.15)
    res, repeat = {}, 0.0
    for tile in (256, 200, 128, 64):  # (200: tiles of 200 rows on the 256-thread instances)
        dm = DeviceMesh(P, T, tile_vertices=tile)
        dm.set_surface_tension(np.ones(nf))
        dm.set_bending_params(kappa, c0)
        dm.set_params(modules=L.MS_MOD_SURFACE | L.MS_MOD_BENDING)
        e, g = dm.energy_and_gradient()
        for _ in range(3):
            _e2, g2 = dm.energy_and_gradient()
            repeat = max(repeat, relerr(g2, g))
        r = dm.step(stepper=L.MS_STEPPER_CG, step_size=1e-5)
        res[tile] = (e.copy(), g.copy(), r.energy, r.trials)
        dm.close()

    def oracle(omp):
        orc.use_openmp(omp)
        try:
            gr = np.zeros_like(P)
            orc.surface_energy_and_gradient(P, T, np.ones(nf), gr)
            orc.bending_energy_and_gradient(P, T, kappa, c0, np.zeros(nv, bool), grad=gr)
        finally:
            orc.use_openmp(False)
        return gr

    noise = relerr(oracle(True), oracle(False))
    tol = 5.0 * max(repeat, noise)
    assert 0.0 < tol < 1e-10, (repeat, noise)
    e0, g0, E0, tr0 = res[256]
    for key, (e, g, E, tr) in res.items():
        assert np.allclose(e, e0, rtol=1e-12, atol=0), key
        assert relerr(g, g0) < tol, (key, relerr(g, g0), repeat, noise)
        assert abs(E - E0) <= 1e-12 * abs(E0) and tr == tr0, key
    # the fixed-order (CSR gather) instances with fewer rows per tile than threads
    det = {}
    for tile in (256, 200):
        dm = DeviceMesh(P, T, tile_vertices=tile)
        dm.set_deterministic(True)
        dm.set_surface_tension(np.ones(nf))
        dm.set_bending_params(kappa, c0)
        dm.set_params(modules=L.MS_MOD_SURFACE | L.MS_MOD_BENDING | L.MS_CON_VOLUME, target_volume=4.0)
        det[tile] = dm.energy_and_gradient()
        e_again, g_again = dm.energy_and_gradient()
        assert np.array_equal(g_again, det[tile][1]) and np.array_equal(e_again, det[tile][0])  # bitwise repeatable
        dm.close()
    assert np.allclose(det[200][0], det[256][0], rtol=1e-12, atol=0)
    assert relerr(det[200][1], det[256][1]) < tol


def test_high_valence_hub_takes_the_unpacked_record_path(L):
    """A cone of 1500 facets around one apex: the apex's tile needs ~1750 LDS slots, more than the 10-bit corner
    fields of the packed facet records address, so the T = 256 instances (and the lean gradient instance) must not
    be picked -- the generic instances with 8-byte records run, and agree with the oracle."""
    from membrane_solver_amd.device import DeviceMesh

    n = 1500
    ang = np.linspace(0, 2 * np.pi, n, endpoint=False)
    rim = np.column_stack([np.cos(ang), np.sin(ang), 0.05 * np.sin(3 * ang)])
    P = np.vstack([[0.0, 0.0, 0.6], rim])
    T = np.column_stack([np.zeros(n, int), 1 + np.arange(n), 1 + (np.arange(n) + 1) % n]).astype(np.int32)
    nv, nf = len(P), len(T)
    B = np.ones(nv, bool)
    B[0] = False
    gamma, kappa, c0 = np.full(nf, 1.3), np.full(nv, 0.7), np.zeros(nv)
    Es, Eb, gref = _oracle_eg(P, T, gamma, kappa, c0, B)
    dm = DeviceMesh(P, T, boundary=B, tile_vertices=256)
    assert dm.tile_stats()["max_halo"] + 256 > 1024
    dm.set_surface_tension(gamma)
    dm.set_bending_params(kappa, c0)
    dm.set_params(modules=L.MS_MOD_SURFACE | L.MS_MOD_BENDING)
    e, g = dm.energy_and_gradient()
    assert abs(e[0] - Es) <= 1e-12 * abs(Es) and abs(e[1] - Eb) <= 1e-11 * abs(Eb)
    assert relerr(g, gref) < 1e-10
    r = dm.step(stepper=L.MS_STEPPER_GD, step_size=1e-4)
    assert r.trials >= 1
    dm.close()


def test_oversized_valence_and_bad_arguments_are_errors_not_crashes(L):
    from membrane_solver_amd.device import DeviceMesh

    # a fan with 70 000 triangles around vertex 0: its patch cannot fit uint16 LDS slots
    n = 70000
    ang = np.linspace(0, 2 * np.pi, n, endpoint=False)
    P = np.vstack([[0.0, 0, 0], np.column_stack([np.cos(ang), np.sin(ang), np.zeros(n)])])
    T = np.column_stack([np.zeros(n, int), 1 + np.arange(n), 1 + (np.arange(n) + 1) % n]).astype(np.int32)
    with pytest.raises(L.MembraneHipError, match="TILE_CAPACITY"):
        DeviceMesh(P, T, tile_vertices=64)
    with pytest.raises(L.MembraneHipError, match="NaN"):
        DeviceMesh(np.array([[0.0, np.nan, 0], [1, 0, 0], [0, 1, 0]]), np.array([[0, 1, 2]], np.int32))
    with pytest.raises(L.MembraneHipError, match="tile_vertices"):
        DeviceMesh(P[:10], T[:3] % 10, tile_vertices=100)
    dm = DeviceMesh(P[:10], (T[:3] % 10).astype(np.int32))
    with pytest.raises(ValueError):
        dm.set_surface_tension(np.ones(7))
    dm.set_params(modules=L.MS_MOD_BENDING)
    with pytest.raises(L.MembraneHipError, match="bad bending_model"):
        dm.set_params(modules=L.MS_MOD_BENDING, bending_model=7)
    dm.close()


def test_set_positions_and_reevaluate_is_consistent(L, deterministic):
    from membrane_solver_amd import meshgen
    from membrane_solver_amd.device import DeviceMesh

    P, T = meshgen.icosphere(6)
    dm = DeviceMesh(P, T)
    dm.set_bending_params(np.ones(len(P)), np.zeros(len(P)))
    dm.set_params(modules=L.MS_MOD_SURFACE | L.MS_MOD_BENDING)
    e1, g1 = dm.energy_and_gradient()
    P2 = meshgen.smooth_displace(P, 0.2)
    dm.set_positions(P2)
    e2, _ = dm.energy_and_gradient()
    dm.set_positions(P)
    e3, g3 = dm.energy_and_gradient()
    assert e2.sum() != e1.sum()
    assert np.array_equal(e1, e3) and np.array_equal(g1, g3)  # bitwise reproducible
    dm.close()


@pytest.mark.parametrize("shape", ["sphere", "disk"])
def test_atomic_and_fixed_order_vertex_sums_agree(L, shape):
    """The default LDS-atomic accumulation and the fixed-order gather (ms_set_deterministic) are
    the same sums in a different order: energies/gradients agree to rounding (1e-12 relative here),
    the mode can be switched on a live context, and the fixed-order mode repeats bitwise."""
    from membrane_solver_amd import meshgen
    from membrane_solver_amd.device import DeviceMesh

    if shape == "sphere":
        P, T = meshgen.icosphere(24)
        P = meshgen.smooth_displace(P, 0.05)
    else:
        P, T, _ = meshgen.disk_patch(40)
        P = P + 0.02 * np.sin(3.0 * P[:, [1, 2, 0]] + 0.3)
    nv, nf = len(P), len(T)
    rng = np.random.default_rng(5)
    dm = DeviceMesh(P, T)
    dm.set_surface_tension(rng.uniform(0.5, 1.5, nf))
    dm.set_bending_params(rng.uniform(0.5, 1.5, nv), rng.uniform(-0.2, 0.2, nv))
    for model in (L.MS_BEND_HELFRICH, L.MS_BEND_WILLMORE):
        dm.set_params(modules=L.MS_MOD_SURFACE | L.MS_MOD_BENDING, bending_model=model)
        dm.set_deterministic(False)
        ea, ga = dm.energy_and_gradient()
        dm.set_deterministic(True)
        ed, gd = dm.energy_and_gradient()
        ed2, gd2 = dm.energy_and_gradient()
        assert np.array_equal(ed, ed2) and np.array_equal(gd, gd2)
        assert relerr(ea, ed) < 1e-12 and relerr(ga, gd) < 1e-12
        # a short CG run in either mode lands on the same trajectory to rounding
        out = []
        for det in (False, True):
            dm.set_positions(P)
            dm.reset_stepper()
            dm.set_deterministic(det)
            step, log = 1e-3, []
            for _ in range(5):
                r = dm.step(stepper=L.MS_STEPPER_CG, step_size=step, tol=1e-12)
                log.append((r.success, r.energy, r.grad_norm))
                step = r.next_step
            out.append(np.array(log, dtype=np.float64))
        assert np.array_equal(out[0][:, 0], out[1][:, 0])
        assert np.allclose(out[0][:, 1], out[1][:, 1], rtol=1e-11)
        assert np.allclose(out[0][:, 2], out[1][:, 2], rtol=1e-8)
    dm.close()


def test_angle_defects_and_gaussian_curvature(L):
    """ms_angle_defects against the reference's compute_angle_defects vectors (closed and open mesh), Gauss-Bonnet at a
    size with many tiles, and the gaussian_curvature module (a topological constant: offset, no force)."""
    from conftest import load_golden
    from membrane_solver_amd import meshgen
    from membrane_solver_amd.device import DeviceMesh
    from membrane_solver_amd.geometry.mesh import ArrayMesh
    from membrane_solver_amd.runtime.constraint_manager import ConstraintModuleManager
    from membrane_solver_amd.runtime.energy_manager import EnergyModuleManager
    from membrane_solver_amd.runtime.minimizer import Minimizer
    from membrane_solver_amd.runtime.steppers import GradientDescent

    g = load_golden("angle_defect_cases.npz")
    for name in ("ico5", "disk5"):
        dm = DeviceMesh(g[name + "_positions"], g[name + "_tri"], boundary=g[name + "_is_boundary"], tile_vertices=64)
        d = dm.angle_defects()
        assert np.max(np.abs(d - g[name + "_defects"])) <= 1e-12
        dm.close()
    P, T = meshgen.icosphere(40)
    P = meshgen.smooth_displace(P, 0.1)
    dm = DeviceMesh(P, T)
    assert abs(dm.angle_defects().sum() - 4.0 * np.pi) < 1e-9  # 2 pi chi, chi = 2
    dm.close()
    gp = {"surface_tension": 1.0, "gaussian_modulus": -0.7, "gaussian_curvature_check_defects": True}
    mods = ["surface", "gaussian_curvature"]
    mesh = ArrayMesh(g["ico5_positions"], g["ico5_tri"], global_parameters=gp, energy_modules=mods)
    mz = Minimizer(mesh, mesh.global_parameters, GradientDescent(), EnergyModuleManager(mods), ConstraintModuleManager([]),
                   quiet=True)
    bd = mz.compute_energy_breakdown()
    assert abs(bd["gaussian_curvature"] - float(g["ico5_gaussian_E"])) <= 1e-14
    E, grad = mz.compute_energy_and_gradient_array()
    assert abs(E - (bd["surface"] + bd["gaussian_curvature"])) <= 1e-12 * abs(E)
    res = mz.minimize(3)
    assert abs(res["energy"] - (mz.compute_energy())) <= 1e-12 * abs(res["energy"])
    module = EnergyModuleManager(mods).get_module("gaussian_curvature")
    from membrane_solver_amd.core.parameters import ParameterResolver
    E_mod = module.compute_energy_and_gradient_array(mesh, mesh.global_parameters, ParameterResolver(mesh.global_parameters),
                                                     positions=mesh.positions_view(), index_map=mesh.vertex_index_to_row,
                                                     grad_arr=np.zeros_like(P[:1]))
    assert abs(E_mod - float(g["ico5_gaussian_E"])) <= 1e-14
    open_mesh = ArrayMesh(g["disk5_positions"], g["disk5_tri"], global_parameters=gp, energy_modules=mods)
    mz2 = Minimizer(open_mesh, open_mesh.global_parameters, GradientDescent(), EnergyModuleManager(mods),
                    ConstraintModuleManager([]), quiet=True)
    # a surface WITH a boundary loop: kappa_bar times the Gauss-Bonnet invariant of the reference
    # (gaussian_curvature.py:128-143), through the Minimizer and through the plugin seam
    bd2 = mz2.compute_energy_breakdown()
    assert abs(bd2["gaussian_curvature"] - float(g["disk5_gaussian_E"])) <= 1e-12
    g_arr = np.zeros_like(g["disk5_positions"])
    E_open = module.compute_energy_and_gradient_array(open_mesh, open_mesh.global_parameters,
                                                      ParameterResolver(open_mesh.global_parameters),
                                                      positions=open_mesh.positions_view(),
                                                      index_map=open_mesh.vertex_index_to_row, grad_arr=g_arr)
    assert abs(E_open - float(g["disk5_gaussian_E"])) <= 1e-12 and not np.any(g_arr)
    strict = ArrayMesh(g["disk5_positions"], g["disk5_tri"], energy_modules=mods,
                       global_parameters=dict(gp, gaussian_curvature_strict_topology=True))
    mz3 = Minimizer(strict, strict.global_parameters, GradientDescent(), EnergyModuleManager(mods),
                    ConstraintModuleManager([]), quiet=True)
    with pytest.raises(L.MembraneHipError, match="strict_topology"):
        mz3.compute_energy()


def test_curvature_fields_match_reference(L):
    """ms_curvature_fields / geometry.curvature.compute_curvature_fields against the reference's
    compute_curvature_fields (geometry/curvature.py:404-448) on a closed noisy sphere and an open bulged disk, two
    tile sizes; the raw angle sums give the reference's Gauss-Bonnet invariant of the open surface."""
    from conftest import load_golden, relerr
    from membrane_solver_amd.device import DeviceMesh
    from membrane_solver_amd.geometry import curvature as gc
    from membrane_solver_amd.geometry.mesh import ArrayMesh

    g = load_golden("angle_defect_cases.npz")
    keys = (("mean_curvature_normal", 1e-11), ("mean_curvature", 1e-11), ("mixed_area", 1e-12),
            ("angle_defect", 1e-11), ("gaussian_curvature", 1e-10), ("principal_curvatures", 1e-10))
    for name in ("ico5", "disk5"):
        for tile in (64, 256):
            dm = DeviceMesh(g[name + "_positions"], g[name + "_tri"], boundary=g[name + "_is_boundary"],
                            tile_vertices=tile)
            f = dm.curvature_fields()
            for key, tol in keys:
                assert relerr(f[key], g[f"{name}_cf_{key}"]) < tol, (name, tile, key)
            dm.close()
        mesh = ArrayMesh(g[name + "_positions"], g[name + "_tri"])
        cf = gc.compute_curvature_fields(mesh, mesh.positions_view(), mesh.vertex_index_to_row)
        assert isinstance(cf, gc.CurvatureFields)
        for key, tol in keys:
            assert relerr(getattr(cf, key), g[f"{name}_cf_{key}"]) < tol, (name, key)
        assert np.max(np.abs(gc.compute_angle_defects(mesh, mesh.positions_view(), mesh.vertex_index_to_row)
                             - g[name + "_defects"])) <= 1e-12
    mesh = ArrayMesh(g["disk5_positions"], g["disk5_tri"])
    G, k_int, b_tot = gc.gauss_bonnet_invariant(mesh, mesh.positions_view())
    assert abs(G - float(g["disk5_gauss_bonnet_G"])) < 1e-12 and abs(G - 2.0 * np.pi) < 1e-12
    assert abs(k_int - float(g["disk5_gauss_bonnet_interior"])) < 1e-12
    assert abs(b_tot - float(g["disk5_gauss_bonnet_boundary"])) < 1e-12
