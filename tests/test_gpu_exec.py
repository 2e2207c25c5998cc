"""One-tile meshes: the library records its launches and runs them in ONE workgroup (k_exec).  Same device code, so
with fixed-order vertex sums every double of a trajectory must equal the launch-per-kernel path's (MS_EXEC=0)."""
from __future__ import annotations

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

LEAFLET = {
    "traj_config5_deck_gd.npz": "gd",
    "traj_disk5_gd_btl_backtrack.npz": "gd",
    "traj_disk5_gd_leaflet_consistent_backtrack.npz": "gd",
    "traj_disk6_cg_disktarget_coupled_gd.npz": "cg",
    "traj_disk6_gd_disktarget_nested_cg.npz": "gd",
    "traj_ico4_cg_btl_coupled_gd.npz": "cg",
    "traj_ico4_gd_leaflet_nested_cg.npz": "gd",
}


def _run_leaflet(fname, kind, observe):
    from test_gpu_leaflet import _leaflet_minimizer

    g = load_golden(fname)
    mesh, mz, log = _leaflet_minimizer(g, kind, observe=observe)
    if fname.startswith("traj_config5") or "disktarget" in fname:
        mesh.disk_rows_in = mesh.disk_rows_out = g["disk_rows"]
    res = mz.minimize(int(g["n_steps"]))
    _mir, dm = mz._device()
    stats = dm.exec_stats()
    return (np.array(log), mesh.positions_view().copy(), mesh.tilts_in_view().copy(), mesh.tilts_out_view().copy(),
            float(res["energy"]), stats)


@pytest.mark.parametrize("fname", sorted(LEAFLET))
@pytest.mark.parametrize("observe", [False, True])
def test_one_workgroup_interpreter_is_bitwise_the_launch_per_kernel_path_leaflets(fname, observe, deterministic, monkeypatch):
    monkeypatch.setenv("MS_EXEC", "0")
    ref = _run_leaflet(fname, LEAFLET[fname], observe)
    assert not ref[5]["active"] and ref[5]["packs"] == 0
    monkeypatch.setenv("MS_EXEC", "1")
    got = _run_leaflet(fname, LEAFLET[fname], observe)
    assert got[5]["active"] and got[5]["packs"] > 0
    # a pack carries several launches: that is the point
    assert got[5]["launches_recorded"] >= 3 * got[5]["packs"]
    assert np.array_equal(got[0], ref[0])
    for a, b in zip(got[1:4], ref[1:4]):
        assert np.array_equal(a, b)
    assert got[4] == ref[4]


SHAPE = ["traj_cube_gd.npz", "traj_disk5_gd_surface_bending_fixed.npz", "traj_ico4_gd_surface_tilt.npz",
         "traj_ico4_gd_tilt_volume_drift.npz", "traj_ico4_cg_bt_nested_gd.npz", "traj_disk5_gd_bt_coupled.npz",
         "traj_ico4_gd_bt_nested_cg.npz"]


def _run_shape(fname, in_library):
    from membrane_solver_amd.geometry.mesh import ArrayMesh
    from membrane_solver_amd.runtime.constraint_manager import ConstraintModuleManager
    from membrane_solver_amd.runtime.energy_manager import EnergyModuleManager
    from membrane_solver_amd.runtime.minimizer import Minimizer
    from membrane_solver_amd.runtime.steppers import ConjugateGradient, GradientDescent
    from test_gpu_bending_tilt import BT_TRAJ
    from test_gpu_minimizer import CASES, _build

    g = load_golden(fname)
    if fname in CASES:
        mods, cons, kind, gp = CASES[fname]
        gp = dict(gp)
        if "gp_volume_stiffness" in g:
            gp["volume_stiffness"] = float(g["gp_volume_stiffness"])
            gp["surface_tension"] = float(g["gp_surface_tension"])
        mesh = _build(g, mods, cons, gp, "target_volume" in g)
    else:
        kind, gp = BT_TRAJ[fname]
        mods, cons = ["surface", "tilt", "bending_tilt"], []
        mesh = ArrayMesh(g["positions0"], g["tri"], fixed=g["fixed"], surface_tension=g["gamma"], tilts=g["tilts0"],
                         tilt_fixed=g["tilt_fixed"], global_parameters=dict(gp), energy_modules=mods,
                         constraint_modules=[])
    stepper = GradientDescent() if kind == "gd" else ConjugateGradient()
    mz = Minimizer(mesh, mesh.global_parameters, stepper, EnergyModuleManager(mods), ConstraintModuleManager(cons),
                   quiet=True, step_size=float(g["step_size0"]))
    log = []
    if not in_library:
        orig = stepper.device_step

        def logged(dm, m, step_size, tol=0.0):
            r = orig(dm, m, step_size, tol=tol)
            log.append((float(r.success), r.next_step, r.energy, r.energy_eval, r.grad_norm, r.g_dot_d, r.alpha))
            return r

        stepper.device_step = logged
    res = mz.minimize(int(g["n_steps"]) + 5)
    _mir, dm = mz._device()
    tilts = mesh.tilts_view().copy() if "tilts0" in g else np.zeros(1)
    return np.array(log), mesh.positions_view().copy(), tilts, float(res["energy"]), dm.exec_stats()


@pytest.mark.parametrize("fname", SHAPE)
@pytest.mark.parametrize("in_library", [True, False])
def test_one_workgroup_interpreter_is_bitwise_the_launch_per_kernel_path_shape(fname, in_library, deterministic, monkeypatch):
    monkeypatch.setenv("MS_EXEC", "0")
    ref = _run_shape(fname, in_library)
    assert not ref[4]["active"]
    monkeypatch.setenv("MS_EXEC", "1")
    got = _run_shape(fname, in_library)
    assert got[4]["active"] and got[4]["packs"] > 0
    assert np.array_equal(got[0], ref[0])
    assert np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2])
    assert got[3] == ref[3]


def test_multi_tile_meshes_keep_the_launch_per_kernel_path():
    from membrane_solver_amd import meshgen
    from membrane_solver_amd.device import DeviceMesh

    P, T = meshgen.icosphere(8)  # 642 vertices: three tiles
    dm = DeviceMesh(P, T)
    st = dm.exec_stats()
    assert not st["active"] and not st["wanted"]
    dm.close()
    P, T = meshgen.icosphere(4)  # 162 vertices: one tile
    dm = DeviceMesh(P, T)
    assert dm.exec_stats()["active"]
    dm.profile_enable(True)     # per-kernel timing needs one launch per kernel
    assert not dm.exec_stats()["active"] and dm.exec_stats()["wanted"]
    dm.profile_enable(False)
    assert dm.exec_stats()["active"]
    dm.close()


@pytest.mark.parametrize("fname", ["traj_config5_deck_gd.npz", "traj_disk6_cg_disktarget_coupled_gd.npz",
                                   "traj_ico4_cg_btl_coupled_gd.npz", "traj_ico4_gd_leaflet_nested_cg.npz",
                                   "traj_disk5_gd_btl_backtrack.npz"])
def test_fused_leaflet_relaxation_agrees_with_the_generic_program(fname, monkeypatch):
    """Default (LDS-atomic) contexts run leaflet relaxations of one-tile meshes with the frozen geometry on the CU
    (csrc/ms_relax_fused.inc): the same operations as the recorded kernels, so the same iteration / evaluation counts and
    fields equal to rounding; whole trajectories agree like any two runs of the default mode do."""
    def run(fused):
        monkeypatch.setenv("MS_EXEC_FUSED", "1" if fused else "0")
        from test_gpu_leaflet import _leaflet_minimizer

        g = load_golden(fname)
        mesh, mz, _log = _leaflet_minimizer(g, LEAFLET.get(fname, "gd"), observe=False)
        if fname.startswith("traj_config5") or "disktarget" in fname:
            mesh.disk_rows_in = mesh.disk_rows_out = g["disk_rows"]
        _mir, dm = mz._device()
        rp = mz._tilt_relax_params()
        counts = None
        if rp is not None:
            counts = dm.relax_leaflet_tilts(solver=rp["solver"], max_iters=rp["max_iters"], step_size=rp["step_size"],
                                            tol=rp["tol"], jacobi=rp["jacobi"])
        t_in, t_out = dm.get_leaflet_tilts("in"), dm.get_leaflet_tilts("out")
        res = mz.minimize(int(g["n_steps"]))
        return counts, t_in, t_out, mesh.positions_view().copy(), float(res["energy"]), dm.exec_stats(), g

    c0, i0, o0, x0, e0, s0, g = run(False)
    c1, i1, o1, x1, e1, s1, _ = run(True)
    assert s0["relax_fused"] == 0
    if c0 is not None:
        assert s1["relax_fused"] > 0
        assert c1 == c0, "iteration / evaluation counts of the relaxation differ"
        scale = max(np.abs(i0).max(), np.abs(o0).max(), 1e-30)
        assert np.abs(i1 - i0).max() <= 1e-11 * scale and np.abs(o1 - o0).max() <= 1e-11 * scale
    assert np.abs(x1 - x0).max() <= 1e-8 * np.abs(x0).max()
    assert abs(e1 - e0) <= 1e-8 * abs(e0)
    # (no comparison with the fixture's final energy: the extra relaxation call above is not part of its trajectory;
    # tests/test_gpu_leaflet.py runs the fixtures themselves, through this path in the default mode)
