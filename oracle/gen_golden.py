#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself.

Run in the build container only (needs /root/reference; never on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python3 oracle/gen_golden.py [--fortran-dir DIR]

The fixtures are data only: seeded inputs and the reference's outputs.  If an
f2py build of the reference's Fortran kernels is on ``--fortran-dir`` (SURVEY
Appendix B recipe) the reference's loader picks it up and the mesh-level
outputs come from the Fortran-enabled path; ``meta_fortran`` in each file
records which path produced it.  The NumPy twins are always recorded too
where the reference's tests compare both (tests/test_fortran_kernels.py).
"""

from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")

ap = argparse.ArgumentParser()
ap.add_argument("--reference", default="/root/reference")
ap.add_argument("--fortran-dir", default="/tmp/fprobe/f2py_try")
ap.add_argument("--only-tilt", action="store_true")
ap.add_argument("--only-bt", action="store_true", help="bending_tilt + tilt relaxation vectors only")
ap.add_argument("--only-ts", action="store_true", help="tilt_smoothness vectors only")
ap.add_argument("--only-leaflet", action="store_true", help="two-leaflet tilt vectors only")
ap.add_argument("--only-btl", action="store_true", help="bending_tilt_in/out vectors only")
ap.add_argument("--only-disk", action="store_true", help="tilt_disk_target_in/out vectors only")
ap.add_argument("--only-defects", action="store_true", help="angle-defect vectors only")
ap.add_argument("--only-guard", action="store_true", help="guard / exhausted-search / volume-drift trajectories only")
ap.add_argument("--only-config5", action="store_true", help="the caveolin deck of BASELINE config 5 only")
ap.add_argument("--only-enforcer", action="store_true", help="line-search enforcer-lane trajectories only")
ap.add_argument("--only-precondition", action="store_true", help="ConjugateGradient(precondition=True) trajectories only")
args = ap.parse_args()

sys.dont_write_bytecode = True
if os.path.isdir(args.fortran_dir):
    sys.path.insert(0, args.fortran_dir)
sys.path.insert(0, args.reference)
sys.path.insert(0, ROOT)

from core.parameters.global_parameters import GlobalParameters  # noqa: E402
from core.parameters.resolver import ParameterResolver  # noqa: E402
from fortran_kernels import loader as fk_loader  # noqa: E402
from geometry.bending_derivatives import grad_cotan  # noqa: E402
from geometry.curvature import compute_curvature_data  # noqa: E402
from geometry.entities import Body, Edge, Facet, Mesh, Vertex  # noqa: E402
from geometry.geom_io import load_data, parse_geometry  # noqa: E402
from geometry.tilt_operators import p1_triangle_shape_gradients  # noqa: E402
from modules.constraints import volume as cvolume  # noqa: E402
from modules.energy import bending, surface, tilt  # noqa: E402
from modules.energy import volume as evolume  # noqa: E402
from modules.energy.bending_math import _apply_beltrami_laplacian  # noqa: E402
from modules.energy.bending_utils import _compute_effective_areas, _vertex_normals  # noqa: E402
from runtime.constraint_manager import ConstraintModuleManager  # noqa: E402
from runtime.energy_manager import EnergyModuleManager  # noqa: E402
from runtime.minimizer import Minimizer  # noqa: E402
from runtime.steppers.conjugate_gradient import ConjugateGradient  # noqa: E402
from runtime.steppers.gradient_descent import GradientDescent  # noqa: E402
from runtime.topology import get_min_edge_length  # noqa: E402

from membrane_solver_amd import meshgen  # noqa: E402

FORTRAN = {
    "surface": fk_loader.get_surface_energy_kernel() is not None,
    "grad_cotan": fk_loader.get_bending_grad_cotan_kernel() is not None,
    "laplacian": fk_loader.get_bending_laplacian_kernel() is not None,
    "curvature": fk_loader.get_tilt_curvature_kernel() is not None,
    "divergence": fk_loader.get_tilt_divergence_kernel() is not None,
}
print("fortran kernels:", FORTRAN)
META = np.array([f"{k}={int(v)}" for k, v in FORTRAN.items()])


def build_mesh(P, T, gp, fixed=None, tilts=None):
    """Programmatic mesh construction (pattern of tests/sample_meshes.py:266-295)."""
    m = Mesh()
    for i, p in enumerate(P):
        m.vertices[i] = Vertex(i, np.array(p, float))
        if fixed is not None and fixed[i]:
            m.vertices[i].fixed = True
        if tilts is not None:
            m.vertices[i].tilt = np.array(tilts[i], float)
    emap = {}
    nid = 1
    for fi, (a, b, c) in enumerate(T):
        se = []
        for u, v in ((a, b), (b, c), (c, a)):
            k = (u, v) if u < v else (v, u)
            e = emap.get(k)
            if e is None:
                e = nid
                emap[k] = e
                m.edges[e] = Edge(e, int(u), int(v))
                nid += 1
            se.append(e if m.edges[e].tail_index == u else -e)
        m.facets[fi] = Facet(fi, se, options={})
    m.global_parameters = GlobalParameters(dict(gp))
    m.build_connectivity_maps()
    m.build_facet_vertex_loops()
    return m


def add_body(m, target=None):
    b = Body(0, list(m.facets.keys()), target_volume=None)
    m.bodies[0] = b
    b.target_volume = float(b.compute_volume(m)) if target is None else float(target)
    return b


def mesh_arrays(m):
    pos = m.positions_view().copy()
    tri, _ = m.triangle_row_cache()
    tri = np.ascontiguousarray(tri, dtype=np.int32)
    nv = pos.shape[0]
    isb = np.zeros(nv, dtype=bool)
    for vid in m.boundary_vertex_ids:
        isb[m.vertex_index_to_row[vid]] = True
    return pos, tri, isb, m.fixed_mask.copy()


# ---------------------------------------------------------------------------
# (a) seeded kernel cases of tests/test_fortran_kernels.py
# ---------------------------------------------------------------------------
def gen_kernel_cases():
    out = {"meta_fortran": META}
    for n in (4, 17):
        rng = np.random.default_rng(123)
        u = rng.normal(size=(n, 3))
        v = rng.normal(size=(n, 3))
        v += 0.3 * rng.normal(size=(n, 3))
        gu, gv = grad_cotan(u, v)
        out[f"gc{n}_u"], out[f"gc{n}_v"], out[f"gc{n}_gu"], out[f"gc{n}_gv"] = u, v, gu, gv
    # degenerate pair (S <= 1e-15 -> zeros)
    u = np.array([[1.0, 0.0, 0.0], [1.0, 2.0, 3.0]])
    v = np.array([[2.0, 0.0, 0.0], [0.5, -1.0, 0.25]])
    gu, gv = grad_cotan(u, v)
    out["gcdeg_u"], out["gcdeg_v"], out["gcdeg_gu"], out["gcdeg_gv"] = u, v, gu, gv

    rng = np.random.default_rng(456)
    nv, nf = 9, 5
    tri = rng.integers(0, nv, size=(nf, 3), dtype=np.int32)
    weights = rng.normal(size=(nf, 3))
    fld = rng.normal(size=(nv, 3))
    out["lap_tri"], out["lap_weights"], out["lap_field"] = tri, weights, fld
    out["lap_out"] = np.asarray(_apply_beltrami_laplacian(weights, tri, fld))

    rng = np.random.default_rng(999)
    nv, nf = 10, 7
    pos = rng.normal(size=(nv, 3))
    tl = rng.normal(size=(nv, 3))
    tri = rng.integers(0, nv, size=(nf, 3), dtype=np.int32)
    area, g0, g1, g2 = p1_triangle_shape_gradients(positions=pos, tri_rows=tri)
    div = (np.einsum("ij,ij->i", tl[tri[:, 0]], g0) + np.einsum("ij,ij->i", tl[tri[:, 1]], g1)
           + np.einsum("ij,ij->i", tl[tri[:, 2]], g2))
    out["div_pos"], out["div_tilts"], out["div_tri"] = pos, tl, tri
    out["div_div"], out["div_area"], out["div_g0"], out["div_g1"], out["div_g2"] = div, area, g0, g1, g2

    # curvature data on random (possibly degenerate / repeated-index) triangles
    rng = np.random.default_rng(2024)
    nv, nf = 12, 9
    pos = rng.normal(size=(nv, 3))
    tri = rng.integers(0, nv, size=(nf, 3), dtype=np.int32)
    # NumPy twin (geometry/curvature.py:254-332) through a throwaway object
    # exposing exactly what compute_curvature_data reads.

    class _M:
        vertex_ids = np.arange(nv)
        _version = 0
        _curvature_version = -1
        _curvature_cache = {}
        _facet_loops_version = 0

        def triangle_row_cache(self):
            return tri, None

        def _geometry_cache_active(self, positions):
            return False

    saved = fk_loader._TILT_CURVATURE
    fk_loader._TILT_CURVATURE = False
    try:
        k, A, w, _ = compute_curvature_data(_M(), pos, {})
    finally:
        fk_loader._TILT_CURVATURE = saved
    out["curv_pos"], out["curv_tri"] = pos, tri
    out["curv_k"], out["curv_A"], out["curv_w"] = np.asarray(k), np.asarray(A), np.asarray(w)

    # surface kernel: right triangle gamma=2 -> E = 1 (tests/test_surface.py:61-93) + random soup
    pos = np.array([[0.0, 0, 0], [1, 0, 0], [0, 1, 0]])
    out["surf_rt_pos"], out["surf_rt_tri"], out["surf_rt_gamma"] = pos, np.array([[0, 1, 2]], np.int32), np.array([2.0])
    out["surf_rt_E"] = np.array(1.0)
    np.savez_compressed(os.path.join(OUT, "kernel_cases.npz"), **out)
    print("kernel_cases.npz written")


# ---------------------------------------------------------------------------
# (b) mesh-level energies / gradients
# ---------------------------------------------------------------------------
def eval_mesh_case(name, P, T, gamma_scale=None, tilts=None):
    gp = {"surface_tension": 1.3, "bending_modulus": 0.8, "bending_energy_model": "helfrich",
          "spontaneous_curvature": 0.0, "volume_constraint_mode": "lagrange",
          "volume_projection_during_minimization": False, "tilt_rigidity": 0.7}
    m = build_mesh(P, T, gp, tilts=tilts)
    if gamma_scale is not None:
        for fi, f in m.facets.items():
            f.options["surface_tension"] = float(gamma_scale[fi])
    add_body(m)
    pos, tri, isb, fixed = mesh_arrays(m)
    im = m.vertex_index_to_row
    pr = ParameterResolver(m.global_parameters)
    out = {"meta_fortran": META, "positions": pos, "tri": tri, "is_boundary": isb,
           "gamma": m.get_facet_parameter_array("surface_tension").copy(),
           "kappa": np.full(len(pos), 0.8), "min_edge": np.array(get_min_edge_length(m))}

    g = np.zeros_like(pos)
    out["E_surface"] = np.array(surface.compute_energy_and_gradient_array(
        m, m.global_parameters, pr, positions=pos, index_map=im, grad_arr=g))
    out["grad_surface"] = g.copy()

    m._curvature_cache = {}
    m._curvature_version = -1
    k, A, w, _ = compute_curvature_data(m, pos, im)
    out["k_vecs"], out["A_vor"], out["weights"] = np.array(k), np.array(A), np.array(w)
    Aeff, va0, va1, va2 = _compute_effective_areas(m, pos, tri, np.asarray(w), im)
    out["A_eff"] = np.array(Aeff)
    out["va_eff"] = np.stack([va0, va1, va2], axis=1)
    out["normals"] = _vertex_normals(m, pos, tri)

    for model in ("helfrich", "willmore"):
        for c0 in (0.0, 0.5):
            if model == "willmore" and c0 != 0.0:
                continue
            for mode in ("analytic", "approx"):
                m.global_parameters.set("bending_energy_model", model)
                m.global_parameters.set("spontaneous_curvature", c0)
                m.global_parameters.set("bending_gradient_mode", mode)
                if hasattr(m, "_bending_vertex_param_cache"):
                    m._bending_vertex_param_cache = None
                m._curvature_cache = {}
                m._curvature_version = -1
                g = np.zeros_like(pos)
                E = bending.compute_energy_and_gradient_array(
                    m, m.global_parameters, pr, positions=pos, index_map=im, grad_arr=g)
                tag = f"{model}_c{int(c0 * 10)}_{mode}"
                out[f"E_bend_{tag}"] = np.array(E)
                out[f"grad_bend_{tag}"] = g.copy()
                if mode == "analytic":
                    m._curvature_cache = {}
                    m._curvature_version = -1
                    ea = bending.compute_energy_array(m, m.global_parameters, pos, im)
                    out[f"Earr_bend_{model}_c{int(c0 * 10)}"] = np.asarray(ea)

    body = m.bodies[0]
    out["volume"] = np.array(body.compute_volume(m, positions=pos))
    gC = cvolume.constraint_gradients_array(m, m.global_parameters, positions=pos, index_map=im)[0]
    out["grad_volume"] = np.array(gC)
    # penalty mode energy/gradient (modules/energy/volume.py:94-128)
    m.global_parameters.set("volume_constraint_mode", "penalty")
    m.global_parameters.set("volume_stiffness", 50.0)
    body.target_volume = float(out["volume"]) * 0.9
    g = np.zeros_like(pos)
    out["E_volpen"] = np.array(evolume.compute_energy_and_gradient_array(
        m, m.global_parameters, pr, positions=pos, index_map=im, grad_arr=g))
    out["grad_volpen"] = g.copy()
    out["volpen_target"] = np.array(body.target_volume)
    out["volpen_k"] = np.array(50.0)

    if tilts is not None:
        tl = np.ascontiguousarray(m.tilts_view())
        g = np.zeros_like(pos)
        tg = np.zeros_like(pos)
        out["tilts"] = tl.copy()
        out["E_tilt"] = np.array(tilt.compute_energy_and_gradient_array(
            m, m.global_parameters, pr, positions=pos, index_map=im, grad_arr=g,
            tilts=tl, tilt_grad_arr=tg))
        out["grad_tilt_shape"], out["grad_tilt_tilt"] = g, tg
        out["k_tilt"] = np.array(0.7)
    np.savez_compressed(os.path.join(OUT, f"mesh_{name}.npz"), **out)
    print(f"mesh_{name}.npz written  nv={len(pos)} nf={len(tri)} boundary={int(isb.sum())} "
          f"obtuse={int((np.asarray(w) < 0).any(axis=1).sum())}")


def gen_mesh_cases():
    rng = np.random.default_rng(7)
    P, T = meshgen.icosphere(4)
    P = meshgen.smooth_displace(P, 0.08)
    eval_mesh_case("ico4", P, T, gamma_scale=1.0 + 0.2 * rng.random(len(T)),
                   tilts=0.1 * rng.normal(size=P.shape))
    P, T = meshgen.icosphere(8)
    P = meshgen.smooth_displace(P, 0.05)
    eval_mesh_case("ico8", P, T)
    P, T, _ = meshgen.disk_patch(5, jitter=0.28, seed=3)
    eval_mesh_case("disk5", P, T, tilts=0.1 * rng.normal(size=P.shape))
    # noisy sphere: many obtuse triangles on a closed surface
    P, T = meshgen.icosphere(5)
    P = P + 0.04 * np.random.default_rng(11).normal(size=P.shape)
    eval_mesh_case("ico5_noisy", P, T)


# ---------------------------------------------------------------------------
# (c) minimizer trajectories
# ---------------------------------------------------------------------------
def run_trajectory(name, m, stepper, n_steps, step_size=1e-3, mesh_path=False):
    em = EnergyModuleManager(m.energy_modules)
    cm = ConstraintModuleManager(m.constraint_modules)
    mz = Minimizer(m, m.global_parameters, stepper, em, cm, quiet=True, step_size=step_size)
    pos0, tri, isb, fixed = mesh_arrays(m)
    log = []
    orig_step = stepper.step

    # keep the stepper's signature: the minimizer passes ``trial_energy_fn`` (the array fast path of
    # the line search, line_search.py:357-421) only to steppers whose ``step`` names it
    # (minimizer.py:257-269); a (*a, **kw) wrapper would silently select the mesh-mutating path
    def logged_step(mesh, grad, step_size, energy_fn, constraint_enforcer=None, trial_energy_fn=None):
        r = orig_step(mesh, grad, step_size, energy_fn, constraint_enforcer=constraint_enforcer,
                      trial_energy_fn=trial_energy_fn)
        log.append((float(bool(r[0])), float(r[1]), float(r[2])))
        return r

    if mesh_path:  # tilt modules: see run_tilt_trajectory
        def logged_step(mesh, grad, step_size, energy_fn, constraint_enforcer=None):  # noqa: F811
            r = orig_step(mesh, grad, step_size, energy_fn, constraint_enforcer=constraint_enforcer)
            log.append((float(bool(r[0])), float(r[1]), float(r[2])))
            return r

    stepper.step = logged_step
    snaps = []

    def cb(mesh, i):
        snaps.append(mesh.positions_view().copy())

    E0, g0 = mz.compute_energy_and_gradient_array()
    res = mz.minimize(n_steps, callback=cb)
    out = {"meta_fortran": META, "positions0": pos0, "tri": tri, "is_boundary": isb, "fixed": fixed,
           "gamma": m.get_facet_parameter_array("surface_tension").copy(),
           "E0": np.array(E0), "grad0": np.array(g0),
           "positions_iter": np.array(snaps), "positions_final": m.positions_view().copy(),
           "step_log": np.array(log), "E_final": np.array(res["energy"]),
           "step_size_final": np.array(mz.step_size), "iterations": np.array(res["iterations"]),
           "n_steps": np.array(n_steps), "step_size0": np.array(step_size)}
    if m.bodies:
        out["target_volume"] = np.array(m.bodies[0].target_volume)
    return out


def gen_trajectories():
    # config 1: meshes/cube.json, g5 (surface + volume penalty, GD)
    data = load_data(os.path.join(args.reference, "meshes", "cube.json"))
    m = parse_geometry(data)
    out = run_trajectory("cube", m, GradientDescent(), 5, step_size=1e-3)
    out["energy_modules"] = np.array(list(m.energy_modules))
    out["constraint_modules"] = np.array(list(m.constraint_modules))
    gp = m.global_parameters
    out["gp_volume_stiffness"] = np.array(float(gp.get("volume_stiffness")))
    out["gp_volume_constraint_mode"] = np.array(str(gp.get("volume_constraint_mode")))
    out["gp_surface_tension"] = np.array(float(gp.get("surface_tension")))
    np.savez_compressed(os.path.join(OUT, "traj_cube_gd.npz"), **out)
    print("traj_cube_gd.npz  E_final=%.16g  log=%s" % (out["E_final"], out["step_log"].tolist()))

    base_gp = {"surface_tension": 1.0, "bending_modulus": 1.0, "bending_energy_model": "helfrich",
               "spontaneous_curvature": 0.0, "volume_constraint_mode": "lagrange",
               "volume_projection_during_minimization": False,
               "mesh_quality_auto_repair_enabled": False}
    P, T = meshgen.icosphere(8)
    P = meshgen.smooth_displace(P, 0.05)

    def fresh(mods, cons, extra=None):
        gp2 = dict(base_gp)
        gp2.update(extra or {})
        mm = build_mesh(P, T, gp2)
        add_body(mm)
        mm.energy_modules = list(mods)
        mm.constraint_modules = list(cons)
        return mm

    # config 2 shape: surface + volume (lagrange) with GD
    mm = fresh(["surface"], ["volume"])
    out = run_trajectory("ico8_gd", mm, GradientDescent(), 6, step_size=1e-3)
    np.savez_compressed(os.path.join(OUT, "traj_ico8_gd_surface_volume.npz"), **out)
    print("traj_ico8_gd_surface_volume.npz E_final=%.16g" % out["E_final"], out["step_log"][:, 0])

    # config 3 shape: surface + bending with CG (no volume constraint -> fast line-search path)
    mm = fresh(["surface", "bending"], [])
    out = run_trajectory("ico8_cg", mm, ConjugateGradient(), 12, step_size=1e-3)
    out["kappa"] = np.array(1.0)
    np.savez_compressed(os.path.join(OUT, "traj_ico8_cg_surface_bending.npz"), **out)
    print("traj_ico8_cg_surface_bending.npz E_final=%.16g" % out["E_final"], out["step_log"][:, 0])

    # surface + bending + volume constraint, CG (the BASELINE.md timing configuration)
    mm = fresh(["surface", "bending"], ["volume"], {"spontaneous_curvature": 0.3})
    out = run_trajectory("ico8_cg_vol", mm, ConjugateGradient(), 8, step_size=1e-3)
    out["kappa"] = np.array(1.0)
    out["c0"] = np.array(0.3)
    np.savez_compressed(os.path.join(OUT, "traj_ico8_cg_surface_bending_volume.npz"), **out)
    print("traj_ico8_cg_surface_bending_volume.npz E_final=%.16g" % out["E_final"], out["step_log"][:, 0])

    # open mesh with fixed boundary ring: surface + bending, GD (boundary + fixed rows)
    Pd, Td, Bd = meshgen.disk_patch(5, jitter=0.15, seed=5)
    gp2 = dict(base_gp)
    mm = build_mesh(Pd, Td, gp2, fixed=Bd)
    mm.energy_modules = ["surface", "bending"]
    mm.constraint_modules = []
    out = run_trajectory("disk5_gd", mm, GradientDescent(), 6, step_size=1e-3)
    out["kappa"] = np.array(1.0)
    np.savez_compressed(os.path.join(OUT, "traj_disk5_gd_surface_bending_fixed.npz"), **out)
    print("traj_disk5_gd_surface_bending_fixed.npz E_final=%.16g" % out["E_final"], out["step_log"][:, 0])


def gen_tilt_trajectory():
    """surface + tilt (vertex tilts fixed, re-projected to the tangent plane each step), GD."""
    rng = np.random.default_rng(21)
    P, T = meshgen.icosphere(4)
    P = meshgen.smooth_displace(P, 0.08)
    tl = 0.3 * rng.normal(size=P.shape)
    gp = {"surface_tension": 1.0, "tilt_rigidity": 2.5, "volume_constraint_mode": "lagrange",
          "volume_projection_during_minimization": False, "mesh_quality_auto_repair_enabled": False}
    mm = build_mesh(P, T, gp, tilts=tl)
    mm.energy_modules = ["surface", "tilt"]
    mm.constraint_modules = []
    out = run_trajectory("ico4_gd_tilt", mm, GradientDescent(), 6, step_size=2e-3, mesh_path=True)
    out["tilts0"] = tl
    out["tilts_final"] = np.ascontiguousarray(mm.tilts_view())
    out["k_tilt"] = np.array(2.5)
    np.savez_compressed(os.path.join(OUT, "traj_ico4_gd_surface_tilt.npz"), **out)
    print("traj_ico4_gd_surface_tilt.npz E_final=%.16g" % out["E_final"], out["step_log"][:, 0])


# ---------------------------------------------------------------------------
# (d) bending_tilt (modules/energy/bending_tilt.py) and nested tilt relaxation
#     (runtime/steppers/tilt_relaxation.py:237-424), single tilt field
# ---------------------------------------------------------------------------
def _tangent_tilts(m, rng, scale):
    pos = m.positions_view()
    tl = scale * rng.normal(size=pos.shape)
    nrm = m.vertex_normals(pos)
    return tl - np.einsum("ij,ij->i", tl, nrm)[:, None] * nrm


def gen_bending_tilt_cases():
    from modules.energy import bending_tilt
    from runtime.preconditioners import build_tilt_cg_preconditioner

    out = {"meta_fortran": META}
    rng = np.random.default_rng(33)
    meshes = {}
    P, T = meshgen.icosphere(5)
    P = meshgen.smooth_displace(P, 0.08) + 4e-3 * rng.normal(size=P.shape)
    meshes["ico5"] = (P, T)
    Pd, Td, _isb = meshgen.disk_patch(5, bulge=0.35, jitter=0.03, seed=5)
    meshes["disk5"] = (Pd, Td)
    for name, (P, T) in meshes.items():
        for mode in ("analytic", "approx"):
            gp = {"surface_tension": 1.0, "bending_modulus": 1.3, "spontaneous_curvature": 0.2,
                  "bending_energy_model": "helfrich", "bending_gradient_mode": mode, "tilt_rigidity": 2.0}
            m = build_mesh(P, T, gp)
            pos, tri, isb, fixed = mesh_arrays(m)
            tl = _tangent_tilts(m, np.random.default_rng(7), 0.25)
            res = ParameterResolver(m.global_parameters)
            g = np.zeros_like(pos)
            tg = np.zeros_like(pos)
            E = bending_tilt.compute_energy_and_gradient_array(
                m, m.global_parameters, res, positions=pos, index_map=m.vertex_index_to_row,
                grad_arr=g, tilts=tl, tilt_grad_arr=tg)
            tg2 = np.zeros_like(pos)
            E2 = bending_tilt.compute_energy_and_gradient_array(
                m, m.global_parameters, res, positions=pos, index_map=m.vertex_index_to_row,
                grad_arr=None, tilts=tl, tilt_grad_arr=tg2)
            k = f"{name}_{mode}"
            out[k + "_E"] = np.array(E)
            out[k + "_E_tiltonly"] = np.array(E2)
            out[k + "_grad"] = g
            out[k + "_tilt_grad"] = tg
            out[k + "_tilt_grad_tiltonly"] = tg2
            if mode == "analytic":
                out[name + "_positions"] = pos
                out[name + "_tri"] = tri
                out[name + "_is_boundary"] = isb
                out[name + "_tilts"] = tl
                em = EnergyModuleManager(["tilt", "bending_tilt"])
                mz = Minimizer(m, m.global_parameters, GradientDescent(), em, ConstraintModuleManager([]),
                               quiet=True)
                M_inv = build_tilt_cg_preconditioner(
                    m, mz.param_resolver, mz.energy_context(), positions=pos,
                    index_map=m.vertex_index_to_row, fixed_mask=np.zeros(len(pos), bool))
                out[name + "_jacobi_Minv"] = M_inv
            print(k, "E=%.16g" % E)
    np.savez_compressed(os.path.join(OUT, "bending_tilt_cases.npz"), **out)


def run_tilt_trajectory(fname, P, T, gp, mods, stepper, n_steps, step_size, tilt_scale=0.3, seed=21,
                        tilt_fixed_every=0):
    mm = build_mesh(P, T, gp)
    rng = np.random.default_rng(seed)
    tl = _tangent_tilts(mm, rng, tilt_scale)
    mm.set_tilts_from_array(tl)
    tfix = np.zeros(len(P), bool)
    if tilt_fixed_every:
        tfix[::tilt_fixed_every] = True
        for i in np.flatnonzero(tfix):
            mm.vertices[int(i)].tilt_fixed = True
    mm.energy_modules = list(mods)
    mm.constraint_modules = []
    tilt_snaps = []
    em = EnergyModuleManager(mm.energy_modules)
    cm = ConstraintModuleManager(mm.constraint_modules)
    mz = Minimizer(mm, mm.global_parameters, stepper, em, cm, quiet=True, step_size=step_size)
    pos0, tri, isb, fixed = mesh_arrays(mm)
    log = []
    orig_step = stepper.step

    # DELIBERATELY without ``trial_energy_fn`` in the signature: the minimizer then does not pass
    # the array trial-energy callback (minimizer.py:257-269) and the line search takes its
    # mesh-mutating path (line_search.py:428-487), which evaluates every trial consistently.  The
    # array fast path (:357-421) evaluates bending_tilt through the EnergyContext's P1-gradient
    # cache, which is keyed to the mesh and therefore STALE for a trial position array (measured
    # here: bending_tilt 40.5908 via the manager vs 40.7643 direct at the same trial point), so it
    # is not a usable oracle for the tilt modules.  For the shape-only modules both paths give
    # identical numbers (run_trajectory above keeps the fast path).
    def logged_step(mesh, grad, step_size, energy_fn, constraint_enforcer=None):
        r = orig_step(mesh, grad, step_size, energy_fn, constraint_enforcer=constraint_enforcer)
        log.append((float(bool(r[0])), float(r[1]), float(r[2])))
        return r

    stepper.step = logged_step
    snaps = []

    def cb(mesh, i):
        snaps.append(mesh.positions_view().copy())
        tilt_snaps.append(np.ascontiguousarray(mesh.tilts_view()).copy())

    E0, g0 = mz.compute_energy_and_gradient_array()
    res = mz.minimize(n_steps, callback=cb)
    out = {"meta_fortran": META, "positions0": pos0, "tri": tri, "is_boundary": isb, "fixed": fixed,
           "tilt_fixed": tfix, "gamma": mm.get_facet_parameter_array("surface_tension").copy(),
           "E0": np.array(E0), "grad0": np.array(g0), "tilts0": tl,
           "positions_iter": np.array(snaps), "tilts_iter": np.array(tilt_snaps),
           "positions_final": mm.positions_view().copy(),
           "tilts_final": np.ascontiguousarray(mm.tilts_view()).copy(),
           "step_log": np.array(log), "E_final": np.array(res["energy"]),
           "n_steps": np.array(n_steps), "step_size0": np.array(step_size)}
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(fname, "E_final=%.16g" % out["E_final"], out["step_log"][:, 0])


def gen_bending_tilt_trajectories():
    base = {"surface_tension": 1.0, "bending_modulus": 1.0, "spontaneous_curvature": 0.1,
            "bending_energy_model": "helfrich", "bending_gradient_mode": "analytic",
            "tilt_rigidity": 2.5, "volume_constraint_mode": "lagrange",
            "volume_projection_during_minimization": False, "mesh_quality_auto_repair_enabled": False}
    P, T = meshgen.icosphere(4)
    P = meshgen.smooth_displace(P, 0.08)
    # tilts fixed (only re-projected), shape steps feel the tilt-splay coupling
    run_tilt_trajectory("traj_ico4_gd_bt_fixed.npz", P, T, dict(base, tilt_solve_mode="fixed"),
                        ["surface", "tilt", "bending_tilt"], GradientDescent(), 5, 1e-3)
    # nested tilt relaxation, preconditioned CG inner solve
    run_tilt_trajectory("traj_ico4_gd_bt_nested_cg.npz", P, T,
                        dict(base, tilt_solve_mode="nested", tilt_solver="cg", tilt_step_size=0.1,
                             tilt_inner_steps=6, tilt_tol=1e-10),
                        ["surface", "tilt", "bending_tilt"], GradientDescent(), 5, 1e-3)
    # nested, gradient-descent inner solve, CG shape stepper, some tilts pinned
    run_tilt_trajectory("traj_ico4_cg_bt_nested_gd.npz", P, T,
                        dict(base, tilt_solve_mode="nested", tilt_solver="gd", tilt_step_size=0.08,
                             tilt_inner_steps=4),
                        ["surface", "tilt", "bending_tilt"], ConjugateGradient(), 6, 1e-3, tilt_fixed_every=9)
    # open patch with boundary: coupled mode, CG inner solve without preconditioner
    Pd, Td, isb = meshgen.disk_patch(5, bulge=0.35, jitter=0.02, seed=5)
    gpd = dict(base, tilt_solve_mode="coupled", tilt_solver="cg", tilt_cg_preconditioner="none",
               tilt_step_size=0.1, tilt_coupled_steps=3, bending_modulus=0.8)
    # boundary vertices are held (fixed positions), like the reference's open-patch decks
    run_tilt_trajectory_disk("traj_disk5_gd_bt_coupled.npz", Pd, Td, gpd, np.asarray(isb, bool).copy())


def run_tilt_trajectory_disk(fname, P, T, gp, fixed):
    """Same as run_tilt_trajectory with fixed boundary positions."""
    global build_mesh
    orig = build_mesh

    def bm(P_, T_, gp_, fixed=None, tilts=None):
        return orig(P_, T_, gp_, fixed=fixed_mask, tilts=tilts)

    fixed_mask = fixed
    build_mesh = bm
    try:
        run_tilt_trajectory(fname, P, T, gp, ["surface", "tilt", "bending_tilt"], GradientDescent(), 5, 1e-3,
                            tilt_scale=0.2, seed=4)
    finally:
        build_mesh = orig


# ---------------------------------------------------------------------------
# (e) tilt_smoothness (modules/energy/tilt_smoothness.py, ambient_v1 transport)
# ---------------------------------------------------------------------------
def gen_tilt_smoothness():
    from modules.energy import tilt_smoothness
    from runtime.preconditioners import build_tilt_cg_preconditioner

    out = {"meta_fortran": META}
    meshes = {}
    rng = np.random.default_rng(41)
    P, T = meshgen.icosphere(5)
    P = meshgen.smooth_displace(P, 0.08) + 4e-3 * rng.normal(size=P.shape)
    meshes["ico5"] = (P, T)
    Pd, Td, _isb = meshgen.disk_patch(5, bulge=0.35, jitter=0.03, seed=5)
    meshes["disk5"] = (Pd, Td)
    for name, (P, T) in meshes.items():
        gp = {"surface_tension": 1.0, "tilt_rigidity": 2.0, "tilt_smoothness_rigidity": 0.7}
        m = build_mesh(P, T, gp)
        pos, tri, isb, fixed = mesh_arrays(m)
        tl = _tangent_tilts(m, np.random.default_rng(9), 0.25)
        res = ParameterResolver(m.global_parameters)
        g = np.zeros_like(pos)
        tg = np.zeros_like(pos)
        E = tilt_smoothness.compute_energy_and_gradient_array(
            m, m.global_parameters, res, positions=pos, index_map=m.vertex_index_to_row, grad_arr=g,
            tilts=tl, tilt_grad_arr=tg)
        out[name + "_positions"], out[name + "_tri"], out[name + "_is_boundary"] = pos, tri, isb
        out[name + "_tilts"], out[name + "_E"], out[name + "_grad"], out[name + "_tilt_grad"] = tl, np.array(E), g, tg
        em = EnergyModuleManager(["tilt", "tilt_smoothness"])
        mz = Minimizer(m, m.global_parameters, GradientDescent(), em, ConstraintModuleManager([]), quiet=True)
        out[name + "_jacobi_Minv"] = build_tilt_cg_preconditioner(
            m, mz.param_resolver, mz.energy_context(), positions=pos, index_map=m.vertex_index_to_row,
            fixed_mask=np.zeros(len(pos), bool))
        print("tilt_smoothness", name, "E=%.16g" % E, "shape grad max", np.abs(g).max())
    np.savez_compressed(os.path.join(OUT, "tilt_smoothness_cases.npz"), **out)
    base = {"surface_tension": 1.0, "bending_modulus": 1.0, "spontaneous_curvature": 0.1,
            "bending_energy_model": "helfrich", "bending_gradient_mode": "analytic",
            "tilt_rigidity": 2.5, "tilt_smoothness_rigidity": 0.6, "volume_constraint_mode": "lagrange",
            "volume_projection_during_minimization": False, "mesh_quality_auto_repair_enabled": False}
    P, T = meshgen.icosphere(4)
    P = meshgen.smooth_displace(P, 0.08)
    run_tilt_trajectory("traj_ico4_gd_ts_nested_cg.npz", P, T,
                        dict(base, tilt_solve_mode="nested", tilt_solver="cg", tilt_step_size=0.1,
                             tilt_inner_steps=6),
                        ["surface", "tilt", "tilt_smoothness", "bending_tilt"], GradientDescent(), 5, 1e-3)
    run_tilt_trajectory("traj_ico4_cg_ts_fixed.npz", P, T, dict(base, tilt_solve_mode="fixed"),
                        ["surface", "tilt", "tilt_smoothness"], ConjugateGradient(), 6, 2e-3)
    # over-long first step: rejected trials.  ConjugateGradient has no trial_energy_fn, so its line
    # search mutates the mesh per trial and the stored tilts keep the projection of a rejected trial
    run_tilt_trajectory("traj_ico4_cg_ts_backtrack.npz", P, T, dict(base, tilt_solve_mode="fixed"),
                        ["surface", "tilt", "tilt_smoothness", "bending_tilt"], ConjugateGradient(), 5, 8e-2)
    run_tilt_trajectory("traj_ico4_gd_ts_backtrack.npz", P, T, dict(base, tilt_solve_mode="fixed"),
                        ["surface", "tilt", "tilt_smoothness", "bending_tilt"], GradientDescent(), 5, 8e-2)


# ---------------------------------------------------------------------------
# two-leaflet tilt fields: tilt_in / tilt_out (lumped + consistent mass), tilt_smoothness_in / _out,
# the leaflet Jacobi preconditioner and relax_leaflet_tilts inside Minimizer.minimize
# ---------------------------------------------------------------------------
def _set_leaflet_fields(m, seed, scale, fixed_in_every=0, fixed_out_every=0):
    rng = np.random.default_rng(seed)
    tin = _tangent_tilts(m, rng, scale)
    tout = _tangent_tilts(m, rng, 0.8 * scale)
    m.set_tilts_in_from_array(tin)
    m.set_tilts_out_from_array(tout)
    nv = len(m.vertex_ids)
    fin, fout = np.zeros(nv, bool), np.zeros(nv, bool)
    if fixed_in_every:
        fin[::fixed_in_every] = True
        for i in np.flatnonzero(fin):
            m.vertices[int(i)].tilt_fixed_in = True
    if fixed_out_every:
        fout[1::fixed_out_every] = True
        for i in np.flatnonzero(fout):
            m.vertices[int(i)].tilt_fixed_out = True
    return tin, tout, fin, fout


def run_leaflet_trajectory(fname, P, T, gp, mods, stepper, n_steps, step_size, tilt_scale=0.3, seed=33,
                           fixed_in_every=0, fixed_out_every=0):
    mm = build_mesh(P, T, gp)
    disk_rows = _tag_disk(mm) if _DISK_TAG else np.zeros(0, dtype=int)
    tin, tout, fin, fout = _set_leaflet_fields(mm, seed, tilt_scale, fixed_in_every, fixed_out_every)
    mm.energy_modules = list(mods)
    mm.constraint_modules = []
    em = EnergyModuleManager(mm.energy_modules)
    cm = ConstraintModuleManager(mm.constraint_modules)
    mz = Minimizer(mm, mm.global_parameters, stepper, em, cm, quiet=True, step_size=step_size)
    pos0, tri, isb, fixed = mesh_arrays(mm)
    log = []
    orig_step = stepper.step

    # without ``trial_energy_fn``: the mesh-mutating line search, as for the single field
    # (run_tilt_trajectory) -- one protocol for every tilt family on the device
    def logged_step(mesh, grad, step_size, energy_fn, constraint_enforcer=None):
        r = orig_step(mesh, grad, step_size, energy_fn, constraint_enforcer=constraint_enforcer)
        log.append((float(bool(r[0])), float(r[1]), float(r[2])))
        return r

    stepper.step = logged_step
    snaps, tin_snaps, tout_snaps = [], [], []

    def cb(mesh, i):
        snaps.append(mesh.positions_view().copy())
        tin_snaps.append(np.ascontiguousarray(mesh.tilts_in_view()).copy())
        tout_snaps.append(np.ascontiguousarray(mesh.tilts_out_view()).copy())

    E0, g0 = mz.compute_energy_and_gradient_array()
    res = mz.minimize(n_steps, callback=cb)
    out = {"meta_fortran": META, "positions0": pos0, "tri": tri, "is_boundary": isb, "fixed": fixed,
           "tilt_fixed_in": fin, "tilt_fixed_out": fout,
           "gamma": mm.get_facet_parameter_array("surface_tension").copy(),
           "E0": np.array(E0), "grad0": np.array(g0), "tilts_in0": tin, "tilts_out0": tout,
           "positions_iter": np.array(snaps), "tilts_in_iter": np.array(tin_snaps),
           "tilts_out_iter": np.array(tout_snaps), "positions_final": mm.positions_view().copy(),
           "tilts_in_final": np.ascontiguousarray(mm.tilts_in_view()).copy(),
           "tilts_out_final": np.ascontiguousarray(mm.tilts_out_view()).copy(),
           "step_log": np.array(log), "E_final": np.array(res["energy"]),
           "n_steps": np.array(n_steps), "step_size0": np.array(step_size),
           "gp_json": np.array(json.dumps(gp, sort_keys=True)), "modules": np.array(list(mods)),
           "disk_rows": disk_rows}
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(fname, "E_final=%.16g" % out["E_final"], out["step_log"][:, 0])


def gen_leaflet():
    import importlib

    from runtime.preconditioners import build_leaflet_tilt_cg_preconditioner

    out = {"meta_fortran": META}
    rng = np.random.default_rng(43)
    P, T = meshgen.icosphere(5)
    P = meshgen.smooth_displace(P, 0.08) + 4e-3 * rng.normal(size=P.shape)
    Pd, Td, _isb = meshgen.disk_patch(5, bulge=0.35, jitter=0.03, seed=5)
    for name, (P_, T_) in {"ico5": (P, T), "disk5": (Pd, Td)}.items():
        for mass in ("lumped", "consistent"):
            gp = {"surface_tension": 1.0, "tilt_modulus_in": 1.7, "tilt_modulus_out": 2.3,
                  "tilt_mass_mode_in": mass, "tilt_mass_mode": "lumped" if mass == "consistent" else "consistent",
                  "bending_modulus": 0.8, "bending_modulus_out": 0.5}
            m = build_mesh(P_, T_, gp)
            tin, tout, _fi, _fo = _set_leaflet_fields(m, 12, 0.25)
            pos, tri, isb, fixed = mesh_arrays(m)
            res = ParameterResolver(m.global_parameters)
            key = f"{name}_{mass}"
            if mass == "lumped":
                out[name + "_positions"], out[name + "_tri"], out[name + "_is_boundary"] = pos, tri, isb
                out[name + "_tilts_in"], out[name + "_tilts_out"] = tin, tout
            out[key + "_gp_json"] = np.array(json.dumps(gp, sort_keys=True))
            for mod in ("tilt_in", "tilt_out", "tilt_smoothness_in", "tilt_smoothness_out"):
                module = importlib.import_module(f"modules.energy.{mod}")
                g = np.zeros_like(pos)
                tgi, tgo = np.zeros_like(pos), np.zeros_like(pos)
                E = module.compute_energy_and_gradient_array(
                    m, m.global_parameters, res, positions=pos, index_map=m.vertex_index_to_row, grad_arr=g,
                    tilts_in=tin, tilts_out=tout, tilt_in_grad_arr=tgi, tilt_out_grad_arr=tgo)
                out[f"{key}_{mod}_E"], out[f"{key}_{mod}_grad"] = np.array(E), g
                out[f"{key}_{mod}_tilt_grad"] = tgi if mod.endswith("_in") else tgo
                assert not np.any(tgo if mod.endswith("_in") else tgi)
                print("leaflet", key, mod, "E=%.16g" % E, "shape grad max", np.abs(g).max())
            mods = ["tilt_in", "tilt_out", "tilt_smoothness_in", "tilt_smoothness_out"]
            m.energy_modules = list(mods)
            m.constraint_modules = []
            mz = Minimizer(m, m.global_parameters, GradientDescent(), EnergyModuleManager(mods),
                           ConstraintModuleManager([]), quiet=True)
            va = m.barycentric_vertex_areas(positions=pos)
            fin = np.zeros(len(pos), bool)
            fout = np.zeros(len(pos), bool)
            fin[::7] = True
            Mi, Mo = build_leaflet_tilt_cg_preconditioner(
                m, mz.param_resolver, mz.energy_context(), positions=pos, index_map=m.vertex_index_to_row,
                fixed_mask_in=fin, fixed_mask_out=fout, tilt_vertex_areas_in=va, tilt_vertex_areas_out=va)
            out[key + "_jacobi_Minv_in"], out[key + "_jacobi_Minv_out"], out[key + "_jacobi_fixed_in"] = Mi, Mo, fin
            # the relaxation's evaluation (vertex-area fast path for the magnitude modules)
            tgi, tgo = np.zeros_like(pos), np.zeros_like(pos)
            E_rel = mz._compute_energy_and_leaflet_tilt_gradients_array(
                positions=pos, tilts_in=tin, tilts_out=tout, tilt_in_grad_arr=tgi, tilt_out_grad_arr=tgo,
                tilt_vertex_areas_in=va, tilt_vertex_areas_out=va, tilt_only=True)
            out[key + "_relax_E"], out[key + "_relax_grad_in"], out[key + "_relax_grad_out"] = np.array(E_rel), tgi, tgo
    np.savez_compressed(os.path.join(OUT, "tilt_leaflet_cases.npz"), **out)

    base = {"surface_tension": 1.0, "tilt_modulus_in": 2.0, "tilt_modulus_out": 1.4, "bending_modulus": 0.6,
            "bending_modulus_in": 0.9, "volume_constraint_mode": "lagrange",
            "volume_projection_during_minimization": False, "mesh_quality_auto_repair_enabled": False}
    allm = ["surface", "tilt_in", "tilt_out", "tilt_smoothness_in", "tilt_smoothness_out"]
    P4, T4 = meshgen.icosphere(4)
    P4 = meshgen.smooth_displace(P4, 0.08)
    run_leaflet_trajectory("traj_ico4_gd_leaflet_nested_cg.npz", P4, T4,
                           dict(base, tilt_solve_mode="nested", tilt_solver="cg", tilt_step_size=0.1,
                                tilt_inner_steps=6), allm, GradientDescent(), 5, 1e-3, fixed_in_every=9)
    run_leaflet_trajectory("traj_ico4_cg_leaflet_coupled_gd.npz", P4, T4,
                           dict(base, tilt_solve_mode="coupled", tilt_solver="gd", tilt_step_size=0.05,
                                tilt_coupled_steps=4, tilt_tol=1e-9), allm, ConjugateGradient(), 6, 2e-3,
                           fixed_out_every=11)
    run_leaflet_trajectory("traj_ico4_cg_leaflet_plaincg.npz", P4, T4,
                           dict(base, tilt_solve_mode="nested", tilt_solver="cg", tilt_cg_preconditioner="none",
                                tilt_step_size=0.08, tilt_inner_steps=5, tilt_cg_max_iters=4), allm,
                           ConjugateGradient(), 5, 2e-3)
    # consistent mass in the outer energy / shape gradient, tilts fixed, over-long first step (rejections)
    Pd5, Td5, _ = meshgen.disk_patch(5, bulge=0.35, jitter=0.02, seed=7)
    m0 = build_mesh(Pd5, Td5, base)
    fixed = mesh_arrays(m0)[2].copy()  # boundary rows clamped
    run_leaflet_trajectory_fixed("traj_disk5_gd_leaflet_consistent_backtrack.npz", Pd5, Td5,
                                 dict(base, tilt_solve_mode="fixed", tilt_mass_mode="consistent"),
                                 ["surface", "tilt_in", "tilt_out", "tilt_smoothness_out"], GradientDescent(), 5, 8e-2,
                                 fixed)



def gen_bending_tilt_leaflet():
    """bending_tilt_in / bending_tilt_out (bending_tilt_leaflet.py:231-758, default options)."""
    import importlib

    out = {"meta_fortran": META}
    rng = np.random.default_rng(47)
    P, T = meshgen.icosphere(5)
    P = meshgen.smooth_displace(P, 0.08) + 4e-3 * rng.normal(size=P.shape)
    Pd, Td, _isb = meshgen.disk_patch(5, bulge=0.35, jitter=0.03, seed=5)
    gp = {"surface_tension": 1.0, "bending_modulus": 0.9, "bending_modulus_in": 1.3, "spontaneous_curvature": 0.1,
          "spontaneous_curvature_out": -0.2, "bending_gradient_mode": "analytic"}
    out["gp_json"] = np.array(json.dumps(gp, sort_keys=True))
    for name, (P_, T_) in {"ico5": (P, T), "disk5": (Pd, Td)}.items():
        m = build_mesh(P_, T_, gp)
        tin, tout, _fi, _fo = _set_leaflet_fields(m, 12, 0.25)
        pos, tri, isb, fixed = mesh_arrays(m)
        res = ParameterResolver(m.global_parameters)
        out[name + "_positions"], out[name + "_tri"], out[name + "_is_boundary"] = pos, tri, isb
        out[name + "_tilts_in"], out[name + "_tilts_out"] = tin, tout
        for mod in ("bending_tilt_in", "bending_tilt_out"):
            module = importlib.import_module(f"modules.energy.{mod}")
            g = np.zeros_like(pos)
            tgi, tgo = np.zeros_like(pos), np.zeros_like(pos)
            E = module.compute_energy_and_gradient_array(
                m, m.global_parameters, res, positions=pos, index_map=m.vertex_index_to_row, grad_arr=g,
                tilts_in=tin, tilts_out=tout, tilt_in_grad_arr=tgi, tilt_out_grad_arr=tgo)
            E_only = module.compute_energy_and_gradient_array(
                m, m.global_parameters, res, positions=pos, index_map=m.vertex_index_to_row, grad_arr=None,
                tilts_in=tin, tilts_out=tout, tilt_in_grad_arr=None, tilt_out_grad_arr=None)
            assert abs(E_only - E) <= 1e-13 * abs(E)
            out[f"{name}_{mod}_E"], out[f"{name}_{mod}_grad"] = np.array(E), g
            out[f"{name}_{mod}_tilt_grad"] = tgi if mod.endswith("_in") else tgo
            print("bending_tilt leaflet", name, mod, "E=%.16g" % E, "shape grad max", np.abs(g).max())
    np.savez_compressed(os.path.join(OUT, "bending_tilt_leaflet_cases.npz"), **out)

    base = {"surface_tension": 1.0, "bending_modulus": 0.7, "bending_modulus_in": 1.1, "spontaneous_curvature": 0.1,
            "spontaneous_curvature_out": -0.15, "bending_energy_model": "helfrich",
            "bending_gradient_mode": "analytic", "tilt_modulus_in": 2.0, "tilt_modulus_out": 1.4,
            "volume_constraint_mode": "lagrange", "volume_projection_during_minimization": False,
            "mesh_quality_auto_repair_enabled": False}
    allm = ["surface", "tilt_in", "tilt_out", "bending_tilt_in", "bending_tilt_out"]
    P4, T4 = meshgen.icosphere(4)
    P4 = meshgen.smooth_displace(P4, 0.08)
    run_leaflet_trajectory("traj_ico4_gd_btl_nested_cg.npz", P4, T4,
                           dict(base, tilt_solve_mode="nested", tilt_solver="cg", tilt_step_size=0.1,
                                tilt_inner_steps=5), allm, GradientDescent(), 5, 1e-3, fixed_in_every=9)
    run_leaflet_trajectory("traj_ico4_cg_btl_coupled_gd.npz", P4, T4,
                           dict(base, tilt_solve_mode="coupled", tilt_solver="gd", tilt_step_size=0.05,
                                tilt_coupled_steps=3), allm + ["tilt_smoothness_in"], ConjugateGradient(), 5, 1e-3,
                           fixed_out_every=11)
    Pd5, Td5, _ = meshgen.disk_patch(5, bulge=0.35, jitter=0.02, seed=7)
    m0 = build_mesh(Pd5, Td5, base)
    fixed = mesh_arrays(m0)[2].copy()
    run_leaflet_trajectory_fixed("traj_disk5_gd_btl_backtrack.npz", Pd5, Td5, dict(base, tilt_solve_mode="fixed"),
                                 ["surface", "bending_tilt_out", "tilt_out", "bending_tilt_in"], GradientDescent(), 5,
                                 5e-2, fixed)



def _tag_disk(m, frac=0.45):
    """Tag the vertices with the smallest in-plane radius as group "disk" for both leaflets."""
    pos = m.positions_view()
    r = np.linalg.norm(pos[:, :2], axis=1)
    rows = np.flatnonzero((r <= np.quantile(r, frac)) & (pos[:, 2] >= np.median(pos[:, 2]) - 1e-9))
    ids = m.vertex_ids
    for row in rows:
        v = m.vertices[int(ids[row])]
        v.options = dict(getattr(v, "options", None) or {})
        v.options["tilt_disk_target_group_in"] = "disk"
        v.options["tilt_disk_target_group_out"] = "disk"
    return rows


def gen_disk_target():
    """tilt_disk_target_in / _out (tilt_disk_target_in.py:160-286)."""
    import importlib

    out = {"meta_fortran": META}
    Pd, Td, _isb = meshgen.disk_patch(6, bulge=0.3, jitter=0.02, seed=11)
    P, T = meshgen.icosphere(5)
    P = meshgen.smooth_displace(P, 0.06)
    gps = {"bessel": {"surface_tension": 1.0, "tilt_disk_target_group_in": "disk", "tilt_disk_target_strength_in": 20.0,
                      "tilt_disk_target_group_out": "disk", "tilt_disk_target_strength_out": 12.0,
                      "tilt_disk_target_theta_B": 0.6, "tilt_disk_target_theta_B_out": -0.4,
                      "tilt_disk_target_lambda": 1.3, "tilt_disk_target_center": [0.02, -0.01, 0.1],
                      "tilt_disk_target_normal": [0.0, 0.1, 1.0]},
           "linear": {"surface_tension": 1.0, "tilt_disk_target_group_in": "disk", "tilt_disk_target_strength_in": 20.0,
                      "tilt_disk_target_group_out": "disk", "tilt_disk_target_strength_out": 12.0,
                      "tilt_disk_target_theta_B": 0.6, "tilt_disk_target_lambda": 0.0, "tilt_disk_target_radius": 0.9,
                      "tilt_disk_target_normal": [0.0, 0.0, 2.0]},
           "moduli": {"surface_tension": 1.0, "tilt_disk_target_group_in": "disk", "tilt_disk_target_strength_in": 20.0,
                      "tilt_disk_target_group_out": "disk", "tilt_disk_target_strength_out": 12.0,
                      "tilt_disk_target_theta_B": 0.6, "tilt_modulus_in": 2.0, "tilt_modulus_out": 1.0,
                      "bending_modulus": 0.5, "tilt_disk_target_normal": [0.0, 0.0, 1.0]}}
    for name, (P_, T_) in {"disk6": (Pd, Td), "ico5": (P, T)}.items():
        for tag, gp in gps.items():
            m = build_mesh(P_, T_, gp)
            rows = _tag_disk(m)
            tin, tout, _fi, _fo = _set_leaflet_fields(m, 14, 0.25)
            pos, tri, isb, fixed = mesh_arrays(m)
            res = ParameterResolver(m.global_parameters)
            key = f"{name}_{tag}"
            out[name + "_positions"], out[name + "_tri"], out[name + "_is_boundary"] = pos, tri, isb
            out[name + "_tilts_in"], out[name + "_tilts_out"], out[name + "_disk_rows"] = tin, tout, rows
            out[key + "_gp_json"] = np.array(json.dumps(gp, sort_keys=True))
            for mod in ("tilt_disk_target_in", "tilt_disk_target_out"):
                module = importlib.import_module(f"modules.energy.{mod}")
                g = np.zeros_like(pos)
                tgi, tgo = np.zeros_like(pos), np.zeros_like(pos)
                E = module.compute_energy_and_gradient_array(
                    m, m.global_parameters, res, positions=pos, index_map=m.vertex_index_to_row, grad_arr=g,
                    tilts_in=tin, tilts_out=tout, tilt_in_grad_arr=tgi, tilt_out_grad_arr=tgo)
                out[f"{key}_{mod}_E"], out[f"{key}_{mod}_grad"] = np.array(E), g
                out[f"{key}_{mod}_tilt_grad"] = tgi if mod.endswith("_in") else tgo
                print("disk target", key, mod, "E=%.16g" % E, len(rows))
    np.savez_compressed(os.path.join(OUT, "tilt_disk_target_cases.npz"), **out)

    base = {"surface_tension": 1.0, "tilt_modulus_in": 2.0, "tilt_modulus_out": 1.4, "bending_modulus": 0.6,
            "tilt_disk_target_group_in": "disk", "tilt_disk_target_strength_in": 15.0,
            "tilt_disk_target_group_out": "disk", "tilt_disk_target_strength_out": 10.0,
            "tilt_disk_target_theta_B": 0.5, "tilt_disk_target_lambda": 1.0, "tilt_disk_target_normal": [0.0, 0.0, 1.0],
            "volume_constraint_mode": "lagrange", "volume_projection_during_minimization": False,
            "mesh_quality_auto_repair_enabled": False}
    allm = ["surface", "tilt_in", "tilt_out", "tilt_smoothness_in", "tilt_disk_target_in", "tilt_disk_target_out"]
    Pd5, Td5, _ = meshgen.disk_patch(6, bulge=0.3, jitter=0.02, seed=11)
    m0 = build_mesh(Pd5, Td5, base)
    fixed = mesh_arrays(m0)[2].copy()
    global _DISK_TAG
    _DISK_TAG = True
    try:
        run_leaflet_trajectory_fixed("traj_disk6_gd_disktarget_nested_cg.npz", Pd5, Td5,
                                     dict(base, tilt_solve_mode="nested", tilt_solver="cg", tilt_step_size=0.05,
                                          tilt_inner_steps=5), allm, GradientDescent(), 5, 2e-3, fixed)
        run_leaflet_trajectory_fixed("traj_disk6_cg_disktarget_coupled_gd.npz", Pd5, Td5,
                                     dict(base, tilt_solve_mode="coupled", tilt_solver="gd", tilt_step_size=0.03,
                                          tilt_coupled_steps=3), allm, ConjugateGradient(), 5, 2e-3, fixed)
    finally:
        _DISK_TAG = False


_DISK_TAG = False



def gen_angle_defects():
    """compute_angle_defects (geometry/curvature.py:335-403) and the closed-surface gaussian_curvature energy."""
    from geometry.curvature import compute_angle_defects, compute_curvature_fields
    from modules.energy import gaussian_curvature as gcm

    out = {"meta_fortran": META}
    rng = np.random.default_rng(3)
    P, T = meshgen.icosphere(5)
    P = meshgen.smooth_displace(P, 0.08) + 4e-3 * rng.normal(size=P.shape)
    Pd, Td, _ = meshgen.disk_patch(5, bulge=0.35, jitter=0.03, seed=5)
    for name, (P_, T_) in {"ico5": (P, T), "disk5": (Pd, Td)}.items():
        m = build_mesh(P_, T_, {"gaussian_modulus": -0.7})
        pos, tri, isb, fixed = mesh_arrays(m)
        d = compute_angle_defects(m, pos, m.vertex_index_to_row)
        out[name + "_positions"], out[name + "_tri"], out[name + "_is_boundary"], out[name + "_defects"] = pos, tri, isb, d
        if name == "ico5":
            g = np.zeros_like(pos)
            E = gcm.compute_energy_and_gradient_array(m, m.global_parameters, ParameterResolver(m.global_parameters),
                                                      positions=pos, index_map=m.vertex_index_to_row, grad_arr=g)
            assert not np.any(g)
            out["ico5_gaussian_E"] = np.array(E)
        print("angle defects", name, "sum=%.15g" % d.sum())
        # compute_curvature_fields (geometry/curvature.py:404-448): every field of the dataclass
        cf = compute_curvature_fields(m, pos, m.vertex_index_to_row)
        out[name + "_cf_mean_curvature_normal"] = np.array(cf.mean_curvature_normal)
        out[name + "_cf_mean_curvature"] = np.array(cf.mean_curvature)
        out[name + "_cf_mixed_area"] = np.array(cf.mixed_area)
        out[name + "_cf_angle_defect"] = np.array(cf.angle_defect)
        out[name + "_cf_gaussian_curvature"] = np.array(cf.gaussian_curvature)
        out[name + "_cf_principal_curvatures"] = np.array(cf.principal_curvatures)
        if name == "disk5":
            # a surface WITH a boundary loop: E = kappa_bar * G, G = sum_interior (2 pi - theta_v) +
            # sum_boundary-loop (pi - theta_v)  (gaussian_curvature.py:128-143, diagnostics/gauss_bonnet.py:260-340)
            from runtime.diagnostics.gauss_bonnet import gauss_bonnet_invariant

            g = np.zeros_like(pos)
            E = gcm.compute_energy_and_gradient_array(m, m.global_parameters, ParameterResolver(m.global_parameters),
                                                      positions=pos, index_map=m.vertex_index_to_row, grad_arr=g)
            assert not np.any(g)
            G, k_int, b_tot, _ = gauss_bonnet_invariant(m)
            out["disk5_gaussian_E"], out["disk5_gauss_bonnet_G"] = np.array(E), np.array(G)
            out["disk5_gauss_bonnet_interior"], out["disk5_gauss_bonnet_boundary"] = np.array(k_int), np.array(b_tot)
            print("gaussian_curvature with boundary: E=%.15g G=%.15g (interior %.6g + boundary %.6g)" % (E, G, k_int, b_tot))
    np.savez_compressed(os.path.join(OUT, "angle_defect_cases.npz"), **out)


def run_leaflet_trajectory_fixed(fname, P, T, gp, mods, stepper, n_steps, step_size, fixed):
    global build_mesh
    orig = build_mesh

    def bm(P_, T_, gp_, fixed=None, tilts=None):
        return orig(P_, T_, gp_, fixed=fixed_rows, tilts=tilts)

    fixed_rows = fixed
    build_mesh = bm
    try:
        run_leaflet_trajectory(fname, P, T, gp, mods, stepper, n_steps, step_size)
    finally:
        build_mesh = orig


# ---------------------------------------------------------------------------
# (k) line-search guard / exhaustion branches and the Lagrange volume-drift projection
#     (runtime/topology.py:13-48, runtime/steppers/line_search.py:314-426,
#      runtime/minimizer.py:1478-1513, modules/constraints/volume.py:69-149)
# ---------------------------------------------------------------------------
def _count_guard():
    """Count what the line search asks the normal-rotation guard and what it answers."""
    import runtime.steppers.line_search as ls

    stats = {"calls": 0, "rejects": 0}
    orig = ls.check_max_normal_change_positions

    def counted(mesh, original_positions, new_positions, *a, **kw):
        ok = orig(mesh, original_positions=original_positions, new_positions=new_positions, *a, **kw)
        stats["calls"] += 1
        stats["rejects"] += 0 if ok else 1
        return ok

    ls.check_max_normal_change_positions = counted
    return stats, lambda: setattr(ls, "check_max_normal_change_positions", orig)


def _count_enforce():
    stats = {"calls": 0, "volumes": []}
    orig = cvolume.enforce_constraint

    def counted(mesh, *a, **kw):
        before = [float(b.compute_volume(mesh)) for b in mesh.bodies.values()]
        r = orig(mesh, *a, **kw)
        after = [float(b.compute_volume(mesh)) for b in mesh.bodies.values()]
        stats["calls"] += 1
        stats["volumes"].append((before[0], after[0]))
        return r

    cvolume.enforce_constraint = counted
    return stats, lambda: setattr(cvolume, "enforce_constraint", orig)


def gen_guard_and_enforce():
    base_gp = {"surface_tension": 1.0, "bending_modulus": 0.5, "bending_energy_model": "helfrich",
               "spontaneous_curvature": 0.0, "volume_constraint_mode": "lagrange",
               "volume_projection_during_minimization": False, "mesh_quality_auto_repair_enabled": False}
    # A: huge initial step -> alpha * max|d| >= 0.3 * min edge: the normal-rotation guard decides trials
    P, T = meshgen.icosphere(6)
    P = meshgen.smooth_displace(P, 0.1)
    mm = build_mesh(P, T, dict(base_gp))
    mm.energy_modules = ["surface", "bending"]
    mm.constraint_modules = []
    stats, undo = _count_guard()
    try:
        out = run_trajectory("ico6_cg_guard", mm, ConjugateGradient(), 8, step_size=5.0)
    finally:
        undo()
    out["guard_calls"] = np.array(stats["calls"])
    out["guard_rejects"] = np.array(stats["rejects"])
    out["kappa"] = np.array(0.5)
    assert stats["rejects"] > 0, "the guard was expected to reject trials in this case"
    np.savez_compressed(os.path.join(OUT, "traj_ico6_cg_guard.npz"), **out)
    print("traj_ico6_cg_guard.npz E_final=%.16g guard %s" % (out["E_final"], stats), out["step_log"].tolist())

    # B: a search that runs out of its max_iter = 10 trials (line_search.py:425-426): GD from a step so large that
    # ten shrinkages by 0.7 do not reach an acceptable point
    for step0 in (20.0,):
        mm = build_mesh(P, T, dict(base_gp))
        mm.energy_modules = ["surface", "bending"]
        mm.constraint_modules = []
        stats, undo = _count_guard()
        try:
            out = run_trajectory("ico6_gd_exhaust", mm, GradientDescent(), 12, step_size=step0)
        finally:
            undo()
        out["guard_calls"] = np.array(stats["calls"])
        out["guard_rejects"] = np.array(stats["rejects"])
        out["kappa"] = np.array(0.5)
        print("traj_ico6_gd_exhaust step0=%g E_final=%.16g guard %s" % (step0, out["E_final"], stats),
              out["step_log"].tolist())
        assert (out["step_log"][:, 0] == 0.0).sum() >= 3 and out["step_log"][-1, 0] == 1.0, \
            "expected exhausted searches followed by a recovery"
        np.savez_compressed(os.path.join(OUT, "traj_ico6_gd_exhaust.npz"), **out)

    # C: Lagrange volume constraint with a tolerance small enough that the drift check (minimizer.py:1478-1513)
    # fires after accepted steps and volume.enforce_constraint re-projects the positions
    P8, T8 = meshgen.icosphere(8)
    P8 = meshgen.smooth_displace(P8, 0.05)
    gp = dict(base_gp)
    gp.update({"bending_modulus": 1.0, "volume_tolerance": 1.0e-11})
    mm = build_mesh(P8, T8, gp)
    add_body(mm)
    mm.energy_modules = ["surface"]
    mm.constraint_modules = ["volume"]
    stats, undo = _count_enforce()
    try:
        out = run_trajectory("ico8_gd_drift", mm, GradientDescent(), 6, step_size=2e-2)
    finally:
        undo()
    out["enforce_calls"] = np.array(stats["calls"])
    out["enforce_volumes"] = np.array(stats["volumes"])
    out["volume_tolerance"] = np.array(1.0e-11)
    out["volume_final"] = np.array(float(mm.bodies[0].compute_volume(mm)))
    assert stats["calls"] >= 3, stats
    np.savez_compressed(os.path.join(OUT, "traj_ico8_gd_volume_drift.npz"), **out)
    print("traj_ico8_gd_volume_drift.npz E_final=%.16g enforce calls %d" % (out["E_final"], stats["calls"]),
          out["step_log"][:, 0], stats["volumes"][:3])

    # D: the same with a tilt module: every enforce outside the line search is followed by
    # mesh.project_tilts_to_tangent() (minimizer.py:1224, :1506, :1186)
    rng = np.random.default_rng(77)
    P4, T4 = meshgen.icosphere(4)
    P4 = meshgen.smooth_displace(P4, 0.08)
    gp = {"surface_tension": 1.0, "tilt_rigidity": 2.5, "volume_constraint_mode": "lagrange",
          "volume_projection_during_minimization": False, "mesh_quality_auto_repair_enabled": False,
          "volume_tolerance": 1.0e-11}
    tl = 0.3 * rng.normal(size=P4.shape)
    mm = build_mesh(P4, T4, gp, tilts=tl)
    add_body(mm)
    mm.energy_modules = ["surface", "tilt"]
    mm.constraint_modules = ["volume"]
    stats, undo = _count_enforce()
    try:
        out = run_trajectory("ico4_gd_tilt_drift", mm, GradientDescent(), 5, step_size=2e-2, mesh_path=True)
    finally:
        undo()
    out["tilts0"] = tl
    out["tilts_final"] = np.ascontiguousarray(mm.tilts_view()).copy()
    out["k_tilt"] = np.array(2.5)
    out["enforce_calls"] = np.array(stats["calls"])
    out["volume_tolerance"] = np.array(1.0e-11)
    assert stats["calls"] >= 3, stats
    np.savez_compressed(os.path.join(OUT, "traj_ico4_gd_tilt_volume_drift.npz"), **out)
    print("traj_ico4_gd_tilt_volume_drift.npz E_final=%.16g enforce calls %d" % (out["E_final"], stats["calls"]),
          out["step_log"][:, 0])


# ---------------------------------------------------------------------------
# (k2) the enforcer lane of the line search (line_search.py:428-487 with constraint_enforcer =
#      Minimizer._enforce_constraints, minimizer.py:1379) with the PROGRAMMATIC defaults of the volume parameters:
#      GlobalParameters() gives volume_constraint_mode = "lagrange" and volume_projection_during_minimization = True
#      (core/parameters/global_parameters.py:18,24), so every trial is projected onto the target volume
#      (volume.enforce_constraint, three linearised steps) before its energy is taken, next to the k = 1 KKT
#      projection of the gradient.
# ---------------------------------------------------------------------------
def gen_enforcer_lane():
    P8, T8 = meshgen.icosphere(8)
    P8 = meshgen.smooth_displace(P8, 0.05)
    cases = [
        ("traj_ico8_gd_volume_enforcer.npz", ["surface"], GradientDescent, 6, 2e-2,
         {"surface_tension": 1.0, "mesh_quality_auto_repair_enabled": False}),
        ("traj_ico8_cg_bending_volume_enforcer.npz", ["surface", "bending"], ConjugateGradient, 8, 1e-3,
         {"surface_tension": 1.0, "bending_modulus": 1.0, "bending_energy_model": "helfrich",
          "spontaneous_curvature": 0.3, "mesh_quality_auto_repair_enabled": False}),
    ]
    for fname, mods, stepper_cls, n_steps, step0, gp in cases:
        mm = build_mesh(P8, T8, dict(gp))
        assert mm.global_parameters.get("volume_projection_during_minimization", True) is True
        assert mm.global_parameters.get("volume_constraint_mode", "lagrange") == "lagrange"
        add_body(mm)
        mm.energy_modules = list(mods)
        mm.constraint_modules = ["volume"]
        stats, undo = _count_enforce()
        try:
            out = run_trajectory(fname, mm, stepper_cls(), n_steps, step_size=step0)
        finally:
            undo()
        out["enforce_calls"] = np.array(stats["calls"])
        out["volume_final"] = np.array(float(mm.bodies[0].compute_volume(mm)))
        if "bending" in mods:
            out["kappa"] = np.array(1.0)
            out["c0"] = np.array(0.3)
        # (start + one per line-search trial + finalize)
        assert stats["calls"] >= n_steps + 2, stats
        np.savez_compressed(os.path.join(OUT, fname), **out)
        print("%s E_final=%.16g enforce calls %d" % (fname, out["E_final"], stats["calls"]), out["step_log"].tolist())


# ---------------------------------------------------------------------------
# (k2) ConjugateGradient(precondition=True) (conjugate_gradient.py:74-76): direction from the row-normalised gradient
# ---------------------------------------------------------------------------
def gen_precondition():
    P8, T8 = meshgen.icosphere(8)
    P8 = meshgen.smooth_displace(P8, 0.05)
    base = {"surface_tension": 1.0, "bending_modulus": 1.0, "bending_energy_model": "helfrich",
            "volume_constraint_mode": "lagrange", "volume_projection_during_minimization": False,
            "mesh_quality_auto_repair_enabled": False}
    cases = [
        # (one restart at iteration 10; a first step long enough for backtracking)
        ("traj_ico8_cg_precondition.npz", [], 0.0, 12, 4e-3),
        ("traj_ico8_cg_precondition_volume.npz", ["volume"], 0.3, 8, 1e-3),
    ]
    for fname, cons, c0, n_steps, step0 in cases:
        mm = build_mesh(P8, T8, dict(base, spontaneous_curvature=c0))
        add_body(mm)
        mm.energy_modules = ["surface", "bending"]
        mm.constraint_modules = list(cons)
        out = run_trajectory(fname, mm, ConjugateGradient(precondition=True), n_steps, step_size=step0)
        out["kappa"] = np.array(1.0)
        out["c0"] = np.array(c0)
        out["precondition"] = np.array(1)
        np.savez_compressed(os.path.join(OUT, fname), **out)
        print("%s E_final=%.16g" % (fname, out["E_final"]), out["step_log"].tolist())


# ---------------------------------------------------------------------------
# (l) BASELINE config 5 on its own deck: meshes/caveolin/kozlov_1disk_3d_tensionless_bilayer_profile.yaml.
#     Kept from the deck: positions, triangle rows, fixed / tilt_fixed_in / tilt_fixed_out flags, the "disk" group
#     rows, every global parameter and the energy-module list.  NOT kept: its three constraint modules
#     (pin_to_plane, pin_to_circle, rim_slope_match_out: SURVEY section 2 puts every constraint but `volume` out of
#     scope) and the mesh-quality auto repair.
# ---------------------------------------------------------------------------
def gen_config5():
    import importlib

    deck = os.path.join(args.reference, "meshes", "caveolin", "kozlov_1disk_3d_tensionless_bilayer_profile.yaml")
    m = parse_geometry(load_data(deck))
    mods = list(m.energy_modules)
    deck_constraints = list(m.constraint_modules)
    m.constraint_modules = []
    m.global_parameters.set("mesh_quality_auto_repair_enabled", False)
    gp = {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in m.global_parameters.to_dict().items()}
    pos0, tri, isb, fixed = mesh_arrays(m)
    ids = m.vertex_ids
    nv = len(ids)
    fin = np.array([bool(getattr(m.vertices[int(v)], "tilt_fixed_in", False)) for v in ids])
    fout = np.array([bool(getattr(m.vertices[int(v)], "tilt_fixed_out", False)) for v in ids])
    disk_rows = np.array([r for r, v in enumerate(ids)
                          if (getattr(m.vertices[int(v)], "options", None) or {}).get("tilt_disk_target_group_in")
                          == gp.get("tilt_disk_target_group_in")], dtype=int)
    disk_rows_out = np.array([r for r, v in enumerate(ids)
                              if (getattr(m.vertices[int(v)], "options", None) or {}).get("tilt_disk_target_group_out")
                              == gp.get("tilt_disk_target_group_out")], dtype=int)
    assert np.array_equal(disk_rows, disk_rows_out)
    out = {"meta_fortran": META, "positions0": pos0, "tri": tri, "is_boundary": isb, "fixed": fixed,
           "tilt_fixed_in": fin, "tilt_fixed_out": fout, "disk_rows": disk_rows,
           "gamma": m.get_facet_parameter_array("surface_tension").copy(),
           "tilts_in0": np.ascontiguousarray(m.tilts_in_view()).copy(),
           "tilts_out0": np.ascontiguousarray(m.tilts_out_view()).copy(),
           "gp_json": np.array(json.dumps(gp, sort_keys=True)), "modules": np.array(mods),
           "deck_constraint_modules_not_kept": np.array(deck_constraints)}
    print("config5 deck: nv=%d nf=%d boundary=%d fixed=%d tilt_fixed_in=%d disk rows=%d modules=%s" %
          (nv, len(tri), int(isb.sum()), int(fixed.sum()), int(fin.sum()), len(disk_rows), mods))

    # (1) every energy module of the deck on the deck's surface with seeded tangent tilt fields (the deck's own
    #     fields are zero): energy, shape gradient, both tilt gradients -- the plugin API
    rng = np.random.default_rng(505)
    tin = _tangent_tilts(m, rng, 0.2)
    tout = _tangent_tilts(m, rng, 0.15)
    tin[fin] = 0.0
    tout[fout] = 0.0
    out["state_b_tilts_in"], out["state_b_tilts_out"] = tin, tout
    m.set_tilts_in_from_array(tin)
    m.set_tilts_out_from_array(tout)
    res = ParameterResolver(m.global_parameters)
    for mod in mods:
        module = importlib.import_module(f"modules.energy.{mod}")
        g = np.zeros_like(pos0)
        tgi, tgo = np.zeros_like(pos0), np.zeros_like(pos0)
        E = module.compute_energy_and_gradient_array(
            m, m.global_parameters, res, positions=pos0, index_map=m.vertex_index_to_row, grad_arr=g,
            tilts_in=tin, tilts_out=tout, tilt_in_grad_arr=tgi, tilt_out_grad_arr=tgo)
        out[f"mod_{mod}_E"], out[f"mod_{mod}_grad"] = np.array(float(E)), g
        out[f"mod_{mod}_tilt_grad_in"], out[f"mod_{mod}_tilt_grad_out"] = tgi, tgo
        print("  config5", mod, "E=%.16g" % float(E), "|grad|max %.3e |tg_in|max %.3e |tg_out|max %.3e" %
              (np.abs(g).max(), np.abs(tgi).max(), np.abs(tgo).max()))
    em = EnergyModuleManager(mods)
    cm = ConstraintModuleManager([])
    m.energy_modules = list(mods)
    mz = Minimizer(m, m.global_parameters, GradientDescent(), em, cm, quiet=True, step_size=float(gp["step_size"]))
    E_b, g_b = mz.compute_energy_and_gradient_array()
    out["state_b_E"], out["state_b_grad"] = np.array(E_b), np.array(g_b)

    # (2) ONE relax_leaflet_tilts call as the deck configures it (coupled mode, jacobi CG, 40 inner steps, step 0.15)
    #     from the deck's own zero fields: the disk-target modules drive the tilts
    m.set_tilts_in_from_array(out["tilts_in0"])
    m.set_tilts_out_from_array(out["tilts_out0"])
    m.increment_version()
    mz._relax_leaflet_tilts(positions=m.positions_view(), mode="coupled")
    out["relax_tilts_in"] = np.ascontiguousarray(m.tilts_in_view()).copy()
    out["relax_tilts_out"] = np.ascontiguousarray(m.tilts_out_view()).copy()
    out["relax_E"] = np.array(float(mz.compute_energy()))
    print("  config5 relax_leaflet_tilts(coupled): E=%.16g |t_in|max %.4f |t_out|max %.4f" %
          (out["relax_E"], np.abs(out["relax_tilts_in"]).max(), np.abs(out["relax_tilts_out"]).max()))

    # (3) the deck's own macro `g`: minimizer steps (fixed step mode 0.01, coupled tilt relaxation at the top of
    #     every iteration) from the deck state -- a short trajectory in the leaflet-trajectory format
    m2 = parse_geometry(load_data(deck))
    m2.constraint_modules = []
    m2.global_parameters.set("mesh_quality_auto_repair_enabled", False)
    stepper = GradientDescent()
    mz2 = Minimizer(m2, m2.global_parameters, stepper, EnergyModuleManager(mods), ConstraintModuleManager([]),
                    quiet=True, step_size=float(gp["step_size"]))
    log = []
    orig_step = stepper.step

    def logged_step(mesh, grad, step_size, energy_fn, constraint_enforcer=None):
        r = orig_step(mesh, grad, step_size, energy_fn, constraint_enforcer=constraint_enforcer)
        log.append((float(bool(r[0])), float(r[1]), float(r[2])))
        return r

    stepper.step = logged_step
    snaps, tin_snaps, tout_snaps = [], [], []

    def cb(mesh, i):
        snaps.append(mesh.positions_view().copy())
        tin_snaps.append(np.ascontiguousarray(mesh.tilts_in_view()).copy())
        tout_snaps.append(np.ascontiguousarray(mesh.tilts_out_view()).copy())

    E0, g0 = mz2.compute_energy_and_gradient_array()
    n_steps = 4
    r = mz2.minimize(n_steps, callback=cb)
    out.update({"E0": np.array(E0), "grad0": np.array(g0), "positions_iter": np.array(snaps),
                "tilts_in_iter": np.array(tin_snaps), "tilts_out_iter": np.array(tout_snaps),
                "positions_final": m2.positions_view().copy(),
                "tilts_in_final": np.ascontiguousarray(m2.tilts_in_view()).copy(),
                "tilts_out_final": np.ascontiguousarray(m2.tilts_out_view()).copy(),
                "step_log": np.array(log), "E_final": np.array(r["energy"]), "n_steps": np.array(n_steps),
                "step_size0": np.array(float(gp["step_size"]))})
    np.savez_compressed(os.path.join(OUT, "traj_config5_deck_gd.npz"), **out)
    print("traj_config5_deck_gd.npz E0=%.16g E_final=%.16g" % (E0, out["E_final"]), out["step_log"].tolist())


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if "--only-tilt" in sys.argv:
        gen_tilt_trajectory()
        sys.exit(0)
    if "--only-leaflet" in sys.argv:
        gen_leaflet()
        sys.exit(0)
    if "--only-defects" in sys.argv:
        gen_angle_defects()
        sys.exit(0)
    if "--only-guard" in sys.argv:
        gen_guard_and_enforce()
        sys.exit(0)
    if "--only-config5" in sys.argv:
        gen_config5()
        sys.exit(0)
    if "--only-enforcer" in sys.argv:
        gen_enforcer_lane()
        sys.exit(0)
    if "--only-precondition" in sys.argv:
        gen_precondition()
        sys.exit(0)
    if "--only-disk" in sys.argv:
        gen_disk_target()
        sys.exit(0)
    if "--only-btl" in sys.argv:
        gen_bending_tilt_leaflet()
        sys.exit(0)
    if "--only-ts" in sys.argv:
        gen_tilt_smoothness()
        sys.exit(0)
    if "--only-bt" in sys.argv:
        gen_bending_tilt_cases()
        gen_bending_tilt_trajectories()
        sys.exit(0)
    gen_kernel_cases()
    gen_mesh_cases()
    gen_trajectories()
    gen_tilt_trajectory()
    gen_bending_tilt_cases()
    gen_bending_tilt_trajectories()
    gen_tilt_smoothness()
    gen_leaflet()
    gen_bending_tilt_leaflet()
    gen_disk_target()
    gen_angle_defects()
    gen_guard_and_enforce()
    gen_enforcer_lane()
    gen_precondition()
    gen_config5()
