"""Shared pieces of the two-leaflet tilt plugins (tilt_in, tilt_out, tilt_smoothness_in, tilt_smoothness_out).

Parameter resolution follows the reference's helpers (modules/energy/tilt_params.py:6-23,
modules/energy/tilt_smoothness_utils.py:95-100, runtime/preconditioners.py:111-120); the toggles the
device path does not implement raise ``MembraneHipError`` instead of being ignored.
"""

from __future__ import annotations

import numpy as np

from ... import _lib as L
from ...geometry.mesh import mirror_for

LEAFLET_BITS = {("tilt", "in"): L.MS_MOD_TILT_IN, ("tilt", "out"): L.MS_MOD_TILT_OUT,
                ("smooth", "in"): L.MS_MOD_TILT_SMOOTH_IN, ("smooth", "out"): L.MS_MOD_TILT_SMOOTH_OUT,
                ("bt", "in"): L.MS_MOD_BENDING_TILT_IN, ("bt", "out"): L.MS_MOD_BENDING_TILT_OUT,
                ("disk", "in"): L.MS_MOD_TILT_DISK_TARGET_IN, ("disk", "out"): L.MS_MOD_TILT_DISK_TARGET_OUT}

# options of the reference's leaflet modules that change their result and are not on the device path
_UNSUPPORTED_KEYS = (
    "leaflet_out_absent_presets", "leaflet_in_absent_presets",  # leaflet_presence.py
    "tilt_in_exclude_shared_rim_outer_rows", "tilt_out_exclude_shared_rim_outer_rows",
    "tilt_exclude_shared_rim_outer_rows_in", "tilt_exclude_shared_rim_outer_rows_out",
    "tilt_out_exclude_shared_rim_rows", "tilt_exclude_shared_rim_rows_out",
    "tilt_in_shared_rim_outer_shell_mass_mode", "tilt_out_shared_rim_outer_shell_mass_mode",
    "tilt_axisymmetric_about_thetaB_center", "inner_coupled_update_mode", "tilt_relax_energy_guard_factor",
    "tilt_thetaB_optimize",
    # bending_tilt_leaflet.py toggles (bt_params.py:13-222, bt_selection.py, bt_divergence.py)
    "theory_parity_lane", "bending_tilt_assume_J0_presets", "bending_tilt_assume_J0_presets_in",
    "bending_tilt_assume_J0_presets_out", "bending_tilt_base_term_reference_mode",
    "bending_tilt_base_term_reference_mode_in", "bending_tilt_base_term_reference_mode_out",
    "bending_tilt_base_term_boundary_group_in", "bending_tilt_base_term_boundary_group_out",
    "bending_tilt_base_term_region_mode", "bending_tilt_in_update_mode",
    "bending_tilt_in_scaffold_shape_stencil_mode", "bending_tilt_interface_divergence_mode",
    "bending_tilt_out_interface_divergence_mode",
)


def _get(param_resolver, global_params, key):
    val = param_resolver.get(None, key) if param_resolver is not None else None
    if val is None and global_params is not None:
        val = global_params.get(key)
    return val


def check_supported(global_params) -> None:
    for key in _UNSUPPORTED_KEYS:
        val = global_params.get(key) if global_params is not None else None
        if val in (None, False, 0, 0.0, "", "off", "none", [], ()):
            continue
        raise L.MembraneHipError(f"leaflet tilt option {key}={val!r} is outside the HIP hot path")
    model = str(global_params.get("tilt_transport_model", "ambient_v1") or "ambient_v1").strip().lower()
    if model != "ambient_v1":
        raise L.MembraneHipError("tilt_transport_model other than ambient_v1 is outside the HIP hot path")
    cadence = str(global_params.get("tilt_projection_cadence", "per_step") or "per_step").strip().lower()
    if cadence not in {"per_step", "per_pass"}:
        raise ValueError("tilt_projection_cadence must be 'per_step' or 'per_pass'.")
    fb = str(global_params.get("tilt_cg_rejection_fallback", "off") or "off").strip().lower()
    if fb not in {"off", "gd"}:
        raise ValueError("tilt_cg_rejection_fallback must be 'off' or 'gd'.")
    if fb == "gd":
        raise L.MembraneHipError("tilt_cg_rejection_fallback=gd is outside the HIP hot path")


def tilt_modulus(param_resolver, global_params, leaflet: str) -> float:
    k = _get(param_resolver, global_params, f"tilt_modulus_{leaflet}")
    if k is None:
        k = _get(param_resolver, global_params, f"tilt_modolus_{leaflet}")  # legacy typo fallback
    return float(k or 0.0)


def tilt_mass_mode(param_resolver, global_params, leaflet: str) -> str:
    mode = _get(param_resolver, global_params, f"tilt_mass_mode_{leaflet}")
    if mode is None:
        mode = _get(param_resolver, global_params, "tilt_mass_mode")
    txt = str(mode or "lumped").strip().lower()
    if txt not in {"lumped", "consistent"}:
        raise ValueError(f"tilt_mass_mode_{leaflet} must be 'lumped' or 'consistent'.")
    return txt


def smoothness_rigidity(param_resolver, global_params, leaflet: str) -> float:
    k = _get(param_resolver, global_params, f"bending_modulus_{leaflet}")
    if k is None:
        k = _get(param_resolver, global_params, "bending_modulus")
    return float(k or 0.0)


def precond_smoothness(param_resolver, global_params, leaflet: str) -> float:
    """Rigidity in the leaflet Jacobi diagonal: ``get(bending_modulus_<l>) or get(bending_modulus) or 0``."""
    return float(_get(param_resolver, global_params, f"bending_modulus_{leaflet}")
                 or _get(param_resolver, global_params, "bending_modulus") or 0.0)


def relax_tilt_modulus(param_resolver, global_params, leaflet: str) -> float:
    """The relaxation's fast path reads ``tilt_modulus_<l>`` only (evaluation_manager.py:566, 674)."""
    return float(_get(param_resolver, global_params, f"tilt_modulus_{leaflet}") or 0.0)


def bending_params(mesh, global_params, leaflet: str):
    """(kappa, c0) arrays of bending_tilt_in/out (bt_params.py:225-318): leaflet modulus / spontaneous curvature with
    the global fallbacks; per-vertex arrays of an ArrayMesh override like vertex options do."""
    nv = len(mesh.vertex_ids)
    k = global_params.get(f"bending_modulus_{leaflet}")
    if k is None:
        k = global_params.get("bending_modulus", 0.0)
    c = global_params.get(f"spontaneous_curvature_{leaflet}")
    if c is None:
        c = global_params.get("spontaneous_curvature")
        if c is None:
            c = global_params.get("intrinsic_curvature", 0.0)
    kappa = np.full(nv, float(k or 0.0))
    c0 = np.full(nv, float(c or 0.0))
    getter = getattr(mesh, "get_vertex_parameter_array", None)
    if getter is not None:
        for key in (f"bending_modulus_{leaflet}", "bending_modulus"):
            arr = getter(key)
            if arr is not None:
                kappa = np.asarray(arr, dtype=np.float64)
                break
        for key in (f"spontaneous_curvature_{leaflet}", "spontaneous_curvature"):
            arr = getter(key)
            if arr is not None:
                c0 = np.asarray(arr, dtype=np.float64)
                break
    elif isinstance(getattr(mesh, "vertices", None), dict):
        rows = mesh.vertex_index_to_row
        for vid, vertex in mesh.vertices.items():
            row = rows.get(int(vid))
            opts = getattr(vertex, "options", None) or {}
            if row is None:
                continue
            for key in (f"bending_modulus_{leaflet}", "bending_modulus"):
                if key in opts:
                    try:
                        kappa[row] = float(opts[key])
                    except (TypeError, ValueError):
                        pass
                    break
            for key in (f"spontaneous_curvature_{leaflet}", "spontaneous_curvature", "intrinsic_curvature"):
                if opts.get(key) is not None:
                    try:
                        c0[row] = float(opts[key])
                    except (TypeError, ValueError):
                        pass
                    break
    return kappa, c0


def disk_target_rows(mesh, leaflet: str, group: str) -> np.ndarray:
    """Rows tagged ``tilt_disk_target_group_<leaflet> == group`` (tilt_disk_target_in.py:137-145): vertex options of
    a reference Mesh, or the ``disk_rows_<leaflet>`` attribute (mask / row list) of an array mesh."""
    nv = len(mesh.vertex_ids)
    own = getattr(mesh, f"disk_rows_{leaflet}", None)
    if own is not None:
        own = np.asarray(own)
        return np.flatnonzero(own) if own.dtype == bool else own.astype(np.int64)
    verts = getattr(mesh, "vertices", None)
    if isinstance(verts, dict):
        rows = []
        key = f"tilt_disk_target_group_{leaflet}"
        for vid in mesh.vertex_ids:
            opts = getattr(verts[int(vid)], "options", None) or {}
            if opts.get(key) == group:
                row = mesh.vertex_index_to_row.get(int(vid))
                if row is not None:
                    rows.append(int(row))
        return np.asarray(rows, dtype=np.int64)
    return np.zeros(0, dtype=np.int64) if nv >= 0 else None


def disk_target_params(mesh, param_resolver, global_params, leaflet: str):
    """Resolved parameters of tilt_disk_target_<leaflet> (tilt_disk_target_in.py:38-134) as kwargs of
    DeviceMesh.set_leaflet_disk_target, or None when the module contributes nothing (:175-191)."""

    def pick(name):
        v = _get(param_resolver, global_params, f"tilt_disk_target_{name}_{leaflet}")
        return _get(param_resolver, global_params, f"tilt_disk_target_{name}") if v is None else v

    raw = _get(param_resolver, global_params, f"tilt_disk_target_group_{leaflet}")
    group = None if raw is None else str(raw).strip()
    if not group:
        return None
    k = float(_get(param_resolver, global_params, f"tilt_disk_target_strength_{leaflet}") or 0.0)
    theta_b = float(pick("theta_B") or 0.0)
    if k == 0.0 or theta_b == 0.0:
        return None
    rows = disk_target_rows(mesh, leaflet, group)
    if rows.size == 0:
        return None
    normal = pick("normal")
    if normal is None or float(np.linalg.norm(np.asarray(normal, dtype=float))) < 1e-15:
        raise L.MembraneHipError("tilt_disk_target without tilt_disk_target_normal (SVD plane fit of the disk rows) "
                                 "is outside the HIP hot path")
    center = pick("center")
    radius = pick("radius")
    try:
        radius = float(radius) if radius is not None and float(radius) > 0.0 else None
    except (TypeError, ValueError):
        radius = None
    lam = pick("lambda")
    if lam is not None:
        try:
            lam = float(lam)
        except (TypeError, ValueError):
            lam = 0.0
    else:
        kt = _get(param_resolver, global_params, f"tilt_modulus_{leaflet}")
        if kt is None and leaflet == "in":
            kt = _get(param_resolver, global_params, "tilt_modolus_in")
        kap = _get(param_resolver, global_params, f"bending_modulus_{leaflet}")
        if kap is None:
            kap = _get(param_resolver, global_params, "bending_modulus")
        lam = 0.0
        try:
            if kt is not None and kap is not None and float(kt) > 0.0 and float(kap) > 0.0:
                lam = float(np.sqrt(float(kt) / float(kap)))
        except (TypeError, ValueError):
            lam = 0.0
    return {"disk_rows": rows, "strength": k, "theta_b": theta_b, "lam": lam,
            "center": tuple(np.asarray([0.0, 0.0, 0.0] if center is None else center, dtype=float).reshape(3)),
            "normal": tuple(np.asarray(normal, dtype=float).reshape(3)), "radius": radius}


def check_bt_supported(global_params) -> None:
    mode = str(global_params.get("bending_gradient_mode", "analytic") or "analytic").strip().lower()
    if mode != "analytic":
        raise L.MembraneHipError("bending_tilt_in/out: only bending_gradient_mode=analytic is on the HIP hot path")


def device_params(param_resolver, global_params, leaflet: str) -> dict:
    return {"tilt_modulus": tilt_modulus(param_resolver, global_params, leaflet),
            "mass_mode": tilt_mass_mode(param_resolver, global_params, leaflet),
            "smoothness": smoothness_rigidity(param_resolver, global_params, leaflet),
            "precond_smoothness": precond_smoothness(param_resolver, global_params, leaflet)}


def evaluate(mesh, global_params, param_resolver, *, kind: str, leaflet: str, positions, tilts, grad_arr,
             tilt_grad_arr) -> float:
    """One leaflet module on caller arrays (H2D/D2H per call: the parity seam, not the hot loop)."""
    check_supported(global_params)
    tri, _f = mesh.triangle_row_cache()
    if tri is None or len(tri) == 0:
        return 0.0
    nv = len(mesh.vertex_ids)
    if tilts is None:
        tilts = mesh.tilts_in_view() if leaflet == "in" else mesh.tilts_out_view()
    tilts = np.ascontiguousarray(tilts, dtype=np.float64)
    if tilts.shape != (nv, 3):
        raise ValueError(f"tilts for leaflet '{leaflet}' must have shape (N_vertices, 3)")
    if tilt_grad_arr is not None and np.shape(tilt_grad_arr) != (nv, 3):
        raise ValueError(f"tilt_grad_arr for leaflet '{leaflet}' must have shape (N_vertices, 3)")
    params = device_params(param_resolver, global_params, leaflet)
    mir = mirror_for(mesh)
    dm = mir.sync(positions=None if positions is mesh.positions_view() else positions)
    other = "out" if leaflet == "in" else "in"
    dm.set_leaflet_tilts(leaflet, tilts, **params)
    dm.set_leaflet_tilts(other, np.zeros((nv, 3)), tilt_modulus=0.0, smoothness=0.0)
    mir._leaflet_keys = {}  # foreign arrays were uploaded
    if kind == "bt":
        check_bt_supported(global_params)
        dm.set_leaflet_bending(leaflet, *bending_params(mesh, global_params, leaflet))
    if kind == "disk":
        prm = disk_target_params(mesh, param_resolver, global_params, leaflet)
        if prm is None:
            return 0.0
        dm.set_leaflet_disk_target(leaflet, **prm)
    dm.set_params(modules=LEAFLET_BITS[(kind, leaflet)])
    if kind == "disk":
        if grad_arr is not None:
            e, g = dm.energy_and_gradient(want_grad=True, raw=True)
            grad_arr += g
            E = float(e[3])
        else:
            E = float(dm.energy()[3])
    elif kind == "bt":
        if grad_arr is not None:
            e, g = dm.energy_and_gradient(want_grad=True, raw=True)
            grad_arr += g
            E = float(e[1])
        else:
            E = float(dm.energy()[1])
    elif grad_arr is not None and kind == "tilt":
        e, g = dm.energy_and_gradient(want_grad=True, raw=True)
        grad_arr += g
        E = float(e[3])
    else:
        E = float(dm.energy()[3])
    if tilt_grad_arr is not None:
        # the module's own mass mode (tilt_leaflet.py:116-150): lumped k A/3 t_k or consistent k A/12 (2 t_k + t_a + t_b)
        _e, gi, go = dm.leaflet_tilt_energy_and_gradient(want_gradient=True, module_form=(kind == "tilt"))
        tilt_grad_arr += gi if leaflet == "in" else go
    return E
