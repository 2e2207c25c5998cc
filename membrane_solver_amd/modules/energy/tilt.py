"""Vertex-tilt magnitude energy plugin on the HIP path.

Drop-in for the reference's modules/energy/tilt.py:99-226:
    E = sum_f 1/2 k_t (|t0|^2 + |t1|^2 + |t2|^2)/3 * A_f        (facets with |n| >= 1e-12)
shape gradient coeff_f * dA/dv (accumulated into ``grad_arr``) and, when
``tilt_grad_arr`` is given, the tilt gradient k_t t_v A_v with barycentric vertex areas
(accumulated into it).  ``tilts`` defaults to ``mesh.tilts_view()``.
"""

from __future__ import annotations

from typing import Dict

import numpy as np

from ... import _lib as L
from ...geometry.mesh import mirror_for

USES_TILT = True


def _k_tilt(param_resolver, global_params) -> float:
    val = param_resolver.get(None, "tilt_rigidity") if param_resolver is not None else None
    if val is None:
        val = global_params.get("tilt_rigidity")
    return float(val or 0.0)


def _evaluate(mesh, global_params, k_tilt, positions, tilts, want_grad):
    mir = mirror_for(mesh)
    dm = mir.sync(positions=None if positions is mesh.positions_view() else positions)
    if tilts is None:
        tilts = mesh.tilts_view()
    tilts = np.ascontiguousarray(tilts, dtype=np.float64)
    if tilts.shape != (len(mesh.vertex_ids), 3):
        raise ValueError("tilts must have shape (N_vertices, 3)")
    dm.set_tilts(tilts, k_tilt)
    mir._tilt_key = None  # a foreign array may have been uploaded
    dm.set_params(modules=L.MS_MOD_TILT)
    if want_grad:
        e, g = dm.energy_and_gradient(want_grad=True, raw=True)
        return float(e[3]), g, dm
    return float(dm.energy()[3]), None, dm


def compute_energy_and_gradient_array(mesh, global_params, param_resolver, *, positions: np.ndarray,
                                      index_map: Dict[int, int], grad_arr: np.ndarray,
                                      tilts: np.ndarray | None = None,
                                      tilt_grad_arr: np.ndarray | None = None) -> float:
    _ = index_map
    k_tilt = _k_tilt(param_resolver, global_params)
    if k_tilt == 0.0:
        return 0.0
    tri, _f = mesh.triangle_row_cache()
    if tri is None or len(tri) == 0:
        return 0.0
    E, g, dm = _evaluate(mesh, global_params, k_tilt, positions, tilts, True)
    if grad_arr is not None:
        grad_arr += g
    if tilt_grad_arr is not None:
        if tilt_grad_arr.shape != (len(mesh.vertex_ids), 3):
            raise ValueError("tilt_grad_arr must have shape (N_vertices, 3)")
        tilt_grad_arr += dm.get_tilt_gradient()
    return E


def compute_energy_array(mesh, global_params, param_resolver, *, positions: np.ndarray,
                         index_map: Dict[int, int], tilts: np.ndarray | None = None) -> float:
    _ = index_map
    k_tilt = _k_tilt(param_resolver, global_params)
    if k_tilt == 0.0:
        return 0.0
    tri, _f = mesh.triangle_row_cache()
    if tri is None or len(tri) == 0:
        return 0.0
    E, _g, _dm = _evaluate(mesh, global_params, k_tilt, positions, tilts, False)
    return E


def compute_energy_and_gradient(mesh, global_params, param_resolver):
    """Dict API of the reference (tilt.py:29-96): (E, shape_grad, tilt_grad)."""
    positions = mesh.positions_view()
    g = np.zeros_like(positions)
    tg = np.zeros_like(positions)
    E = compute_energy_and_gradient_array(mesh, global_params, param_resolver, positions=positions,
                                          index_map=mesh.vertex_index_to_row, grad_arr=g,
                                          tilt_grad_arr=tg)
    ids = mesh.vertex_ids
    return (float(E), {int(v): g[r].copy() for r, v in enumerate(ids)},
            {int(v): tg[r].copy() for r, v in enumerate(ids)})


__all__ = ["compute_energy_and_gradient", "compute_energy_and_gradient_array", "compute_energy_array"]
