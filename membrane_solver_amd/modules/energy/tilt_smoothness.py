"""Tilt smoothness (Dirichlet) energy plugin on the HIP path.

Drop-in for the reference's modules/energy/tilt_smoothness.py (ambient_v1 transport):
    E = k_s/4 sum_f [c0 |t1 - t2|^2 + c1 |t2 - t0|^2 + c2 |t0 - t1|^2]
with the triangle cotangents of ``positions``; exact tilt gradient accumulated into
``tilt_grad_arr``; the module has no shape gradient (tilt_smoothness.py:21-23), so ``grad_arr``
is left untouched.
"""

from __future__ import annotations

from typing import Dict

import numpy as np

from ... import _lib as L
from ...geometry.mesh import mirror_for

USES_TILT = True


def _k_smooth(param_resolver, global_params) -> float:
    val = param_resolver.get(None, "tilt_smoothness_rigidity") if param_resolver is not None else None
    if val is None:
        val = global_params.get("tilt_smoothness_rigidity")
    return float(val or 0.0)


def _device(mesh, global_params, k_smooth, positions, tilts):
    model = str(global_params.get("tilt_transport_model", "ambient_v1") or "ambient_v1").strip().lower()
    if model != "ambient_v1":
        raise L.MembraneHipError("tilt_transport_model other than ambient_v1 is outside the HIP hot path")
    mir = mirror_for(mesh)
    dm = mir.sync(positions=None if positions is mesh.positions_view() else positions)
    if tilts is None:
        tilts = mesh.tilts_view()
    tilts = np.ascontiguousarray(tilts, dtype=np.float64)
    if tilts.shape != (len(mesh.vertex_ids), 3):
        raise ValueError("tilts must have shape (N_vertices, 3)")
    dm.set_tilts(tilts, float(global_params.get("tilt_rigidity", 0.0) or 0.0))
    mir._tilt_key = None  # a foreign array may have been uploaded
    dm.set_tilt_smoothness(k_smooth)
    dm.set_params(modules=L.MS_MOD_TILT_SMOOTH)
    return dm


def compute_energy_and_gradient_array(mesh, global_params, param_resolver, *, positions: np.ndarray,
                                      index_map: Dict[int, int], grad_arr: np.ndarray | None,
                                      tilts: np.ndarray | None = None,
                                      tilt_grad_arr: np.ndarray | None = None, ctx=None) -> float:
    _ = (index_map, grad_arr, ctx)
    k_s = _k_smooth(param_resolver, global_params)
    if k_s == 0.0:
        return 0.0
    tri, _f = mesh.triangle_row_cache()
    if tri is None or len(tri) == 0:
        return 0.0
    dm = _device(mesh, global_params, k_s, positions, tilts)
    E, tg = dm.tilt_energy_and_gradient(want_gradient=tilt_grad_arr is not None)
    if tilt_grad_arr is not None:
        tilt_grad_arr += tg
    return float(E)


def compute_energy_array(mesh, global_params, param_resolver, *, positions: np.ndarray,
                         index_map: Dict[int, int], tilts: np.ndarray | None = None, ctx=None) -> float:
    return compute_energy_and_gradient_array(mesh, global_params, param_resolver, positions=positions,
                                             index_map=index_map, grad_arr=None, tilts=tilts,
                                             tilt_grad_arr=None, ctx=ctx)


def compute_energy_and_gradient(mesh, global_params, param_resolver, *, compute_gradient: bool = True):
    """Legacy dict API (tilt_smoothness.py:200-237)."""
    positions = mesh.positions_view()
    tg = np.zeros_like(positions) if compute_gradient else None
    E = compute_energy_and_gradient_array(mesh, global_params, param_resolver, positions=positions,
                                          index_map=mesh.vertex_index_to_row, grad_arr=None, tilts=None,
                                          tilt_grad_arr=tg)
    if not compute_gradient:
        return float(E), {}
    tilt = {int(v): tg[r].copy() for r, v in enumerate(mesh.vertex_ids) if np.any(tg[r])}
    return float(E), {}, tilt


__all__ = ["compute_energy_and_gradient", "compute_energy_and_gradient_array", "compute_energy_array"]
