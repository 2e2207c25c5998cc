"""Helfrich bending with Kozlov-Hamm tilt-splay coupling on the HIP path.

Drop-in for the reference's modules/energy/bending_tilt.py:151-482:
    E = 1/2 sum_f sum_k kappa_k (2 H_k - c0_k + div_f t)^2 va_eff[f,k]
on the bending module's discretisation (cotan curvature vectors, mixed-Voronoi areas with
the boundary redistribution), ``div_f t`` the P1 facet divergence of the tangent tilt
field.  The shape gradient treats div t as constant in x (bending_tilt.py:24-28) and is the
bending back-propagation with term -> base + div_eff; the tilt gradient is exact.
``grad_arr`` / ``tilt_grad_arr`` are ACCUMULATED into; ``grad_arr=None`` gives the
energy (+ tilt gradient) only, as the reference's early-return branch (:258-268).
"""

from __future__ import annotations

from typing import Dict

import numpy as np

from ... import _lib as L
from ...geometry.mesh import mirror_for
from ._common import bending_gradient_mode

USES_TILT = True


def _device(mesh, global_params, positions, tilts):
    mir = mirror_for(mesh)
    dm = mir.sync(positions=None if positions is mesh.positions_view() else positions)
    if tilts is None:
        tilts = mesh.tilts_view()
    tilts = np.ascontiguousarray(tilts, dtype=np.float64)
    if tilts.shape != (len(mesh.vertex_ids), 3):
        raise ValueError("tilts must have shape (N_vertices, 3)")
    dm.set_tilts(tilts, float(global_params.get("tilt_rigidity", 0.0) or 0.0))
    mir._tilt_key = None  # a foreign array may have been uploaded
    mir.upload_bending_params(global_params, "helfrich")  # bending_tilt.py:212-215: always Helfrich
    mode = bending_gradient_mode(global_params)
    dm.set_params(modules=L.MS_MOD_BENDING_TILT, bending_model=L.MS_BEND_HELFRICH,
                  bending_grad_mode=L.MS_GRAD_ANALYTIC if mode == "analytic" else L.MS_GRAD_APPROX)
    return dm


def compute_energy_and_gradient_array(mesh, global_params, param_resolver, *, positions: np.ndarray,
                                      index_map: Dict[int, int], grad_arr: np.ndarray | None,
                                      ctx=None, tilts: np.ndarray | None = None,
                                      tilt_grad_arr: np.ndarray | None = None) -> float:
    _ = (param_resolver, index_map, ctx)
    tri, _f = mesh.triangle_row_cache()
    if tri is None or len(tri) == 0:
        return 0.0
    dm = _device(mesh, global_params, positions, tilts)
    if grad_arr is not None:
        e, g = dm.energy_and_gradient(want_grad=True, raw=True)
        grad_arr += g
        E = float(e[1])
        if tilt_grad_arr is not None:
            _e, tg = dm.tilt_energy_and_gradient()
            tilt_grad_arr += tg
        return E
    if tilt_grad_arr is not None:
        E, tg = dm.tilt_energy_and_gradient()
        tilt_grad_arr += tg
        return float(E)
    return float(dm.energy()[1])


def compute_energy_and_gradient(mesh, global_params, param_resolver, *, compute_gradient: bool = True):
    """Legacy dict API (bending_tilt.py:485-524)."""
    positions = mesh.positions_view()
    grad = np.zeros_like(positions)
    tg = np.zeros_like(positions) if compute_gradient else None
    E = compute_energy_and_gradient_array(mesh, global_params, param_resolver, positions=positions,
                                          index_map=mesh.vertex_index_to_row, grad_arr=grad, tilts=None,
                                          tilt_grad_arr=tg)
    if not compute_gradient:
        return float(E), {}
    shape = {int(v): grad[r].copy() for r, v in enumerate(mesh.vertex_ids) if np.any(grad[r])}
    tilt = {int(v): tg[r].copy() for r, v in enumerate(mesh.vertex_ids) if np.any(tg[r])}
    return float(E), shape, tilt


__all__ = ["compute_energy_and_gradient", "compute_energy_and_gradient_array"]
