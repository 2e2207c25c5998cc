"""Helfrich / Willmore bending plugin on the HIP path.

Drop-in for modules/energy/bending.py:62-181 of the reference: cotan curvature
data (geometry/curvature.py:113-332), effective areas
(bending_utils.py:37-171), per-vertex density (bending.py:117-161) and the
analytic back-propagated shape gradient (bending_gradient.py:17-175) or the
``approx`` Laplacian gradient (bending.py:163-167).  ``finite_difference`` mode
is out of scope (debug only).
"""

from __future__ import annotations

from typing import Dict

import numpy as np

from ... import _lib as L
from ._common import bending_gradient_mode, bending_model, evaluate_single_module


def _params(global_params):
    model = bending_model(global_params)
    mode = bending_gradient_mode(global_params)
    return dict(bending_model=L.MS_BEND_HELFRICH if model == "helfrich" else L.MS_BEND_WILLMORE,
                bending_grad_mode=L.MS_GRAD_ANALYTIC if mode == "analytic" else L.MS_GRAD_APPROX)


def compute_energy_and_gradient_array(mesh, global_params, param_resolver, *, positions: np.ndarray,
                                      index_map: Dict[int, int], grad_arr: np.ndarray) -> float:
    _ = (param_resolver, index_map)
    tri, _f = mesh.triangle_row_cache()
    if tri is None or len(tri) == 0:
        return 0.0
    if bending_gradient_mode(global_params) == "approx" and grad_arr is not None:
        # bending.py:165-166 zeroes the boundary rows of the WHOLE accumulated array
        scratch = np.zeros_like(grad_arr)
        e, mir = evaluate_single_module(mesh, global_params, modules=L.MS_MOD_BENDING,
                                        positions=positions, grad_arr=scratch, **_params(global_params))
        from ...geometry.mesh import _boundary_mask_of

        bmask = _boundary_mask_of(mesh, grad_arr.shape[0])
        grad_arr += scratch
        if bmask.any():
            grad_arr[bmask] = 0.0
        return float(e[1])
    e, _mir = evaluate_single_module(mesh, global_params, modules=L.MS_MOD_BENDING,
                                     positions=positions, grad_arr=grad_arr, **_params(global_params))
    return float(e[1])


def compute_energy_array(mesh, global_params, positions, index_map) -> float:
    """Energy-only entry (bending.py:62-87); the reference returns a per-vertex
    array that EvaluationManager sums (_coerce_energy_value :82-86) -- the
    device reduces it, so the sum is returned."""
    _ = index_map
    tri, _f = mesh.triangle_row_cache()
    if tri is None or len(tri) == 0:
        return 0.0
    e, _mir = evaluate_single_module(mesh, global_params, modules=L.MS_MOD_BENDING, positions=positions,
                                     grad_arr=None, want_grad=False, **_params(global_params))
    return float(e[1])


def compute_total_energy(mesh, global_params, positions, index_map) -> float:
    return compute_energy_array(mesh, global_params, positions, index_map)


def compute_energy_and_gradient(mesh, global_params, param_resolver, *, compute_gradient: bool = True):
    positions = mesh.positions_view()
    grad_arr = np.zeros_like(positions)
    E = compute_energy_and_gradient_array(mesh, global_params, param_resolver, positions=positions,
                                          index_map=mesh.vertex_index_to_row, grad_arr=grad_arr)
    if not compute_gradient:
        return float(E), {}
    return float(E), {int(vid): grad_arr[row].copy() for row, vid in enumerate(mesh.vertex_ids)
                      if np.any(grad_arr[row])}


__all__ = ["compute_energy_and_gradient", "compute_energy_and_gradient_array", "compute_energy_array"]
