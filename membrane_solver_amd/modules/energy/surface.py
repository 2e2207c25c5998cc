"""Surface-tension energy plugin on the HIP path.

Drop-in for the reference's modules/energy/surface.py:100-239 (array API):
E = sum_f gamma_f A_f and dE/dx accumulated into ``grad_arr``; gamma comes from
``mesh.get_facet_parameter_array("surface_tension")`` exactly as there
(:112).  Unlike the reference there is no NumPy fallback: a kernel failure
raises ``MembraneHipError``.
"""

from __future__ import annotations

from typing import Dict

import numpy as np

from ... import _lib as L
from ._common import evaluate_single_module


def compute_energy_and_gradient_array(mesh, global_params, param_resolver, *, positions: np.ndarray,
                                      index_map: Dict[int, int], grad_arr: np.ndarray) -> float:
    _ = (param_resolver, index_map)
    e, _mir = evaluate_single_module(mesh, global_params, modules=L.MS_MOD_SURFACE,
                                     positions=positions, grad_arr=grad_arr)
    return float(e[0])


def compute_energy_and_gradient(mesh, global_params, param_resolver, *, compute_gradient: bool = True):
    """Dict wrapper (modules/energy/surface.py:242-276)."""
    positions = mesh.positions_view()
    grad_arr = np.zeros_like(positions)
    E = compute_energy_and_gradient_array(mesh, global_params, param_resolver, positions=positions,
                                          index_map=mesh.vertex_index_to_row, grad_arr=grad_arr)
    if not compute_gradient:
        return float(E), {}
    grad = {int(vid): grad_arr[row].copy() for row, vid in enumerate(mesh.vertex_ids)
            if np.any(grad_arr[row])}
    return float(E), grad


__all__ = ["compute_energy_and_gradient", "compute_energy_and_gradient_array"]
