"""Volume penalty energy plugin on the HIP path.

Drop-in for modules/energy/volume.py:94-128: E = 1/2 k (V - V0)^2 per body in
``volume_constraint_mode == "penalty"``, zero otherwise.  V and dV/dx follow
geometry/body.py:104-123,150-190.  One body per mesh.
"""

from __future__ import annotations

from typing import Dict

import numpy as np

from ... import _lib as L
from ...geometry.mesh import mirror_for
from ._common import evaluate_single_module


def body_penalty_params(mesh, global_params, param_resolver):
    """(k, V0) of the mesh's single body, resolved as volume.py:108-117 does."""
    bodies = getattr(mesh, "bodies", None) or {}
    if not bodies:
        return None
    body = next(iter(bodies.values()))
    k = param_resolver.get(body, "volume_stiffness") if param_resolver is not None else None
    if k is None:
        k = global_params.get("volume_stiffness")
    V0 = body.target_volume if body.target_volume is not None else (body.options or {}).get("target_volume", 0)
    return float(k), float(V0)


def compute_energy_and_gradient_array(mesh, global_params, param_resolver, *, positions: np.ndarray,
                                      index_map: Dict[int, int], grad_arr: np.ndarray) -> float:
    _ = index_map
    if global_params.get("volume_constraint_mode", "lagrange") != "penalty":
        return 0.0
    mirror_for(mesh).sync()  # validates the single-body restriction
    kv = body_penalty_params(mesh, global_params, param_resolver)
    if kv is None:
        return 0.0
    e, _mir = evaluate_single_module(mesh, global_params, modules=L.MS_MOD_VOLUME_PENALTY,
                                     positions=positions, grad_arr=grad_arr,
                                     volume_stiffness=kv[0], target_volume=kv[1])
    return float(e[2])


def compute_energy_and_gradient(mesh, global_params, param_resolver, *, compute_gradient: bool = True):
    positions = mesh.positions_view()
    grad_arr = np.zeros_like(positions)
    E = compute_energy_and_gradient_array(mesh, global_params, param_resolver, positions=positions,
                                          index_map=mesh.vertex_index_to_row, grad_arr=grad_arr)
    if not compute_gradient:
        return float(E), {}
    return float(E), {int(vid): grad_arr[row].copy() for row, vid in enumerate(mesh.vertex_ids)
                      if np.any(grad_arr[row])}


__all__ = ["compute_energy_and_gradient", "compute_energy_and_gradient_array"]
