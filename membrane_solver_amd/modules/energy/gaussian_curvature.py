"""Gaussian curvature (Helfrich Gaussian modulus) energy plugin.

Drop-in for the reference's modules/energy/gaussian_curvature.py:104-176.  Closed surfaces: with a constant
``gaussian_modulus`` the energy is the topological constant 2 pi kappa_bar chi (Gauss-Bonnet), chi = V - E + F from the
triangle rows; the optional defect check (``gaussian_curvature_check_defects``) sums the per-vertex angle defects
computed on the device (``ms_angle_defects``, geometry/curvature.py:335-403).  Surfaces with boundary loops
(:128-143): kappa_bar times the Gauss-Bonnet invariant G = sum_interior (2 pi - theta_v) + sum_boundary (pi - theta_v)
(runtime/diagnostics/gauss_bonnet.py:260-340) from the device's per-vertex angle sums (``ms_curvature_fields``).  The
shape gradient is zero in both cases, as in the reference.  Facet filters (``gauss_bonnet_exclude``) and the strict
topology check stay outside the hot path and raise.
"""

from __future__ import annotations

import logging
from typing import Dict

import numpy as np

from ... import _lib as L
from ...geometry.mesh import mirror_for

logger = logging.getLogger("membrane_solver")


def _gaussian_modulus(global_params) -> float:
    return float(global_params.get("gaussian_modulus", 0.0) or 0.0)


def euler_characteristic(mesh) -> int:
    """V - E + F (gaussian_curvature.py:41-43); the reference Mesh carries the counts, an array mesh its rows."""
    if all(hasattr(mesh, a) for a in ("vertices", "edges", "facets")) and isinstance(mesh.vertices, dict):
        return int(len(mesh.vertices) - len(mesh.edges) + len(mesh.facets))
    tri, _f = mesh.triangle_row_cache()
    tri = np.asarray(tri, dtype=np.int64)
    e = np.sort(np.concatenate([tri[:, [0, 1]], tri[:, [1, 2]], tri[:, [2, 0]]]), axis=1)
    n_edges = np.unique(e, axis=0).shape[0] if e.size else 0
    return int(len(mesh.vertex_ids) - n_edges + tri.shape[0])


def has_boundary(mesh) -> bool:
    return bool(len(getattr(mesh, "boundary_vertex_ids", ()) or ()))


def constant_energy(mesh, global_params, positions=None) -> float:
    """The module's energy; a topological constant on closed surfaces.  With boundary loops it depends on the
    positions through the boundary angle sums (``positions`` None = the mesh's own)."""
    kappa_bar = _gaussian_modulus(global_params)
    if kappa_bar == 0.0:
        return 0.0
    if bool(global_params.get("gaussian_curvature_strict_topology", False)):
        raise L.MembraneHipError("gaussian_curvature_strict_topology is outside the HIP hot path")
    if has_boundary(mesh):
        from ...geometry.curvature import gauss_bonnet_invariant

        pos = mesh.positions_view() if positions is None else positions
        g_total, _k_int, _b_tot = gauss_bonnet_invariant(mesh, pos)
        return float(kappa_bar * g_total)
    return float(2.0 * np.pi * kappa_bar * euler_characteristic(mesh))


def compute_energy_and_gradient_array(mesh, global_params, param_resolver, *, positions: np.ndarray,
                                      index_map: Dict[int, int], grad_arr: np.ndarray) -> float:
    _ = (param_resolver, index_map, grad_arr)
    energy = constant_energy(mesh, global_params, positions)
    if energy != 0.0 and not has_boundary(mesh) and bool(global_params.get("gaussian_curvature_check_defects", False)):
        dm = mirror_for(mesh).sync(positions=None if positions is mesh.positions_view() else positions)
        defect_sum = float(np.sum(dm.angle_defects()))
        target = float(2.0 * np.pi * euler_characteristic(mesh))
        if abs(defect_sum - target) > 1e-6:
            logger.warning("Gaussian curvature defect sum mismatch: sum(defect)=%.6e, 2 pi chi=%.6e (|d|=%.3e). "
                           "Check for non-manifold topology.", defect_sum, target, abs(defect_sum - target))
    return energy


def compute_energy_and_gradient(mesh, global_params, param_resolver, *, compute_gradient: bool = True):
    positions = mesh.positions_view()
    energy = compute_energy_and_gradient_array(mesh, global_params, param_resolver, positions=positions,
                                               index_map=mesh.vertex_index_to_row, grad_arr=np.zeros_like(positions))
    return float(energy), {}


__all__ = ["compute_energy_and_gradient", "compute_energy_and_gradient_array", "constant_energy"]
