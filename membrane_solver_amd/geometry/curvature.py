"""Curvature fields on the device, behind the reference's geometry/curvature.py entry points.

``compute_angle_defects`` (geometry/curvature.py:335-403) and ``compute_curvature_fields`` (:404-448) keep the
reference's signatures; the arithmetic runs in one tile-kernel pass (``ms_angle_defects`` / ``ms_curvature_fields``)
on the mesh's device mirror.  Nothing here falls back to the CPU: without the HIP library the calls raise.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Dict

import numpy as np

from .mesh import mirror_for


@dataclass
class CurvatureFields:
    """The reference's dataclass of the same name (geometry/curvature.py:18-27)."""

    mean_curvature_normal: np.ndarray
    mean_curvature: np.ndarray
    mixed_area: np.ndarray
    angle_defect: np.ndarray
    gaussian_curvature: np.ndarray
    principal_curvatures: np.ndarray


def _device(mesh, positions):
    return mirror_for(mesh).sync(positions=None if positions is mesh.positions_view() else positions)


def compute_angle_defects(mesh, positions: np.ndarray, index_map: Dict[int, int]) -> np.ndarray:
    """2 pi - sum of incident triangle angles per vertex, 0 on boundary rows (geometry/curvature.py:335-403)."""
    _ = index_map
    return _device(mesh, positions).angle_defects()


def compute_curvature_fields(mesh, positions: np.ndarray, index_map: Dict[int, int]) -> CurvatureFields:
    """Mean-curvature normal K / (2 A), H = |.|, mixed-Voronoi area, angle defect, K_G = defect / A and the principal
    curvatures H +- sqrt(max(H^2 - K_G, 0)) (geometry/curvature.py:404-448)."""
    _ = index_map
    f = _device(mesh, positions).curvature_fields()
    return CurvatureFields(mean_curvature_normal=f["mean_curvature_normal"], mean_curvature=f["mean_curvature"],
                           mixed_area=f["mixed_area"], angle_defect=f["angle_defect"],
                           gaussian_curvature=f["gaussian_curvature"],
                           principal_curvatures=f["principal_curvatures"])


def gauss_bonnet_invariant(mesh, positions: np.ndarray):
    """runtime/diagnostics/gauss_bonnet.py:305-340 for a manifold triangle mesh: G = sum over interior vertices of
    (2 pi - theta_v) + sum over the boundary-loop vertices of (pi - theta_v), theta_v the device's per-vertex angle
    sums -> (G, interior part, boundary part)."""
    theta = _device(mesh, positions).curvature_fields()["angle_sum"]
    isb = np.zeros(len(theta), dtype=bool)
    rows = mesh.vertex_index_to_row
    for vid in getattr(mesh, "boundary_vertex_ids", ()) or ():
        isb[rows[int(vid)]] = True
    k_int = float(np.sum(2.0 * np.pi - theta[~isb]))
    b_tot = float(np.sum(np.pi - theta[isb]))
    return k_int + b_tot, k_int, b_tot


__all__ = ["CurvatureFields", "compute_angle_defects", "compute_curvature_fields", "gauss_bonnet_invariant"]
