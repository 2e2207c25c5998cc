"""Rim slope matching energy plugin (outer leaflet) -- only its switched-off state is on the HIP path.

The reference's modules/energy/rim_slope_match_out.py:352-382 returns 0.0 when no rim / outer group is configured or
when ``rim_slope_match_strength`` is 0 (the setting of the caveolin ``kozlov_1disk_3d_*`` decks, which use the module
name for the CONSTRAINT of the same name).  With a non-zero strength the module couples rim tilts to the local slope
of the outer shell; that is outside the hot path and raises.
"""

from __future__ import annotations

from typing import Dict

import numpy as np

from ... import _lib as L

USES_TILT_LEAFLETS = True


def _strength(param_resolver, global_params) -> float:
    val = param_resolver.get(None, "rim_slope_match_strength") if param_resolver is not None else None
    if val is None and global_params is not None:
        val = global_params.get("rim_slope_match_strength")
    return float(val or 0.0)


def constant_energy(mesh, global_params, param_resolver=None) -> float:
    if _strength(param_resolver, global_params) != 0.0:
        raise L.MembraneHipError("rim_slope_match_out with a non-zero rim_slope_match_strength is outside the HIP hot path")
    return 0.0


def compute_energy_and_gradient_array(mesh, global_params, param_resolver, *, positions: np.ndarray,
                                      index_map: Dict[int, int], grad_arr: np.ndarray | None,
                                      tilts_in: np.ndarray | None = None, tilts_out: np.ndarray | None = None,
                                      tilt_in_grad_arr: np.ndarray | None = None,
                                      tilt_out_grad_arr: np.ndarray | None = None) -> float:
    _ = (positions, index_map, grad_arr, tilts_in, tilts_out, tilt_in_grad_arr, tilt_out_grad_arr)
    return constant_energy(mesh, global_params, param_resolver)


def compute_energy_array(mesh, global_params, param_resolver, *, positions: np.ndarray, index_map: Dict[int, int],
                         tilts_in: np.ndarray | None = None, tilts_out: np.ndarray | None = None) -> float:
    return constant_energy(mesh, global_params, param_resolver)


def compute_energy_and_gradient(mesh, global_params, param_resolver, *, compute_gradient: bool = True):
    return float(constant_energy(mesh, global_params, param_resolver)), {}, {}


__all__ = ["compute_energy_and_gradient", "compute_energy_and_gradient_array", "compute_energy_array"]
