"""Outer-leaflet bending + tilt-splay coupling energy plugin on the HIP path.

Drop-in for the reference's modules/energy/bending_tilt_out.py (-> bending_tilt_leaflet.py:231-758, div_sign = +1).
    E = 1/2 sum_f sum_k kappa_k (2 H_k - c0_k + s div_f t)^2 va_eff[f,k],  s = +1, H = (K . n)/(2 A_vor) signed;
shape gradient: back-propagation with per-corner area factors + s dE/ddiv d(div)/dx into ``grad_arr``; exact tilt
gradient into ``tilt_out_grad_arr``.  Default options and bending_gradient_mode=analytic only (others raise).
"""

from __future__ import annotations

from typing import Dict

import numpy as np

from . import leaflet_common as _lc

USES_TILT_LEAFLETS = True
_LEAFLET = "out"
_KIND = "bt"


def compute_energy_and_gradient_array(mesh, global_params, param_resolver, *, positions: np.ndarray,
                                      index_map: Dict[int, int], grad_arr: np.ndarray | None, ctx=None,
                                      tilts_in: np.ndarray | None = None, tilts_out: np.ndarray | None = None,
                                      tilt_in_grad_arr: np.ndarray | None = None,
                                      tilt_out_grad_arr: np.ndarray | None = None) -> float:
    _ = (index_map, ctx)
    if _rigidity(param_resolver, global_params) == 0.0:
        return 0.0
    return _lc.evaluate(mesh, global_params, param_resolver, kind=_KIND, leaflet=_LEAFLET, positions=positions,
                        tilts=tilts_in if _LEAFLET == "in" else tilts_out, grad_arr=grad_arr,
                        tilt_grad_arr=tilt_in_grad_arr if _LEAFLET == "in" else tilt_out_grad_arr)


def compute_energy_array(mesh, global_params, param_resolver, *, positions: np.ndarray, index_map: Dict[int, int],
                         tilts_in: np.ndarray | None = None, tilts_out: np.ndarray | None = None, ctx=None) -> float:
    return compute_energy_and_gradient_array(mesh, global_params, param_resolver, positions=positions,
                                             index_map=index_map, grad_arr=None, ctx=ctx, tilts_in=tilts_in,
                                             tilts_out=tilts_out)


def compute_energy_and_gradient(mesh, global_params, param_resolver, *, compute_gradient: bool = True):
    """Dict API of the reference: (E, shape_grad, tilt_grad)."""
    positions = mesh.positions_view()
    g = np.zeros_like(positions)
    tg = np.zeros_like(positions) if compute_gradient else None
    kw = {"tilt_in_grad_arr": tg} if _LEAFLET == "in" else {"tilt_out_grad_arr": tg}
    E = compute_energy_and_gradient_array(mesh, global_params, param_resolver, positions=positions,
                                          index_map=mesh.vertex_index_to_row, grad_arr=g, **kw)
    if not compute_gradient:
        return float(E), {}
    ids = mesh.vertex_ids
    return (float(E), {int(v): g[r].copy() for r, v in enumerate(ids)},
            {int(v): tg[r].copy() for r, v in enumerate(ids)})


def _rigidity(param_resolver, global_params) -> float:
    return 1.0  # the module has no early-out on the modulus (bending_tilt_leaflet.py:231-260)


__all__ = ["compute_energy_and_gradient", "compute_energy_and_gradient_array", "compute_energy_array"]
