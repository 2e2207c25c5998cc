"""Outer-leaflet disk tilt target (soft profile enforcement) plugin on the HIP path.

Drop-in for the reference's modules/energy/tilt_disk_target_out.py:160-286.
    E = 1/2 k int |t_out - theta(r) r_hat|^2 dA over the rows tagged ``tilt_disk_target_group_out``,
theta(r) = theta_B I1(lambda r)/I1(lambda R) (or theta_B r/R); shape gradient coeff_f dA/dx into ``grad_arr``, tilt
gradient k diff_v A_v into ``tilt_out_grad_arr``.  ``tilt_disk_target_normal`` must be given (the SVD plane fit
of the reference raises here).
"""

from __future__ import annotations

from typing import Dict

import numpy as np

from . import leaflet_common as _lc

USES_TILT_LEAFLETS = True
_LEAFLET = "out"
_KIND = "disk"


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
    return 1.0  # the early-outs are in leaflet_common.disk_target_params (tilt_disk_target_in.py:175-191)


__all__ = ["compute_energy_and_gradient", "compute_energy_and_gradient_array", "compute_energy_array"]
