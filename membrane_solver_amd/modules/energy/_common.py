"""Shared plumbing of the HIP-backed energy plugins (host-array seam).

Each plugin keeps the reference's signature
``compute_energy_and_gradient_array(mesh, global_params, param_resolver, *,
positions, index_map, grad_arr) -> float`` (runtime/evaluation_manager.py:88-124
filters kwargs to this signature) and accumulates into ``grad_arr``.  The data
path is: positions H2D -> tiled HIP kernels -> gradient D2H.  For a whole
minimisation use ``membrane_solver_amd.runtime.minimizer.Minimizer``, which
keeps all state in HBM and never calls these per-module functions.
"""

from __future__ import annotations

import numpy as np

from ... import _lib as L
from ...geometry.mesh import mirror_for


def bending_model(global_params) -> str:
    """modules/energy/bending_params.py:19-22."""
    model = str(global_params.get("bending_energy_model", "helfrich") or "helfrich").lower().strip()
    return "helfrich" if model == "helfrich" else "willmore"


def bending_gradient_mode(global_params) -> str:
    """modules/energy/bending_params.py:25-33 (finite_difference is out of scope)."""
    mode = str(global_params.get("bending_gradient_mode", "analytic") or "analytic").lower().strip()
    if mode in {"fd", "finite_difference"}:
        raise L.MembraneHipError(
            "bending_gradient_mode=finite_difference is a debug mode of the reference and is "
            "not provided by the HIP path")
    return "analytic" if mode == "analytic" else "approx"


def evaluate_single_module(mesh, global_params, *, modules: int, positions, grad_arr,
                           want_grad: bool = True, **params):
    """Run one module mask on the device at ``positions``; add into ``grad_arr``."""
    mir = mirror_for(mesh)
    dm = mir.sync(positions=None if positions is mesh.positions_view() else positions)
    if modules & L.MS_MOD_SURFACE:
        mir.upload_surface_tension()
    if modules & L.MS_MOD_BENDING:
        mir.upload_bending_params(global_params, bending_model(global_params))
    dm.set_params(modules=modules, **params)
    if want_grad and grad_arr is not None:
        e, g = dm.energy_and_gradient(want_grad=True, raw=True)
        np.add(grad_arr, g, out=grad_arr)
    else:
        e = dm.energy()
    return e, mir
