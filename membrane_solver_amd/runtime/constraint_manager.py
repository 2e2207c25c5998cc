"""Constraint plugin loader (runtime/constraint_manager.py, reduced to the
in-scope ``volume`` module) and the k == 1 dense KKT projection (:293-301)."""

from __future__ import annotations

import importlib
import logging

import numpy as np

logger = logging.getLogger("membrane_solver")

_PACKAGE = "membrane_solver_amd.modules.constraints"


class ConstraintModuleManager:
    def __init__(self, module_names):
        self.modules = {}
        for name in module_names:
            try:
                self.modules[name] = importlib.import_module(f"{_PACKAGE}.{name}")
            except ImportError as e:
                logger.error("Could not load constraint module '%s': %s", name, e)
                raise

    def get_constraint(self, name):
        if name not in self.modules:
            raise KeyError(f"Constraint module '{name}' not found.")
        return self.modules[name]

    def __contains__(self, name):
        return name in self.modules

    def __getitem__(self, name):
        return self.modules[name]

    def apply_gradient_modifications_array(self, grad_arr, mesh, global_params):
        """Host-array variant of the k == 1 dense branch (constraint_manager.py:293-301)."""
        positions = mesh.positions_view()
        rows = []
        for module in self.modules.values():
            fn = getattr(module, "constraint_gradients_array", None)
            if fn is None:
                continue
            g_list = fn(mesh, global_params, positions=positions, index_map=mesh.vertex_index_to_row)
            if g_list:
                rows.extend(g_list)
        if not rows:
            return
        if len(rows) != 1:
            raise NotImplementedError("only the single dense constraint row (volume) is in scope")
        gC = rows[0]
        norm_sq = float(np.sum(gC * gC))
        if norm_sq > 1e-18:
            lam = float(np.sum(grad_arr * gC)) / norm_sq
            grad_arr -= lam * gC

    def enforce_all(self, mesh, *, context="minimize", global_params=None, **options):
        """Hard projection of every loaded constraint that has one -- here: ``volume`` -- onto its target
        (interface of constraint_manager.py:843-905).

        Inside a minimisation step (``context == "minimize"``) the volume projection only runs when
        ``volume_projection_during_minimization`` is on; with it off the optimizer holds the volume through the
        KKT projection of the gradient alone.  After mesh operations and at the end of ``minimize`` (the other
        contexts) it always runs, with ``force_projection=True``."""
        during_step = context == "minimize"
        step_projection = True if global_params is None else bool(
            global_params.get("volume_projection_during_minimization", True))
        options.pop("force_projection", None)
        volume = self.modules.get("volume")
        if volume is None or (during_step and not step_projection):
            return
        volume.enforce_constraint(mesh, force_projection=True, context=context, global_params=global_params,
                                  **options)
