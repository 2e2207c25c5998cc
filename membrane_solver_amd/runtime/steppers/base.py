"""Stepper base class with the reference's contract (runtime/steppers/base.py:14-52).

``step(mesh, grad, step_size, energy_fn, constraint_enforcer=None,
trial_energy_fn=None) -> (success, next_step_size, accepted_energy)``; on
success the new positions are left in the mesh and ``mesh._version`` is bumped.

On the HIP path direction, trial positions, Armijo tests and the CG history all
live in HBM (``DeviceMesh.step``).  ``device_step`` is what the device-resident
``Minimizer`` calls; ``step`` keeps the reference's host-side signature for
callers that drive a stepper directly: it re-evaluates the gradient on the
device (identical to the ``grad`` passed in when that came from the same
modules), so ``energy_fn`` / ``trial_energy_fn`` are not invoked.
"""

from __future__ import annotations

from abc import ABC, abstractmethod

import numpy as np

from ... import _lib as L
from ...geometry.mesh import mirror_for


def write_back_positions(mesh, dm, mirror=None):
    """Device -> mesh (array, and Vertex objects for a reference Mesh)."""
    pos = dm.get_positions()
    view = mesh.positions_view()
    view[...] = pos
    if hasattr(mesh, "vertices") and isinstance(getattr(mesh, "vertices"), dict):
        for row, vid in enumerate(mesh.vertex_ids):
            mesh.vertices[int(vid)].position[:] = pos[row]
    mesh.increment_version()
    if mirror is not None:
        mirror.mark_device_positions_current()


class BaseStepper(ABC):
    stepper_id: int = L.MS_STEPPER_GD

    def __init__(self, max_iter=10, beta=0.7, c=1e-4, gamma=1.5, alpha_max_factor=10.0):
        self.max_iter = max_iter
        self.beta = beta
        self.c = c
        self.gamma = gamma
        self.alpha_max_factor = alpha_max_factor
        # evaluation reuse level of ms_step (include/membrane_hip.h, ms_stepper_params):
        # 0 re-evaluates everything the reference re-evaluates; 2 skips the passes whose
        # result is already on the device bit for bit (tests assert identical trajectories)
        self.reuse_energy0 = 2
        # set by the Minimizer: every trial is projected onto the target volume before its energy is taken
        # (line_search.py:428-487 with the volume module's enforce_constraint as the enforcer)
        self.enforce_volume = 0
        self._dm = None

    @abstractmethod
    def _extra(self) -> dict:
        ...

    def _max_iter_for(self, mesh) -> int:
        return int(self.max_iter)

    def device_step(self, dm, mesh, step_size: float, tol: float = 0.0):
        """Run one step on the device.  -> device.StepResult"""
        if self._dm is not dm:
            self._dm = dm
        gp = getattr(mesh, "global_parameters", None)
        edge_fraction = float((gp.get("shape_step_edge_fraction", 0.0) if gp is not None else 0.0) or 0.0)
        return dm.step(stepper=self.stepper_id, step_size=step_size, tol=tol,
                       max_iter=self._max_iter_for(mesh), beta=self.beta, c=self.c, gamma=self.gamma,
                       alpha_max_factor=self.alpha_max_factor, edge_fraction=edge_fraction,
                       reuse_energy0=self.reuse_energy0, enforce_volume=int(self.enforce_volume), **self._extra())

    def step(self, mesh, grad, step_size, energy_fn=None, constraint_enforcer=None,
             trial_energy_fn=None):
        _ = (grad, energy_fn, trial_energy_fn)
        if constraint_enforcer is not None:
            # the enforcer of the hot path is the volume module's projection (the only constraint in scope): with
            # volume_projection_during_minimization on it runs on the device inside every trial
            gp = getattr(mesh, "global_parameters", None)
            self.enforce_volume = int(bool(gp is not None and gp.get("volume_projection_during_minimization", True)
                                           and gp.get("volume_constraint_mode", "lagrange") == "lagrange"
                                           and getattr(mesh, "bodies", None)))
        mir = mirror_for(mesh)
        dm = mir.sync()
        r = self.device_step(dm, mesh, float(step_size))
        if r.success and not r.converged:
            write_back_positions(mesh, dm, mir)
        return bool(r.success), float(r.next_step), float(r.energy)

    def reset(self):
        if self._dm is not None:
            self._dm.reset_stepper()

    def __repr__(self) -> str:  # pragma: no cover
        params = ", ".join(f"{k}={v!r}" for k, v in vars(self).items() if not k.startswith("_"))
        return f"{self.__class__.__name__}({params})"


def as_array(x):
    return np.asarray(x, dtype=np.float64)
