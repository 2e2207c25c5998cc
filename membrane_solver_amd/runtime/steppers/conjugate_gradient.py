"""Conjugate gradient with per-row Polak-Ribiere beta on the HIP path
(runtime/steppers/conjugate_gradient.py:17-119): one beta per vertex row,
restart every ``restart_interval`` accepted steps, rows with beta < 0 reset to
steepest descent, fixed rows zeroed, history updated on accepted steps only.
``precondition=True`` (:74-76) builds the direction from the row-normalised gradient
g_i/(|g_i|+1e-8); history and Armijo slope keep the raw gradient (ms_stepper_params.precondition:
unfused direction pass, unqueued trials)."""

from __future__ import annotations

from ... import _lib as L
from .base import BaseStepper


class ConjugateGradient(BaseStepper):
    stepper_id = L.MS_STEPPER_CG

    def __init__(self, restart_interval: int = 10, precondition: bool = False, max_iter: int = 10,
                 beta: float = 0.7, c: float = 1e-4, gamma: float = 1.5,
                 alpha_max_factor: float = 10.0) -> None:
        super().__init__(max_iter, beta, c, gamma, alpha_max_factor)
        self.restart_interval = restart_interval
        self.precondition = bool(precondition)

    def _extra(self) -> dict:
        return {"restart_interval": int(self.restart_interval), "precondition": int(self.precondition)}
