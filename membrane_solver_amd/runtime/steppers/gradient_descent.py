"""Gradient descent with Armijo backtracking on the HIP path
(runtime/steppers/gradient_descent.py:17-84): d = -g, line search parameters
max_iter=10, beta=0.7, c=1e-4, gamma=1.5, alpha_max_factor=10;
``gp["shape_line_search_max_iter"]`` overrides max_iter (:45-50)."""

from __future__ import annotations

from ... import _lib as L
from .base import BaseStepper


class GradientDescent(BaseStepper):
    stepper_id = L.MS_STEPPER_GD

    def __init__(self, max_iter: int = 10, beta: float = 0.7, c: float = 1e-4, gamma: float = 1.5,
                 alpha_max_factor: float = 10.0) -> None:
        super().__init__(max_iter, beta, c, gamma, alpha_max_factor)

    def _extra(self) -> dict:
        return {}

    def _max_iter_for(self, mesh) -> int:
        gp = getattr(mesh, "global_parameters", None)
        if gp is None:
            return int(self.max_iter)
        return int(gp.get("shape_line_search_max_iter", self.max_iter) or self.max_iter)
