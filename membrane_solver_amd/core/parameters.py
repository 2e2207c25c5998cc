"""Minimal host-side parameter containers with the reference's read API.

The hot path only *reads* ``GlobalParameters.get`` and
``ParameterResolver.get`` (core/parameters/global_parameters.py:4-118,
core/parameters/resolver.py:17-21); the reference's own objects can be passed
instead of these (duck typing).  Defaults listed here are the ones the path
reads, with the reference's values.
"""

from __future__ import annotations

_DEFAULTS = {
    "surface_tension": 1.0,
    "volume_stiffness": 1000.0,
    "volume_constraint_mode": "lagrange",
    "volume_projection_during_minimization": True,
    "volume_tolerance": 1e-3,
    "max_zero_steps": 10,
    "step_size_floor": 1e-8,
    "step_size": 1e-3,
    "step_size_mode": "adaptive",
    "intrinsic_curvature": 0.0,
    "bending_modulus": 0.0,
    "bending_energy_model": "helfrich",
    "bending_gradient_mode": "analytic",
    "mesh_quality_auto_repair_enabled": True,
}


class GlobalParameters:
    def __init__(self, initial_params=None):
        self._params = dict(_DEFAULTS)
        if initial_params:
            self.update(initial_params)

    def get(self, key, default=None):
        return self._params.get(key, default)

    def set(self, key, value):
        self._params[key] = value

    def unset(self, key):
        self._params.pop(key, None)

    def update(self, other):
        self._params.update(dict(other))

    def to_dict(self):
        return self._params

    def __contains__(self, key):
        return key in self._params

    def __getattr__(self, name):
        params = self.__dict__.get("_params")
        if params is not None and name in params:
            return params[name]
        raise AttributeError(name)

    def __repr__(self):
        return f"GlobalParameters({self._params!r})"


class ParameterResolver:
    """obj.options[name] if present else global_params.get(name) (resolver.py:17-21)."""

    def __init__(self, global_params):
        self.global_params = global_params

    def get(self, obj, name):
        opts = getattr(obj, "options", None) if obj is not None else None
        if opts and name in opts:
            return opts[name]
        return self.global_params.get(name)
