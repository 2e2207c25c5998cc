"""Plugin loader with the reference's interface (runtime/energy_manager.py:10-33):
``EnergyModuleManager(module_names).get_module(name)`` imports
``membrane_solver_amd.modules.energy.<name>``."""

from __future__ import annotations

import importlib
import logging

logger = logging.getLogger("membrane_solver")

_PACKAGE = "membrane_solver_amd.modules.energy"


class EnergyModuleManager:
    def __init__(self, module_names):
        self.modules = {}
        for name in module_names:
            try:
                self.modules[name] = importlib.import_module(f"{_PACKAGE}.{name}")
                logger.info("Loaded energy module: %s", name)
            except ImportError as e:
                logger.error("Could not load energy module '%s': %s", name, e)
                raise

    def get_module(self, mod):
        if mod not in self.modules:
            raise KeyError(f"Energy module '{mod}' not found.")
        return self.modules[mod]

    def get_energy_function(self, mod, type_index):
        module = self.get_module(mod)
        return getattr(module, "compute_energy_and_gradient", None)
