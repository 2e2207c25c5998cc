"""Device-resident Minimizer with the reference's entry points.

Mirrors ``runtime/minimizer.py`` of the reference for the hot-path scope:
``Minimizer(mesh, global_params, stepper, energy_manager, constraint_manager,
energy_modules=None, constraint_modules=None, step_size=1e-3, tol=1e-6,
quiet=False)`` with ``minimize(n_steps, callback=None) -> dict``,
``compute_energy()``, ``compute_energy_and_gradient_array()``,
``compute_energy_breakdown()``, ``refresh_modules()``, ``reset_soa_caches()``
and the attributes ``step_size``, ``stepper``, ``mesh``.

One iteration (minimizer.py:1230-1515) =
  energy + gradient over all modules            (:1314, evaluation_manager.py:134-151)
  volume-constraint KKT projection, fixed rows  (:982-990, constraint_manager.py:293-301)
  convergence test |g| < tol                    (:1324)
  stepper.step: direction + Armijo line search  (:1374, line_search.py:267-426)
  zero-step / reset bookkeeping                 (:1439-1464)
  Lagrange volume-drift check -> projection     (:1478-1513)
All array work runs in libmembrane_hip.so with positions, gradient, direction,
CG history and trial positions resident in HBM; the host sees a handful of
scalars per step.  ``Vertex.position`` objects / the mesh position array are
written back once at the end (or before each ``callback``).

On the device as well: the tilt relaxation at the top of every iteration
(``tilt_solve_mode`` nested / coupled, single field and two leaflets:
ms_relax_tilts / ms_relax_leaflet_tilts, minimizer.py:1237-1307), the tilt
modules (``tilt``, ``bending_tilt``, ``tilt_smoothness`` and their ``_in`` /
``_out`` leaflet forms, ``tilt_disk_target_in/out``), ``gaussian_curvature``
and the switched-off ``rim_slope_match_out`` as host-side constants.

Out of scope here (raises MembraneHipError): every other energy module,
constraints other than ``volume``, the benchmark toggles of the leaflet modules
(modules/energy/leaflet_common._UNSUPPORTED_KEYS), the reference's auto
mesh-quality repair hook.
"""

from __future__ import annotations

import logging
from typing import Callable, Dict, List, Optional

import numpy as np

from .. import _lib as L
from ..core.parameters import ParameterResolver
from .steppers.base import BaseStepper
from ..geometry.mesh import mirror_for
from ..modules.energy import leaflet_common as _lc
from ..modules.energy._common import bending_gradient_mode, bending_model
from ..modules.energy.volume import body_penalty_params
from .steppers.base import write_back_positions

logger = logging.getLogger("membrane_solver")

_ENERGY_BITS = {"surface": L.MS_MOD_SURFACE, "bending": L.MS_MOD_BENDING, "volume": L.MS_MOD_VOLUME_PENALTY,
                "tilt": L.MS_MOD_TILT, "bending_tilt": L.MS_MOD_BENDING_TILT,
                "tilt_smoothness": L.MS_MOD_TILT_SMOOTH,
                "tilt_in": L.MS_MOD_TILT_IN, "tilt_out": L.MS_MOD_TILT_OUT,
                "tilt_smoothness_in": L.MS_MOD_TILT_SMOOTH_IN, "tilt_smoothness_out": L.MS_MOD_TILT_SMOOTH_OUT,
                "bending_tilt_in": L.MS_MOD_BENDING_TILT_IN, "bending_tilt_out": L.MS_MOD_BENDING_TILT_OUT,
                "tilt_disk_target_in": L.MS_MOD_TILT_DISK_TARGET_IN,
                "tilt_disk_target_out": L.MS_MOD_TILT_DISK_TARGET_OUT,
                # host-side constants, no kernel: a topological constant on closed surfaces; a module that is only
                # accepted in its switched-off state (strength 0, as in the caveolin decks)
                "gaussian_curvature": 0, "rim_slope_match_out": 0}
_ENERGY_SLOT = {"surface": 0, "bending": 1, "volume": 2, "tilt": 3, "bending_tilt": 1, "tilt_smoothness": 3,
                "tilt_in": 3, "tilt_out": 3, "tilt_smoothness_in": 3, "tilt_smoothness_out": 3,
                "bending_tilt_in": 1, "bending_tilt_out": 1, "tilt_disk_target_in": 3, "tilt_disk_target_out": 3,
                "gaussian_curvature": None, "rim_slope_match_out": None}
_SINGLE_TILT_BITS = L.MS_MOD_TILT | L.MS_MOD_BENDING_TILT | L.MS_MOD_TILT_SMOOTH
_LEAFLET_BT_BITS = L.MS_MOD_BENDING_TILT_IN | L.MS_MOD_BENDING_TILT_OUT
_LEAFLET_BITS = (L.MS_MOD_TILT_IN | L.MS_MOD_TILT_OUT | L.MS_MOD_TILT_SMOOTH_IN | L.MS_MOD_TILT_SMOOTH_OUT
                 | _LEAFLET_BT_BITS | L.MS_MOD_TILT_DISK_TARGET_IN | L.MS_MOD_TILT_DISK_TARGET_OUT)
_TILT_BITS = _SINGLE_TILT_BITS | _LEAFLET_BITS
# scalar slot of every module that shares energies[3]
_TILT_SCALAR = {"tilt": L.MS_S_ETILT, "tilt_smoothness": L.MS_S_ETS, "tilt_in": L.MS_S_ETILT_IN,
                "tilt_out": L.MS_S_ETILT_OUT, "tilt_smoothness_in": L.MS_S_ETS_IN,
                "tilt_smoothness_out": L.MS_S_ETS_OUT, "tilt_disk_target_in": L.MS_S_EDT_IN,
                "tilt_disk_target_out": L.MS_S_EDT_OUT}
_BEND_SCALAR = {"bending": L.MS_S_EBEND, "bending_tilt": L.MS_S_EBT, "bending_tilt_in": L.MS_S_EBT_IN,
                "bending_tilt_out": L.MS_S_EBT_OUT}


class GradientRows:
    """Lazy stand-in for the reference's ``{vertex_id: grad_row}`` dict of non-zero rows
    (minimizer.py:1067-1073); materialised on first dict-style access."""

    def __init__(self, mesh, grad_arr):
        """grad_arr: the (nv,3) array, or a zero-argument callable that downloads it on first use
        (minimize(sync_mesh=False) keeps the 24 B/vertex D2H copy out of the loop that way; read
        it before stepping again: it fetches the gradient then resident on the device)."""
        self._array = None if callable(grad_arr) else grad_arr
        self._fetch = grad_arr if callable(grad_arr) else None
        self._mesh = mesh
        self._dict = None

    @property
    def array(self) -> np.ndarray:
        if self._array is None:
            self._array = self._fetch()
            self._fetch = None
        return self._array

    def _materialise(self):
        if self._dict is None:
            nz = np.flatnonzero(np.any(self.array != 0.0, axis=1))
            ids = self._mesh.vertex_ids
            self._dict = {int(ids[r]): self.array[r].copy() for r in nz}
        return self._dict

    def __getitem__(self, k):
        return self._materialise()[k]

    def __iter__(self):
        return iter(self._materialise())

    def __len__(self):
        return len(self._materialise())

    def items(self):
        return self._materialise().items()

    def keys(self):
        return self._materialise().keys()

    def values(self):
        return self._materialise().values()

    def __contains__(self, k):
        return k in self._materialise()


class Minimizer:
    """Coordinate the optimisation loop for a mesh, on the GPU."""

    def __init__(self, mesh, global_params, stepper, energy_manager, constraint_manager,
                 energy_modules: Optional[List[str]] = None,
                 constraint_modules: Optional[List[str]] = None, step_size: float = 1e-3,
                 tol: float = 1e-6, quiet: bool = False, *, device: int = 0, tile_vertices: int = 0,
                 deterministic: Optional[bool] = None):
        self.mesh = mesh
        self.global_params = global_params
        self.energy_manager = energy_manager
        self.constraint_manager = constraint_manager
        self.stepper = stepper
        self.step_size = step_size
        self.tol = tol
        self.quiet = quiet
        self.device = device
        self.tile_vertices = tile_vertices
        # None: library default (LDS-atomic vertex sums unless MS_DETERMINISTIC=1); True: fixed-order
        # sums, bitwise reproducible run to run (ms_set_deterministic)
        self.deterministic = deterministic
        self.max_zero_steps = int(global_params.get("max_zero_steps", 10))
        self.step_size_floor = float(global_params.get("step_size_floor", 1e-8))
        self.param_resolver = ParameterResolver(global_params)
        module_list = energy_modules if energy_modules is not None else mesh.energy_modules
        self.energy_module_names = list(module_list)
        constraint_list = constraint_modules if constraint_modules is not None else mesh.constraint_modules
        self.constraint_module_names = list(constraint_list)
        self.refresh_modules()

    # -- module wiring (minimizer.py:100-118, :235-243) -----------------------
    def refresh_modules(self):
        self.energy_modules = [self.energy_manager.get_module(m) for m in self.energy_module_names]
        for name, mod in zip(self.energy_module_names, self.energy_modules):
            if not hasattr(mod, "compute_energy_and_gradient_array"):
                raise TypeError(f"energy module {name!r} lacks compute_energy_and_gradient_array")
            if name not in _ENERGY_BITS:
                raise L.MembraneHipError(
                    f"energy module {name!r} is outside the HIP hot path (surface, bending, volume, tilt, "
                    "bending_tilt, tilt_smoothness, tilt_in, tilt_out, tilt_smoothness_in, tilt_smoothness_out, "
                    "bending_tilt_in, bending_tilt_out, tilt_disk_target_in, tilt_disk_target_out)")
        self.constraint_modules = [self.constraint_manager.get_constraint(c)
                                   for c in self.constraint_module_names]
        for name in self.constraint_module_names:
            if name != "volume":
                raise L.MembraneHipError(f"constraint module {name!r} is outside the HIP hot path (volume)")
        self._has_enforceable_constraints = any(hasattr(m, "enforce_constraint")
                                                for m in self.constraint_modules)
        self._configured_key = None
        self._device_ahead = False

    def sync_mesh_from_device(self):
        """Write device positions back into the mesh (after minimize(sync_mesh=False))."""
        mir, dm = self._device_nosync()
        write_back_positions(self.mesh, dm, mir)
        self._device_ahead = False

    def _device_nosync(self):
        mir = mirror_for(self.mesh, device=self.device, tile_vertices=self.tile_vertices)
        if mir.dm is None:
            mir.sync()
        return mir, mir.dm

    def reset_soa_caches(self):
        """Forget the device mirror (call after topology surgery, minimizer.py:208)."""
        mir = getattr(self.mesh, "_hip_mirror", None)
        if mir is not None and mir.dm is not None:
            mir.dm.close()
        self.mesh._hip_mirror = None
        self._configured_key = None

    # -- device configuration --------------------------------------------------
    def _device(self):
        gp = self.global_params
        mir = mirror_for(self.mesh, device=self.device, tile_vertices=self.tile_vertices)
        dm = mir.sync()
        if self.deterministic is not None:
            dm.set_deterministic(self.deterministic)
        mods = 0
        self._const_energy = {}
        disk_params = {}
        vol_mode = gp.get("volume_constraint_mode", "lagrange")
        for name in self.energy_module_names:
            if name == "volume":
                if vol_mode == "penalty" and getattr(self.mesh, "bodies", None):
                    mods |= L.MS_MOD_VOLUME_PENALTY
            elif name == "tilt":
                if float(gp.get("tilt_rigidity", 0.0) or 0.0) != 0.0:  # tilt.py:110-112
                    mods |= L.MS_MOD_TILT
            elif name == "tilt_smoothness":
                if float(gp.get("tilt_smoothness_rigidity", 0.0) or 0.0) != 0.0:  # tilt_smoothness.py:258-260
                    mods |= L.MS_MOD_TILT_SMOOTH
            elif name in ("tilt_in", "tilt_out"):
                if _lc.tilt_modulus(self.param_resolver, gp, name[5:]) != 0.0:  # tilt_leaflet.py:41-43
                    mods |= _ENERGY_BITS[name]
            elif name in ("tilt_smoothness_in", "tilt_smoothness_out"):
                if _lc.smoothness_rigidity(self.param_resolver, gp, name[16:]) != 0.0:  # tilt_smoothness_leaflet.py:32-34
                    mods |= _ENERGY_BITS[name]
            elif name == "gaussian_curvature":
                from ..modules.energy import gaussian_curvature as _gc

                self._const_energy[name] = _gc.constant_energy(self.mesh, gp)  # gaussian_curvature.py:117-134
            elif name == "rim_slope_match_out":
                from ..modules.energy import rim_slope_match_out as _rs

                self._const_energy[name] = _rs.constant_energy(self.mesh, gp, self.param_resolver)
            elif name in ("tilt_disk_target_in", "tilt_disk_target_out"):
                prm = _lc.disk_target_params(self.mesh, self.param_resolver, gp, name[17:])
                if prm is not None:  # tilt_disk_target_in.py:175-191
                    mods |= _ENERGY_BITS[name]
                    disk_params[name[17:]] = prm
            else:
                mods |= _ENERGY_BITS[name]
        target = 0.0
        stiffness = float(gp.get("volume_stiffness") or 0.0)
        body = mir.body
        if body is not None:
            t = body.target_volume if body.target_volume is not None else (body.options or {}).get("target_volume")
            if t is not None:
                target = float(t)
        if mods & L.MS_MOD_VOLUME_PENALTY:
            kv = body_penalty_params(self.mesh, gp, self.param_resolver)
            stiffness, target = kv
        if "volume" in self.constraint_module_names and vol_mode == "lagrange" and body is not None \
                and self._target_volume() is not None:
            mods |= L.MS_CON_VOLUME
        if body is not None and self._target_volume() is not None and vol_mode == "lagrange" \
                and not gp.get("volume_projection_during_minimization", True):
            mods |= L.MS_TRACK_VOLUME  # drift check of minimizer.py:1478-1513
        model = bending_model(gp)
        if (mods & L.MS_MOD_BENDING) and (mods & L.MS_MOD_BENDING_TILT):
            raise L.MembraneHipError("bending and bending_tilt together are outside the HIP hot path")
        if mods & L.MS_MOD_BENDING_TILT:
            model = "helfrich"  # bending_tilt.py:212-215
        any_bend = mods & (L.MS_MOD_BENDING | L.MS_MOD_BENDING_TILT)
        mode = bending_gradient_mode(gp) if any_bend else "analytic"
        if mode == "approx" and (mods & L.MS_MOD_BENDING_TILT) and np.any(_boundary(self.mesh)) \
                and self.energy_module_names.index("bending_tilt") != len(self.energy_module_names) - 1:
            raise L.MembraneHipError(
                "bending_gradient_mode=approx with modules listed AFTER bending_tilt on an open mesh is "
                "not on the fused device path (bending_tilt.py:297-299 zeroes boundary rows of what was "
                "accumulated so far); list bending_tilt last")
        if mode == "approx" and (mods & L.MS_MOD_BENDING):
            order = self.energy_module_names
            if order.index("bending") != len(order) - 1 and np.any(_boundary(self.mesh)):
                raise L.MembraneHipError(
                    "bending_gradient_mode=approx with energy modules listed AFTER bending on an "
                    "open mesh is not on the fused device path (bending.py:165-166 zeroes boundary "
                    "rows of what was accumulated so far); list bending last")
        if mods & L.MS_MOD_SURFACE:
            mir.upload_surface_tension()
        if any_bend:
            mir.upload_bending_params(gp, model)
        if (mods & _SINGLE_TILT_BITS) and (mods & _LEAFLET_BITS):
            raise L.MembraneHipError("single-field and leaflet tilt modules together are outside the HIP hot path")
        if mods & _LEAFLET_BITS:
            _lc.check_supported(gp)
            if gp.get("line_search_reduced_energy", False):
                raise L.MembraneHipError("line_search_reduced_energy is outside the HIP hot path")
            mir.upload_leaflets({lf: _lc.device_params(self.param_resolver, gp, lf) for lf in ("in", "out")})
            for lf, prm in disk_params.items():
                key = (mir._topo_key, prm["disk_rows"].tobytes(), tuple((k, v) for k, v in prm.items() if k != "disk_rows"))
                if mir._leaflet_keys.get("bend_disk_" + lf) != key:
                    dm.set_leaflet_disk_target(lf, **prm)
                    mir._leaflet_keys["bend_disk_" + lf] = key
            if mods & _LEAFLET_BT_BITS:
                _lc.check_bt_supported(gp)
                if mods & (L.MS_MOD_BENDING | L.MS_MOD_BENDING_TILT):
                    raise L.MembraneHipError("bending / bending_tilt together with bending_tilt_in/out is outside "
                                             "the HIP hot path")
                for lf, bit in (("in", L.MS_MOD_BENDING_TILT_IN), ("out", L.MS_MOD_BENDING_TILT_OUT)):
                    if mods & bit:
                        kappa, c0 = _lc.bending_params(self.mesh, gp, lf)
                        key = (mir._topo_key, kappa.tobytes(), c0.tobytes())
                        if mir._leaflet_keys.get("bend_" + lf) != key:
                            dm.set_leaflet_bending(lf, kappa, c0)
                            mir._leaflet_keys["bend_" + lf] = key
        if mods & (_SINGLE_TILT_BITS):
            if gp.get("line_search_reduced_energy", False):
                raise L.MembraneHipError("line_search_reduced_energy (inner tilt relaxation inside every "
                                         "line-search trial, minimizer.py:568-608) is outside the HIP hot path")
            if str(gp.get("tilt_transport_model", "ambient_v1") or "ambient_v1").strip().lower() != "ambient_v1":
                raise L.MembraneHipError("tilt_transport_model other than ambient_v1 is outside the HIP hot path")
            mir.upload_tilts(gp)
            mir.upload_tilt_fixed()
            dm.set_tilt_smoothness(float(gp.get("tilt_smoothness_rigidity", 0.0) or 0.0))
        key = (mods, model, mode, stiffness, target, id(dm))
        if key != self._configured_key:
            dm.set_params(modules=mods,
                          bending_model=L.MS_BEND_HELFRICH if model == "helfrich" else L.MS_BEND_WILLMORE,
                          bending_grad_mode=L.MS_GRAD_ANALYTIC if mode == "analytic" else L.MS_GRAD_APPROX,
                          volume_stiffness=stiffness, target_volume=target)
            self._configured_key = key
        return mir, dm

    def _target_volume(self):
        bodies = getattr(self.mesh, "bodies", None) or {}
        if not bodies:
            return None
        body = next(iter(bodies.values()))
        t = body.target_volume
        if t is None:
            t = (body.options or {}).get("target_volume")
        return None if t is None else float(t)

    # -- evaluation entry points ------------------------------------------------
    def compute_energy_and_gradient_array(self):
        """Total energy and dense gradient (minimizer.py:941-992)."""
        _mir, dm = self._device()
        e, g = dm.energy_and_gradient(want_grad=True)
        return float(e.sum()) + self._energy_offset, g

    def compute_energy_and_gradient(self):
        E, g = self.compute_energy_and_gradient_array()
        return E, GradientRows(self.mesh, g)

    def compute_energy(self) -> float:
        """minimizer.py:1051-1054."""
        _mir, dm = self._device()
        return float(dm.energy().sum()) + self._energy_offset

    def compute_energy_breakdown(self) -> Dict[str, float]:
        """Per-module energies (minimizer.py:1056-1065)."""
        _mir, dm = self._device()
        e = dm.energy()
        out = {}
        for name in self.energy_module_names:
            out[name] = self._const_energy.get(name, 0.0) if _ENERGY_SLOT[name] is None else float(e[_ENERGY_SLOT[name]])
        for table in (_TILT_SCALAR, _BEND_SCALAR):
            sharing = [n for n in out if n in table]
            if len(sharing) > 1:  # they share one entry of the energy vector: split via the scalars
                sc = dm.fetch_scalars()
                for n in sharing:
                    out[n] = float(sc[table[n]]) if dm.modules & _ENERGY_BITS[n] else 0.0
        return out

    # -- constraint enforcement (minimizer.py:1103-1188) ---------------------------
    def _enforce(self, dm, context: str, first_step_cached: bool = False) -> bool:
        """Volume projection on the device.  Returns True if positions moved.  ``first_step_cached``: the
        reference's Body still holds the volume gradient of its previous projection and a compute_volume at the
        current mesh version has made that cache look current (ms_project_volume_cached)."""
        if not self._has_enforceable_constraints:
            return False
        gp = self.global_params
        if context == "minimize" and not gp.get("volume_projection_during_minimization", True):
            return False
        target = self._target_volume()
        if target is None:
            return False
        max_iter = 12 if context in ("finalize", "mesh_operation") else 3
        iters, _v = dm.project_volume(target, tol=1e-12, max_iter=max_iter, first_step_cached=first_step_cached)
        moved = iters > 0
        if context in ("finalize", "mesh_operation") and dm.modules & _TILT_BITS:
            # minimizer.py:1186, :1224, :1506: every enforce outside the line search is followed by
            # mesh.project_tilts_to_tangent() -- the tilts must be tangent to the PROJECTED surface before the
            # next energy / gradient evaluation reads them
            dm.project_tilts_to_tangent()
            moved = True
        return moved

    # -- tilt relaxation (runtime/steppers/tilt_relaxation.py:237-300 parameter handling) --
    def _tilt_relax_params(self):
        """Resolved relaxation parameters or None when tilt_solve_mode leaves the tilts alone."""
        gp = self.global_params
        mode = str(gp.get("tilt_solve_mode", "fixed") or "").strip().lower()
        if mode in ("", "none", "off", "false", "fixed"):
            return None
        if mode not in ("nested", "coupled"):
            logger.warning("Unknown tilt_solve_mode=%r; treating as 'fixed'.", mode)
            return None
        step = float(gp.get("tilt_step_size", 0.0) or 0.0)
        if step <= 0.0:
            return None
        tol = float(gp.get("tilt_tol", 0.0) or 0.0)
        if mode == "nested":
            n_inner = int(gp.get("tilt_inner_steps", 0) or 0)
        else:
            n_inner = int(gp.get("tilt_coupled_steps", gp.get("tilt_inner_steps", 0)) or 0)
        if n_inner <= 0:
            return None
        solver = str(gp.get("tilt_solver", "cg") or "cg").strip().lower()
        if solver not in ("gd", "cg"):
            logger.warning("Unknown tilt_solver=%r; using gradient descent.", solver)
            solver = "gd"
        max_iters = int(gp.get("tilt_cg_max_iters", n_inner) or 0) if solver == "cg" else n_inner
        if max_iters <= 0:
            return None
        pre = str(gp.get("tilt_cg_preconditioner", "jacobi") or "jacobi").strip().lower()
        return {"solver": solver, "max_iters": max_iters, "step_size": step, "tol": max(tol, 0.0),
                "jacobi": pre == "jacobi"}

    def _relax_tilts(self, dm) -> bool:
        rp = self._tilt_relax_params()
        if rp is None:
            return False
        if dm.modules & _LEAFLET_BITS:
            dm.relax_leaflet_tilts(**rp)  # minimizer.py:1240-1305 (no energy guard)
        else:
            dm.relax_tilts(**rp)
        return True

    @property
    def _energy_offset(self) -> float:
        """Sum of the host-side constant modules (set by _device())."""
        return float(sum(getattr(self, "_const_energy", {}).values()))

    def _write_back_tilts(self, dm, mir):
        """Device tilt fields -> mesh (the reference leaves relaxed / projected tilts on the mesh)."""
        if dm.modules & _SINGLE_TILT_BITS:
            self.mesh.set_tilts_from_array(dm.get_tilts())
            mir.mark_device_tilts_current()
        if dm.modules & _LEAFLET_BITS:
            self.mesh.set_tilts_in_from_array(dm.get_leaflet_tilts("in"))
            self.mesh.set_tilts_out_from_array(dm.get_leaflet_tilts("out"))
            mir.mark_device_leaflets_current()

    def _fast_path_ok(self, callback) -> bool:
        """The whole loop can run inside the library (ms_minimize) when nothing on the Python side
        wants to see individual steps: no callback, quiet, and the stepper's device_step is the
        stock one (tests and tools wrap it to log steps)."""
        st = self.stepper
        return (callback is None and self.quiet and isinstance(st, BaseStepper)
                and "device_step" not in vars(st) and type(st).device_step is BaseStepper.device_step)

    def _minimize_in_library(self, mir, dm, n_steps, sync_mesh):
        gp = self.global_params
        st = self.stepper
        st._dm = dm
        edge_fraction = float(gp.get("shape_step_edge_fraction", 0.0) or 0.0)
        extra = st._extra()
        mp = L.ms_minimize_params()
        mp.stepper = L.ms_stepper_params(int(st.stepper_id), int(st._max_iter_for(self.mesh)), float(st.beta),
                                         float(st.c), float(st.gamma), float(st.alpha_max_factor),
                                         int(extra.get("restart_interval", 10)), edge_fraction,
                                         int(st.reuse_energy0), int(getattr(st, "enforce_volume", 0)),
                                         int(extra.get("precondition", 0)))
        step_mode = str(gp.get("step_size_mode", "adaptive") or "adaptive").lower()
        mp.step_size = float(self.step_size)
        mp.tol = float(self.tol)
        mp.fixed_step_mode = 1 if step_mode == "fixed" else 0
        mp.fixed_step = float(gp.get("step_size", self.step_size) or self.step_size)
        mp.max_zero_steps = int(self.max_zero_steps)
        mp.step_size_floor = float(self.step_size_floor)
        target = self._target_volume()
        mp.drift_check = 1 if (gp.get("volume_constraint_mode", "lagrange") == "lagrange"
                               and not gp.get("volume_projection_during_minimization", True)
                               and target is not None) else 0
        mp.target_volume = float(target) if target is not None else 0.0
        mp.volume_tolerance = float(gp.get("volume_tolerance", 1e-3))
        mp.project_on_drift = 1 if self._has_enforceable_constraints else 0
        rp = self._tilt_relax_params() if dm.modules & (_TILT_BITS) else None
        mp.relax_tilts = 0 if rp is None else 1
        if rp is not None:
            mp.relax = L.ms_tilt_relax_params(1 if rp["solver"] == "cg" else 0, rp["max_iters"], rp["step_size"],
                                              rp["tol"], 1 if rp["jacobi"] else 0)
        out, _log = dm.minimize(mp, n_steps)
        self.step_size = float(out.step_size)
        self.last_run = {"accepted": out.accepted, "trials": out.trials, "guard_rejects": out.guard_rejects,
                         "iterations": out.iterations}
        return out

    # -- the loop -------------------------------------------------------------------
    def minimize(self, n_steps: int = 1, callback: Optional[Callable] = None, *, sync_mesh: bool = True):
        """Run ``n_steps`` iterations (minimizer.py:1189-1535).

        ``sync_mesh=False`` leaves the new positions in HBM only (the mesh object is
        refreshed by the next call that uses ``sync_mesh=True`` or by
        ``sync_mesh_from_device()``); benchmarks use it to keep PCIe out of the loop."""
        mir, dm = self._device()
        gp = self.global_params
        # minimizer.py:1379: the stepper gets Minimizer._enforce_constraints as its constraint enforcer; with
        # volume_projection_during_minimization on (the programmatic default) that projects every trial onto the
        # target volume before its energy is taken (line_search.py:428-487) -- on the device: ms_stepper_params.enforce_volume
        self.stepper.enforce_volume = int(bool(
            self._has_enforceable_constraints and gp.get("volume_projection_during_minimization", True)
            and gp.get("volume_constraint_mode", "lagrange") == "lagrange" and self._target_volume() is not None
            and (dm.modules & L.MS_CON_VOLUME)))
        if n_steps <= 0:
            E, g = self.compute_energy_and_gradient_array()
            moved = self._enforce(dm, "minimize")
            if moved:
                write_back_positions(self.mesh, dm, mir)
            return {"energy": float(self.compute_energy()), "gradient": GradientRows(self.mesh, g),
                    "mesh": self.mesh, "step_success": True, "iterations": 0, "terminated_early": True}

        dirty = False
        if self._has_enforceable_constraints:
            dirty |= self._enforce(dm, "mesh_operation")

        if self._fast_path_ok(callback):
            out = self._minimize_in_library(mir, dm, n_steps, sync_mesh)
            moved = bool(out.moved) or dirty
            early = bool(out.converged or out.zero_step_exit)
            if out.converged:
                logger.info("Converged in %d iterations; |grad E|=%.3e", out.iterations - 1, out.grad_norm)
            moved_by_finalize = False
            if not out.zero_step_exit:  # minimizer.py:1324-1337 / :1516-1535 finalize the constraints
                moved_by_finalize = self._enforce(dm, "finalize", first_step_cached=bool(out.volume_cache_current))
                moved |= moved_by_finalize
            if out.converged:
                energy = float(out.energy_eval) + self._energy_offset
            elif out.energy_current_valid and not moved_by_finalize:
                # the step logic already holds the energies of the positions the loop ended at (reuse level 2: a pass
                # whose result is on the device is not repeated)
                energy = float(out.energy_current) + self._energy_offset
            else:
                energy = float(dm.energy().sum()) + self._energy_offset
            if moved or self._device_ahead:
                if sync_mesh:
                    write_back_positions(self.mesh, dm, mir)
                    if dm.modules & (_TILT_BITS):
                        self._write_back_tilts(dm, mir)
                    self._device_ahead = False
                else:
                    self._device_ahead = True
            return {"energy": energy, "gradient": GradientRows(self.mesh, dm.get_gradient() if sync_mesh
                                                               else dm.get_gradient),
                    "mesh": self.mesh, "step_success": bool(out.step_success) if not out.zero_step_exit else False,
                    "iterations": int(out.iterations), "terminated_early": early,
                    "steps_accepted": int(out.accepted), "line_search_trials": int(out.trials)}

        zero_step_counter = 0
        step_success = True
        # Body's cached volume is current (minimizer.py:1492 ran at this mesh version and nothing has bumped it since)
        vol_cache_current = False
        proj_flag = gp.get("volume_projection_during_minimization", True)
        vol_tol = float(gp.get("volume_tolerance", 1e-3))
        vol_mode = gp.get("volume_constraint_mode", "lagrange")
        target = self._target_volume()
        have_grad = False

        def finish(result):
            if dirty_box[0] or self._device_ahead:
                if sync_mesh:
                    write_back_positions(self.mesh, dm, mir)
                    if dm.modules & (_TILT_BITS):  # tilts changed on the device
                        self._write_back_tilts(dm, mir)
                    self._device_ahead = False
                else:
                    self._device_ahead = True
            return result

        dirty_box = [dirty]
        for i in range(n_steps):
            if callback:
                if dirty_box[0]:  # the callback sees the mesh as the reference's does: positions and tilt fields
                    write_back_positions(self.mesh, dm, mir)
                    if dm.modules & (_TILT_BITS):
                        self._write_back_tilts(dm, mir)
                    dirty_box[0] = False
                callback(self.mesh, i)
                mir, dm = self._device()
            if dm.modules & (_TILT_BITS):
                if self._relax_tilts(dm):  # minimizer.py:1237-1307, before the convergence check
                    dirty_box[0] = True
            step_mode = str(gp.get("step_size_mode", "adaptive") or "adaptive").lower()
            fixed_step = float(gp.get("step_size", self.step_size) or self.step_size)
            step_size_in = fixed_step if step_mode == "fixed" else self.step_size
            r = self.stepper.device_step(dm, self.mesh, step_size_in, tol=self.tol)
            have_grad = True
            if r.converged:  # minimizer.py:1324-1337
                logger.info("Converged in %d iterations; |grad E|=%.3e", i, r.grad_norm)
                dirty_box[0] |= self._enforce(dm, "finalize", first_step_cached=vol_cache_current)
                return finish({"energy": r.energy_eval + self._energy_offset,
                               "gradient": GradientRows(self.mesh, dm.get_gradient()),
                               "mesh": self.mesh, "step_success": True, "iterations": i + 1,
                               "terminated_early": True})
            step_success = r.success
            vol_cache_current = False  # minimizer.py:1415-1416: project_tilts_to_tangent + increment_version
            self.step_size = r.next_step
            if r.success:
                dirty_box[0] = True
            if not self.quiet:
                print(f"Step {i:4d}: Energy = {r.energy:.5f}, Step Size  = {step_size_in:.2e}")
            if step_mode == "fixed":
                self.step_size = fixed_step
            if not step_success:
                if self.step_size <= self.step_size_floor:
                    zero_step_counter += 1
                    if zero_step_counter >= self.max_zero_steps:
                        logger.info("Terminating early after %d consecutive zero-steps", zero_step_counter)
                        return finish({"energy": float(dm.energy().sum()) + self._energy_offset,
                                       "gradient": GradientRows(self.mesh, dm.get_gradient()),
                                       "mesh": self.mesh, "step_success": False, "iterations": i + 1,
                                       "terminated_early": True})
                else:
                    zero_step_counter = 0
                self.stepper.reset()
            else:
                zero_step_counter = 0
                # Lagrange volume drift check (minimizer.py:1478-1513)
                if vol_mode == "lagrange" and not proj_flag and target is not None:
                    vol_cache_current = True  # body.compute_volume at the current mesh version
                    denom = max(abs(target), 1.0)
                    if abs(r.volume - target) / denom > vol_tol:
                        dirty_box[0] |= self._enforce(dm, "mesh_operation", first_step_cached=True)
                        vol_cache_current = False  # enforce_constraints_after_mesh_ops: increment_version
                        self.stepper.reset()
        dirty_box[0] |= self._enforce(dm, "finalize", first_step_cached=vol_cache_current)
        final_energy = float(dm.energy().sum()) + self._energy_offset
        grad = GradientRows(self.mesh, dm.get_gradient() if sync_mesh else dm.get_gradient) if have_grad else {}
        return finish({"energy": final_energy, "gradient": grad, "mesh": self.mesh,
                       "step_success": step_success, "iterations": n_steps, "terminated_early": False})


def _boundary(mesh):
    from ..geometry.mesh import _boundary_mask_of

    return _boundary_mask_of(mesh, len(mesh.vertex_ids))
