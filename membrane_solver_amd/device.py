"""DeviceMesh: the HBM-resident mirror of what the hot path reads from the
reference's ``Mesh`` (positions_view, triangle_row_cache, fixed_mask,
boundary_vertex_ids, per-facet / per-vertex parameter arrays:
geometry/mesh.py:372-389, :597-624, :210-232, :304-319, :234-265).

Thin object wrapper over the C ABI (include/membrane_hip.h).  All arithmetic
runs in libmembrane_hip.so on the GPU; every error raises MembraneHipError.
"""

from __future__ import annotations

import ctypes
from dataclasses import dataclass

import numpy as np

from . import _lib as L


def _f64(a, shape=None, name="array"):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError(f"{name} must have shape {tuple(shape)}, got {a.shape}")
    return a


def _pd(a):
    return None if a is None else a.ctypes.data_as(L._D)


def _pu8(a):
    return None if a is None else a.ctypes.data_as(L._U8)


@dataclass(slots=True)
class StepResult:
    success: bool
    converged: bool
    trials: int
    guard_rejects: int
    next_step: float
    energy: float
    alpha: float
    energy_eval: float
    grad_norm: float
    g_dot_d: float
    volume: float


class DeviceMesh:
    """One mesh resident on one GPU (or one shard of it)."""

    def __init__(self, positions, tri_rows, *, fixed=None, boundary=None, body_facets=None,
                 device: int = 0, tile_vertices: int = 0, shard_rank: int = 0, shard_count: int = 1):
        lib = L.lib()
        pos = _f64(positions, name="positions")
        if pos.ndim != 2 or pos.shape[1] != 3:
            raise ValueError("positions must be (nv,3)")
        tri = np.ascontiguousarray(tri_rows, dtype=np.int32)
        if tri.size and (tri.ndim != 2 or tri.shape[1] != 3):
            raise ValueError("tri_rows must be (nf,3)")
        self.nv = int(pos.shape[0])
        self._tri_for_tests = tri
        self.nf = int(tri.shape[0]) if tri.size else 0
        fx = None if fixed is None else np.ascontiguousarray(fixed, dtype=np.uint8)
        bd = None if boundary is None else np.ascontiguousarray(boundary, dtype=np.uint8)
        bf = None if body_facets is None else np.ascontiguousarray(body_facets, dtype=np.uint8)
        for arr, n, nm in ((fx, self.nv, "fixed"), (bd, self.nv, "boundary"), (bf, self.nf, "body_facets")):
            if arr is not None and arr.shape != (n,):
                raise ValueError(f"{nm} must have shape ({n},)")
        self._h = ctypes.c_void_p()
        rc = lib.ms_create(ctypes.byref(self._h), int(device), self.nv, self.nf, _pd(pos),
                           tri.ctypes.data_as(L._I32) if tri.size else None, _pu8(fx), _pu8(bd),
                           _pu8(bf), int(tile_vertices), int(shard_rank), int(shard_count))
        L.check(rc, None, "ms_create")
        self.device = int(device)
        self.shard_rank, self.shard_count = int(shard_rank), int(shard_count)
        self._params = L.ms_params(L.MS_MOD_SURFACE, L.MS_BEND_HELFRICH, L.MS_GRAD_ANALYTIC, 1000.0, 0.0)

    # -- lifetime -------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            L.lib().ms_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        L.check(rc, self._h, what)

    # -- parameters -----------------------------------------------------------
    def set_stream(self, hip_stream: int | None):
        self._chk(L.lib().ms_set_stream(self._h, ctypes.c_void_p(hip_stream or 0)), "ms_set_stream")

    def set_surface_tension(self, gamma):
        g = _f64(gamma, (self.nf,), "gamma")
        self._chk(L.lib().ms_set_surface_tension(self._h, _pd(g)), "ms_set_surface_tension")

    def set_bending_params(self, kappa, c0):
        k = _f64(kappa, (self.nv,), "kappa")
        z = _f64(c0, (self.nv,), "c0")
        self._chk(L.lib().ms_set_bending_params(self._h, _pd(k), _pd(z)), "ms_set_bending_params")

    def set_params(self, *, modules: int, bending_model: int = L.MS_BEND_HELFRICH,
                   bending_grad_mode: int = L.MS_GRAD_ANALYTIC, volume_stiffness: float = 1000.0,
                   target_volume: float = 0.0):
        self._params = L.ms_params(int(modules), int(bending_model), int(bending_grad_mode),
                                   float(volume_stiffness), float(target_volume))
        self._chk(L.lib().ms_set_params(self._h, ctypes.byref(self._params)), "ms_set_params")

    @property
    def modules(self) -> int:
        return int(self._params.modules)

    # -- host <-> HBM ---------------------------------------------------------
    def set_positions(self, positions):
        p = _f64(positions, (self.nv, 3), "positions")
        self._chk(L.lib().ms_set_positions(self._h, _pd(p)), "ms_set_positions")

    def get_positions(self) -> np.ndarray:
        out = np.empty((self.nv, 3), dtype=np.float64)
        self._chk(L.lib().ms_get_positions(self._h, _pd(out)), "ms_get_positions")
        return out

    def get_gradient(self) -> np.ndarray:
        out = np.empty((self.nv, 3), dtype=np.float64)
        self._chk(L.lib().ms_get_gradient(self._h, _pd(out)), "ms_get_gradient")
        return out

    def get_vertex_buffer(self, buffer: int) -> np.ndarray:
        ncomp = 2 if buffer == L.MS_BUF_FA else 3
        out = np.empty((self.nv, ncomp), dtype=np.float64)
        self._chk(L.lib().ms_get_vertex_buffer(self._h, int(buffer), _pd(out)), "ms_get_vertex_buffer")
        return out

    # -- evaluation -----------------------------------------------------------
    def set_tilts(self, tilts, tilt_rigidity: float):
        t = _f64(tilts, (self.nv, 3), "tilts")
        self._chk(L.lib().ms_set_tilts(self._h, _pd(t), float(tilt_rigidity)), "ms_set_tilts")

    def get_tilts(self) -> np.ndarray:
        out = np.empty((self.nv, 3), dtype=np.float64)
        self._chk(L.lib().ms_get_tilts(self._h, _pd(out)), "ms_get_tilts")
        return out

    def get_tilt_gradient(self) -> np.ndarray:
        out = np.empty((self.nv, 3), dtype=np.float64)
        self._chk(L.lib().ms_get_tilt_gradient(self._h, _pd(out)), "ms_get_tilt_gradient")
        return out

    def set_tilt_fixed(self, tilt_fixed=None):
        """vertex.tilt_fixed flags of the reference (None clears them)."""
        if tilt_fixed is None:
            ptr = None
        else:
            arr = np.ascontiguousarray(np.asarray(tilt_fixed, dtype=bool).astype(np.uint8))
            if arr.shape != (self.nv,):
                raise ValueError("tilt_fixed must have shape (nv,)")
            ptr = arr.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))
        self._chk(L.lib().ms_set_tilt_fixed(self._h, ptr), "ms_set_tilt_fixed")

    def set_deterministic(self, on: bool = True):
        """Fixed-order (bitwise reproducible) per-vertex sums instead of LDS atomics (~20 % slower)."""
        self._chk(L.lib().ms_set_deterministic(self._h, int(bool(on))), "ms_set_deterministic")

    def set_tilt_smoothness(self, k_smooth: float):
        self._chk(L.lib().ms_set_tilt_smoothness(self._h, float(k_smooth)), "ms_set_tilt_smoothness")

    def tilt_energy_and_gradient(self, want_gradient: bool = True):
        """-> (energy of the tilt-reading modules, dE/dt (nv,3) or None) at the stored tilts."""
        e = ctypes.c_double(0.0)
        out = np.empty((self.nv, 3), dtype=np.float64) if want_gradient else None
        self._chk(L.lib().ms_tilt_energy_and_gradient(self._h, ctypes.byref(e), _pd(out) if want_gradient else None),
                  "ms_tilt_energy_and_gradient")
        return float(e.value), out

    def relax_tilts(self, *, solver: str = "cg", max_iters: int, step_size: float, tol: float = 0.0,
                    jacobi: bool = True):
        """TiltRelaxationManager.relax_tilts on the device -> (iterations, energy evaluations)."""
        rp = L.ms_tilt_relax_params(1 if solver == "cg" else 0, int(max_iters), float(step_size), float(tol),
                                    1 if jacobi else 0)
        it, ev = ctypes.c_int(0), ctypes.c_int(0)
        self._chk(L.lib().ms_relax_tilts(self._h, ctypes.byref(rp), ctypes.byref(it), ctypes.byref(ev)),
                  "ms_relax_tilts")
        return int(it.value), int(ev.value)

    # -- two-leaflet tilt fields (tilts_in / tilts_out) ----------------------------
    def set_leaflet_tilts(self, leaflet: str, tilts, *, tilt_fixed=None, tilt_modulus: float = 0.0,
                          mass_mode: str = "lumped", smoothness: float = 0.0,
                          precond_smoothness: float | None = None):
        """Mesh.tilts_in_view()/tilts_out_view() + the leaflet's module parameters."""
        lf = {"in": L.MS_LEAFLET_IN, "out": L.MS_LEAFLET_OUT}[leaflet]
        arr = _f64(tilts, (self.nv, 3), f"tilts_{leaflet}")
        if tilt_fixed is None:
            ptr = None
        else:
            fx = np.ascontiguousarray(np.asarray(tilt_fixed, dtype=bool).astype(np.uint8))
            if fx.shape != (self.nv,):
                raise ValueError("tilt_fixed must have shape (nv,)")
            ptr = fx.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))
        if mass_mode not in ("lumped", "consistent"):
            raise ValueError(f"tilt_mass_mode_{leaflet} must be 'lumped' or 'consistent'.")
        lp = L.ms_leaflet_params(float(tilt_modulus), 1 if mass_mode == "consistent" else 0, float(smoothness),
                                 float(smoothness if precond_smoothness is None else precond_smoothness))
        self._chk(L.lib().ms_set_leaflet_tilts(self._h, lf, _pd(arr), ptr, ctypes.byref(lp)), "ms_set_leaflet_tilts")

    def set_leaflet_bending(self, leaflet: str, kappa, c0):
        """Per-vertex (kappa, c0) of bending_tilt_in / bending_tilt_out."""
        lf = {"in": L.MS_LEAFLET_IN, "out": L.MS_LEAFLET_OUT}[leaflet]
        k = _f64(np.broadcast_to(np.asarray(kappa, dtype=np.float64), (self.nv,)), (self.nv,), "kappa")
        c = _f64(np.broadcast_to(np.asarray(c0, dtype=np.float64), (self.nv,)), (self.nv,), "c0")
        self._chk(L.lib().ms_set_leaflet_bending(self._h, lf, _pd(k), _pd(c)), "ms_set_leaflet_bending")

    def set_leaflet_disk_target(self, leaflet: str, disk_rows, *, strength: float, theta_b: float, lam: float,
                                center=(0.0, 0.0, 0.0), normal=(0.0, 0.0, 1.0), radius: float | None = None):
        """tilt_disk_target_in/out: tagged rows (bool mask or row indices) and the profile parameters."""
        lf = {"in": L.MS_LEAFLET_IN, "out": L.MS_LEAFLET_OUT}[leaflet]
        rows = np.asarray(disk_rows)
        if rows.dtype == bool:
            mask = rows.astype(np.uint8)
        else:
            mask = np.zeros(self.nv, dtype=np.uint8)
            mask[rows.astype(np.int64)] = 1
        if mask.shape != (self.nv,):
            raise ValueError("disk_rows must be a (nv,) mask or a list of rows")
        mask = np.ascontiguousarray(mask)
        p = L.ms_disk_target_params(float(strength), float(theta_b), float(lam), (ctypes.c_double * 3)(*map(float, center)),
                                    (ctypes.c_double * 3)(*map(float, normal)), float(radius or 0.0))
        self._chk(L.lib().ms_set_leaflet_disk_target(self._h, lf, mask.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
                                                     ctypes.byref(p)), "ms_set_leaflet_disk_target")

    def get_leaflet_tilts(self, leaflet: str) -> np.ndarray:
        out = np.empty((self.nv, 3), dtype=np.float64)
        lf = {"in": L.MS_LEAFLET_IN, "out": L.MS_LEAFLET_OUT}[leaflet]
        self._chk(L.lib().ms_get_leaflet_tilts(self._h, lf, _pd(out)), "ms_get_leaflet_tilts")
        return out

    def leaflet_tilt_energy_and_gradient(self, want_gradient: bool = True, module_form: bool = False):
        """-> (tilt-dependent energy, dE/dt_in, dE/dt_out) at frozen positions, magnitude modules
        in their vertex-area form (evaluation_manager.py:630-742 with tilt_vertex_areas); with
        ``module_form`` in their own mass mode, as the plugin API evaluates them
        (tilt_leaflet.py:101-150: lumped, or consistent k A/12 (2 t_k + t_a + t_b))."""
        e = ctypes.c_double(0.0)
        gi = np.empty((self.nv, 3), dtype=np.float64) if want_gradient else None
        go = np.empty((self.nv, 3), dtype=np.float64) if want_gradient else None
        self._chk(L.lib().ms_leaflet_tilt_energy_and_gradient_ex(
            self._h, 1 if module_form else 0, ctypes.byref(e), _pd(gi) if want_gradient else None,
            _pd(go) if want_gradient else None), "ms_leaflet_tilt_energy_and_gradient_ex")
        return float(e.value), gi, go

    def relax_leaflet_tilts(self, *, solver: str = "cg", max_iters: int, step_size: float, tol: float = 0.0,
                            jacobi: bool = True):
        """TiltRelaxationManager.relax_leaflet_tilts on the device -> (iterations, energy evaluations)."""
        rp = L.ms_tilt_relax_params(1 if solver == "cg" else 0, int(max_iters), float(step_size), float(tol),
                                    1 if jacobi else 0)
        it, ev = ctypes.c_int(0), ctypes.c_int(0)
        self._chk(L.lib().ms_relax_leaflet_tilts(self._h, ctypes.byref(rp), ctypes.byref(it), ctypes.byref(ev)),
                  "ms_relax_leaflet_tilts")
        return int(it.value), int(ev.value)

    def angle_defects(self) -> np.ndarray:
        """Per-vertex angle defects (integrated Gaussian curvature) at the current positions."""
        out = np.empty(self.nv, dtype=np.float64)
        self._chk(L.lib().ms_angle_defects(self._h, _pd(out)), "ms_angle_defects")
        return out

    def curvature_fields(self) -> dict:
        """geometry/curvature.compute_curvature_fields at the context's positions (ms_curvature_fields) + the raw
        per-vertex angle sums."""
        a, b, c, d = (np.empty((self.nv, 3), dtype=np.float64) for _ in range(4))
        self._chk(L.lib().ms_curvature_fields(self._h, _pd(a), _pd(b), _pd(c), _pd(d)), "ms_curvature_fields")
        return {"mean_curvature_normal": a, "mean_curvature": b[:, 0].copy(), "mixed_area": b[:, 1].copy(),
                "angle_sum": b[:, 2].copy(), "angle_defect": c[:, 0].copy(), "gaussian_curvature": c[:, 1].copy(),
                "principal_curvatures": d[:, :2].copy()}

    def project_tilts_to_tangent(self):
        self._chk(L.lib().ms_project_tilts_to_tangent(self._h), "ms_project_tilts_to_tangent")

    def energy_and_gradient(self, want_grad: bool = True, raw: bool = False):
        """-> (energies[surface, bending, volume_penalty, tilt], grad (nv,3) | None).  ``raw``: the module loop's
        plain sum as an energy-module plugin accumulates it -- fixed rows not zeroed, no constraint projection."""
        e = np.zeros(4)
        g = np.empty((self.nv, 3), dtype=np.float64) if want_grad else None
        if raw:
            self._chk(L.lib().ms_energy_and_raw_gradient(self._h, _pd(e), _pd(g)), "ms_energy_and_raw_gradient")
        else:
            self._chk(L.lib().ms_energy_and_gradient(self._h, _pd(e), _pd(g)), "ms_energy_and_gradient")
        return e, g

    def energy(self) -> np.ndarray:
        e = np.zeros(4)
        self._chk(L.lib().ms_energy(self._h, _pd(e)), "ms_energy")
        return e

    # -- stepping -------------------------------------------------------------
    def step(self, *, stepper: int, step_size: float, tol: float = 1e-6, max_iter: int = 10,
             beta: float = 0.7, c: float = 1e-4, gamma: float = 1.5, alpha_max_factor: float = 10.0,
             restart_interval: int = 10, edge_fraction: float = 0.0,
             reuse_energy0: int = 0, enforce_volume: int = 0, precondition: int = 0) -> StepResult:
        # the parameter block and the result struct are reused between calls: at ~140 us per
        # step every microsecond of ctypes marshalling shows
        key = (stepper, max_iter, beta, c, gamma, alpha_max_factor, restart_interval, edge_fraction,
               reuse_energy0, enforce_volume, precondition)
        cache = self.__dict__.get("_step_cache")
        if cache is None or cache[0] != key:
            sp = L.ms_stepper_params(int(stepper), int(max_iter), float(beta), float(c), float(gamma),
                                     float(alpha_max_factor), int(restart_interval), float(edge_fraction),
                                     int(reuse_energy0), int(enforce_volume), int(precondition))
            r = L.ms_step_result()
            cache = (key, sp, r, ctypes.byref(sp), ctypes.byref(r), L.lib().ms_step)
            self._step_cache = cache
        _k, sp, r, sp_ref, r_ref, fn = cache
        rc = fn(self._h, sp_ref, step_size, tol, r_ref)
        if rc != 0:
            self._chk(rc, "ms_step")
        return StepResult(r.success != 0, r.converged != 0, r.trials, r.guard_rejects, r.next_step, r.energy,
                          r.alpha, r.energy_eval, r.grad_norm, r.g_dot_d, r.volume)

    def minimize(self, params: "L.ms_minimize_params", n_steps: int, want_log: bool = False):
        """ms_minimize: n_steps iterations of the minimizer loop inside the library.
        -> (ms_minimize_result, step_log (iterations,8) | None)"""
        out = L.ms_minimize_result()
        log = np.zeros((int(n_steps), 8)) if want_log else None
        self._chk(L.lib().ms_minimize(self._h, ctypes.byref(params), int(n_steps), ctypes.byref(out),
                                      _pd(log) if want_log else None), "ms_minimize")
        return out, (log[: out.iterations] if want_log else None)

    def reset_stepper(self):
        self._chk(L.lib().ms_reset_stepper(self._h), "ms_reset_stepper")

    def project_volume(self, target: float, tol: float = 1e-12, max_iter: int = 3, first_step_cached: bool = False):
        """volume.enforce_constraint's projection loop; ``first_step_cached``: Body's cached-gradient quirk
        (include/membrane_hip.h, ms_project_volume_cached)."""
        it = ctypes.c_int(0)
        v = ctypes.c_double(0.0)
        self._chk(L.lib().ms_project_volume_cached(self._h, float(target), float(tol), int(max_iter),
                                                   int(bool(first_step_cached)), ctypes.byref(it), ctypes.byref(v)),
                  "ms_project_volume_cached")
        return int(it.value), float(v.value)

    # -- phase API (multi-GPU drivers) ----------------------------------------
    def phase_energy(self, *, use_direction=False, alpha=0.0, write_trial=False, guard=False,
                     write_bending_factors=False):
        self._chk(L.lib().ms_phase_energy(self._h, int(use_direction), float(alpha), int(write_trial),
                                          int(guard), int(write_bending_factors)), "ms_phase_energy")

    def phase_gradient(self):
        self._chk(L.lib().ms_phase_gradient(self._h), "ms_phase_gradient")

    def phase_direction(self, stepper: int, use_history: bool):
        self._chk(L.lib().ms_phase_direction(self._h, int(stepper), int(use_history)), "ms_phase_direction")

    def phase_accept(self, keep_history: bool):
        self._chk(L.lib().ms_phase_accept(self._h, int(keep_history)), "ms_phase_accept")

    def phase_commit_trial(self, alpha: float, keep_history: bool):
        self._chk(L.lib().ms_phase_commit_trial(self._h, float(alpha), int(keep_history)),
                  "ms_phase_commit_trial")

    def phase_gradient_direction(self, stepper: int, use_history: bool):
        self._chk(L.lib().ms_phase_gradient_direction(self._h, int(stepper), int(use_history)),
                  "ms_phase_gradient_direction")

    def phase_set_factors_valid(self, valid: bool):
        self._chk(L.lib().ms_phase_set_factors_valid(self._h, int(valid)), "ms_phase_set_factors_valid")

    # -- library-side sharded driver ----------------------------------------------
    @staticmethod
    def shard_unique_id() -> bytes:
        buf = ctypes.create_string_buffer(128)
        L.check(L.lib().ms_shard_unique_id(buf), None, "ms_shard_unique_id")
        return buf.raw

    def shard_comm_init(self, unique_id: bytes):
        if len(unique_id) != 128:
            raise ValueError("ncclUniqueId is 128 bytes")
        self._chk(L.lib().ms_shard_comm_init(self._h, ctypes.create_string_buffer(unique_id, 128)),
                  "ms_shard_comm_init")

    def shard_set_allgather(self, fn):
        """fn(send_ptr, recv_ptr, bytes_per_rank) -> None: in-process stand-in for ncclAllGather."""
        def thunk(_user, send, recv, nbytes):
            try:
                fn(int(send), int(recv), int(nbytes))
                return 0
            except Exception:  # pragma: no cover - surfaced as MS_ERR_STATE
                import traceback

                traceback.print_exc()
                return 1

        self._allgather_cb = L.ALLGATHER_FN(thunk)  # keep the trampoline alive
        self._chk(L.lib().ms_shard_set_allgather(self._h, self._allgather_cb, None), "ms_shard_set_allgather")

    # peer-to-peer exchange of the library driver (include/membrane_hip.h, ms_shard_peer_*)
    def shard_peer_local(self):
        """-> (slab pointer, flag-word pointer) of this rank, as integers (contexts of one process exchange these)."""
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        self._chk(L.lib().ms_shard_peer_local(self._h, ctypes.byref(a), ctypes.byref(b)), "ms_shard_peer_local")
        return int(a.value or 0), int(b.value or 0)

    def shard_peer_set_pointers(self, slabs, flags):
        n = len(slabs)
        a = (ctypes.c_void_p * n)(*[ctypes.c_void_p(int(x)) for x in slabs])
        b = (ctypes.c_void_p * n)(*[ctypes.c_void_p(int(x)) for x in flags])
        self._chk(L.lib().ms_shard_peer_set_pointers(self._h, a, b), "ms_shard_peer_set_pointers")

    def shard_peer_set_barrier(self, fn):
        """fn() -> None: returns once every rank of this process has raised its flags (host-side wait)."""
        def thunk(_user):
            try:
                fn()
                return 0
            except Exception:  # pragma: no cover - surfaced as MS_ERR_STATE
                import traceback

                traceback.print_exc()
                return 1

        self._peer_barrier_cb = L.BARRIER_FN(thunk)  # keep the trampoline alive
        self._chk(L.lib().ms_shard_peer_set_barrier(self._h, ctypes.cast(self._peer_barrier_cb, ctypes.c_void_p), None),
                  "ms_shard_peer_set_barrier")

    def shard_peer_export(self) -> bytes:
        buf = ctypes.create_string_buffer(128)
        self._chk(L.lib().ms_shard_peer_export(self._h, buf), "ms_shard_peer_export")
        return bytes(buf.raw)

    def shard_peer_open(self, handles_all: bytes):
        buf = ctypes.create_string_buffer(bytes(handles_all), len(handles_all))
        self._chk(L.lib().ms_shard_peer_open(self._h, buf), "ms_shard_peer_open")

    def shard_step(self, *, stepper: int, step_size: float, tol: float = 1e-6, max_iter: int = 10,
                   beta: float = 0.7, c: float = 1e-4, gamma: float = 1.5, alpha_max_factor: float = 10.0,
                   restart_interval: int = 10, edge_fraction: float = 0.0, reuse_energy0: int = 2) -> StepResult:
        sp = L.ms_stepper_params(int(stepper), int(max_iter), float(beta), float(c), float(gamma),
                                 float(alpha_max_factor), int(restart_interval), float(edge_fraction),
                                 int(reuse_energy0))
        r = L.ms_step_result()
        self._chk(L.lib().ms_shard_step(self._h, ctypes.byref(sp), float(step_size), float(tol), ctypes.byref(r)),
                  "ms_shard_step")
        return StepResult(r.success != 0, r.converged != 0, r.trials, r.guard_rejects, r.next_step, r.energy,
                          r.alpha, r.energy_eval, r.grad_norm, r.g_dot_d, r.volume)

    def shard_comm_ranks(self) -> int:
        """Ranks of the context's RCCL communicator (ncclCommCount); 0 without one."""
        return int(L.lib().ms_shard_comm_ranks(self._h))

    def shard_exchange_count(self) -> int:
        return int(L.lib().ms_shard_exchange_count(self._h))

    def shard_chain_stats(self):
        """Device-side trial decisions of ms_shard_step and the passes chained behind them (ms_shard_chain_stats)."""
        v = np.zeros(8, dtype=np.int64)
        self._chk(L.lib().ms_shard_chain_stats(self._h, v.ctypes.data_as(L._I64)), "ms_shard_chain_stats")
        return {"queued": int(v[0]), "ran": int(v[1]), "adopted": int(v[2]), "dropped": int(v[3]),
                "ahead_queued": int(v[4]), "ahead_adopted": int(v[5]), "ahead_dropped": int(v[6])}

    def peer_memory_kind(self) -> str:
        """Memory kind of the peer exchange's slabs / flag words (ms_shard_peer_memory_kind)."""
        k = int(L.lib().ms_shard_peer_memory_kind(self._h))
        return {0: "uncached device memory", 1: "fine-grained device memory",
                2: "plain hipMalloc (coarse-grained)"}.get(k, "not allocated")

    # -- shard boundary exchange ------------------------------------------------
    def boundary_info(self):
        v = np.zeros(4, dtype=np.int64)
        self._chk(L.lib().ms_boundary_info(self._h, v.ctypes.data_as(L._I64)), "ms_boundary_info")
        return {"max_rows": int(v[0]), "my_rows": int(v[1]), "halo_rows": int(v[2]), "world": int(v[3])}

    @staticmethod
    def _ids(buffers):
        return (ctypes.c_int * max(1, len(buffers)))(*[int(b) for b in buffers])

    def exchange_bytes(self, buffers) -> int:
        return int(L.lib().ms_exchange_bytes(self._h, len(buffers), self._ids(buffers)))

    def pack_boundary(self, buffers, send_ptr: int, send_bytes: int):
        self._chk(L.lib().ms_pack_boundary(self._h, len(buffers), self._ids(buffers),
                                           ctypes.c_void_p(send_ptr), int(send_bytes)), "ms_pack_boundary")

    def unpack_boundary(self, buffers, recv_ptr: int, stride_bytes: int, world: int) -> np.ndarray:
        out = np.zeros((world, L.MS_NSCAL))
        self._chk(L.lib().ms_unpack_boundary(self._h, len(buffers), self._ids(buffers),
                                             ctypes.c_void_p(recv_ptr), int(stride_bytes), _pd(out)),
                  "ms_unpack_boundary")
        return out

    def state_bytes(self) -> int:
        return int(L.lib().ms_state_bytes(self._h))

    def rebind_state(self, device_ptr: int, nbytes: int):
        self._chk(L.lib().ms_rebind_state(self._h, ctypes.c_void_p(device_ptr), int(nbytes)),
                  "ms_rebind_state")

    def fetch_scalars(self) -> np.ndarray:
        out = np.zeros(L.MS_NSCAL)
        self._chk(L.lib().ms_fetch_scalars(self._h, _pd(out)), "ms_fetch_scalars")
        return out

    def store_scalars(self, values):
        v = _f64(values, (L.MS_NSCAL,), "scalars")
        self._chk(L.lib().ms_store_scalars(self._h, _pd(v)), "ms_store_scalars")

    def device_buffer(self, buffer: int):
        p = ctypes.c_void_p()
        n = ctypes.c_size_t(0)
        self._chk(L.lib().ms_device_buffer(self._h, int(buffer), ctypes.byref(p), ctypes.byref(n)),
                  "ms_device_buffer")
        return int(p.value or 0), int(n.value)

    def profile_enable(self, on: bool = True):
        self._chk(L.lib().ms_profile_enable(self._h, int(on)), "ms_profile_enable")

    def profile_read(self):
        """-> {kind: (total_ms, launches)} per kernel kind (include/membrane_hip.h, ms_profile_read)."""
        ms = np.zeros(13)
        n = np.zeros(13, dtype=np.int64)
        self._chk(L.lib().ms_profile_read(self._h, _pd(ms), n.ctypes.data_as(L._I64)), "ms_profile_read")
        names = ("energy", "gradient", "direction", "reduce", "tilt", "bending_tilt", "tilt_vec", "energy_pair", "energy_triple",
                 "gradient_lean", "energy_multi", "tilt_smoothness", "tilt_search")
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(names)}

    def queue_stats(self):
        """Line-search queue statistics (include/membrane_hip.h, ms_queue_stats)."""
        v = np.zeros(8, dtype=np.int64)
        self._chk(L.lib().ms_queue_stats(self._h, v.ctypes.data_as(L._I64)), "ms_queue_stats")
        return {"rounds": int(v[0]), "multi_launches": int(v[1]), "wasted_evaluations": int(v[2]),
                "side_accepts": int(v[3]), "mismatches": int(v[4]), "ahead": int(v[5]), "adopted": int(v[6]),
                "dropped": int(v[7])}

    def exec_stats(self):
        """One-workgroup interpreter of one-tile meshes (include/membrane_hip.h, ms_exec_stats)."""
        v = np.zeros(4, dtype=np.int64)
        self._chk(L.lib().ms_exec_stats(self._h, v.ctypes.data_as(L._I64)), "ms_exec_stats")
        return {"active": bool(v[0]), "packs": int(v[1]), "launches_recorded": int(v[2]), "wanted": bool(v[3] & 1),
                "relax_programs": int((v[3] >> 8) & 0xffffff), "relax_fused": int(v[3] >> 32)}

    def resident_stats(self):
        """The resident step kernel (include/membrane_hip.h, ms_resident_stats)."""
        v = np.zeros(4, dtype=np.int64)
        self._chk(L.lib().ms_resident_stats(self._h, v.ctypes.data_as(L._I64)), "ms_resident_stats")
        return {"co_resident": int(v[0]), "launches": int(v[1]), "steps": int(v[2]), "declined": int(v[3])}

    def tsearch_stats(self):
        """Search passes of the tilt relaxations (include/membrane_hip.h, ms_tsearch_stats)."""
        v = np.zeros(2, dtype=np.int64)
        self._chk(L.lib().ms_tsearch_stats(self._h, v.ctypes.data_as(L._I64)), "ms_tsearch_stats")
        return {"passes": int(v[0]), "step_sizes": int(v[1])}

    EXEC_KINDS = {1: "energy", 2: "gradient", 3: "tilt", 4: "bt", 5: "tsmooth", 6: "tvec", 7: "disk_target", 8: "reduce",
                  9: "direction", 10: "row_dot", 11: "axpy_masked", 12: "memset", 13: "relax", 14: "relax_fused", 15: "tsearch"}

    def exec_trace(self, on: bool = True):
        """Per-record durations of the one-workgroup interpreter since the last call (ms_exec_trace):
        -> list of {kind, mode, inst, count, total_us, avg_us}."""
        rows = np.zeros((256, 4), dtype=np.float64)
        n = ctypes.c_int(0)
        self._chk(L.lib().ms_exec_trace(self._h, 1 if on else 0, _pd(rows), 256, ctypes.byref(n)), "ms_exec_trace")
        out = []
        for k in range(n.value):
            kind, mi, cnt, tot = int(rows[k, 0]), int(rows[k, 1]), int(rows[k, 2]), float(rows[k, 3])
            out.append({"kind": self.EXEC_KINDS.get(kind, str(kind)), "mode": mi & 0xffff, "inst": mi >> 16, "count": cnt,
                        "total_us": tot, "avg_us": tot / max(cnt, 1)})
        return out

    def shard_info(self):
        v = [ctypes.c_int64(0) for _ in range(4)]
        self._chk(L.lib().ms_shard_info(self._h, *[ctypes.byref(x) for x in v]), "ms_shard_info")
        return {"nvp": v[0].value, "row0": v[1].value, "row1": v[2].value, "rows_per_shard": v[3].value}

    def tile_stats(self):
        v = [ctypes.c_int64(0) for _ in range(5)]
        self._chk(L.lib().ms_tile_stats(self._h, *[ctypes.byref(x) for x in v]), "ms_tile_stats")
        return {"n_tiles": v[0].value, "facet_instances": v[1].value, "max_halo": v[2].value,
                "lds_bytes_energy": v[3].value, "lds_bytes_gradient": v[4].value}
