"""ctypes binding of libmembrane_hip.so (include/membrane_hip.h).

The product path has NO CPU fallback: if the shared library is missing, or a
call returns an error, a ``MembraneHipError`` is raised.  ``build()`` compiles
the library in-tree with hipcc for gfx950.
"""

from __future__ import annotations

import ctypes
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MEMBRANE_HIP_LIB") or os.path.join(_PKG, "libmembrane_hip.so")  # override: A/B builds
_CSRC = os.path.join(_PKG, "csrc")

MS_OK = 0
MS_ERR = {-1: "MS_ERR_INVALID", -2: "MS_ERR_HIP", -3: "MS_ERR_TILE_CAPACITY",
          -4: "MS_ERR_STATE", -5: "MS_ERR_NOMEM"}

MS_MOD_SURFACE = 1
MS_MOD_BENDING = 2
MS_MOD_VOLUME_PENALTY = 4
MS_CON_VOLUME = 8
MS_TRACK_VOLUME = 16
MS_MOD_TILT = 32
MS_MOD_BENDING_TILT = 64
MS_MOD_TILT_SMOOTH = 128
MS_MOD_TILT_IN, MS_MOD_TILT_OUT, MS_MOD_TILT_SMOOTH_IN, MS_MOD_TILT_SMOOTH_OUT = 256, 512, 1024, 2048
MS_MOD_BENDING_TILT_IN, MS_MOD_BENDING_TILT_OUT = 4096, 8192
MS_MOD_TILT_DISK_TARGET_IN, MS_MOD_TILT_DISK_TARGET_OUT = 16384, 32768
MS_LEAFLET_IN, MS_LEAFLET_OUT = 0, 1
MS_BEND_HELFRICH, MS_BEND_WILLMORE = 0, 1
MS_GRAD_ANALYTIC, MS_GRAD_APPROX = 0, 1
MS_STEPPER_GD, MS_STEPPER_CG = 0, 1

(MS_BUF_X, MS_BUF_XT, MS_BUF_G, MS_BUF_GC, MS_BUF_D, MS_BUF_PG, MS_BUF_PD, MS_BUF_FK,
 MS_BUF_FA, MS_BUF_SCAL) = range(10)
(MS_S_ESURF, MS_S_VOL, MS_S_EBEND, MS_S_MINEDGE2, MS_S_GUARD, MS_S_GGC, MS_S_GCGC,
 MS_S_GNORM2, MS_S_GDOTD, MS_S_MAXD2, MS_S_ETILT, MS_S_EBT, MS_S_TGNORM2, MS_S_TRZ, MS_S_MAXG2,
 MS_S_ETS, MS_S_ETILT_IN, MS_S_ETILT_OUT, MS_S_ETS_IN, MS_S_ETS_OUT, MS_S_TGNORM2_IN, MS_S_TGNORM2_OUT,
 MS_S_TRZ_IN, MS_S_TRZ_OUT, MS_S_EBT_IN, MS_S_EBT_OUT, MS_S_EDT_IN, MS_S_EDT_OUT, MS_S_DTR_IN,
 MS_S_DTR_OUT) = range(30)
MS_NSCAL = 30


class MembraneHipError(RuntimeError):
    """Raised for every failure of the HIP path (there is no CPU fallback)."""


class ms_params(ctypes.Structure):
    _fields_ = [("modules", ctypes.c_uint32), ("bending_model", ctypes.c_int),
                ("bending_grad_mode", ctypes.c_int), ("volume_stiffness", ctypes.c_double),
                ("target_volume", ctypes.c_double)]


BARRIER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p)


class ms_stepper_params(ctypes.Structure):
    _fields_ = [("stepper", ctypes.c_int), ("max_iter", ctypes.c_int), ("beta", ctypes.c_double),
                ("c", ctypes.c_double), ("gamma", ctypes.c_double),
                ("alpha_max_factor", ctypes.c_double), ("restart_interval", ctypes.c_int),
                ("edge_fraction", ctypes.c_double), ("reuse_energy0", ctypes.c_int), ("enforce_volume", ctypes.c_int),
                ("precondition", ctypes.c_int)]


class ms_tilt_relax_params(ctypes.Structure):
    _fields_ = [("solver", ctypes.c_int), ("max_iters", ctypes.c_int), ("step_size", ctypes.c_double),
                ("tol", ctypes.c_double), ("jacobi", ctypes.c_int)]


class ms_leaflet_params(ctypes.Structure):
    _fields_ = [("tilt_modulus", ctypes.c_double), ("tilt_mass_consistent", ctypes.c_int),
                ("smoothness", ctypes.c_double), ("precond_smoothness", ctypes.c_double)]


class ms_disk_target_params(ctypes.Structure):
    _fields_ = [("strength", ctypes.c_double), ("theta_b", ctypes.c_double), ("lam", ctypes.c_double),
                ("center", ctypes.c_double * 3), ("normal", ctypes.c_double * 3), ("radius", ctypes.c_double)]


class ms_minimize_params(ctypes.Structure):
    _fields_ = [("stepper", ms_stepper_params), ("step_size", ctypes.c_double), ("tol", ctypes.c_double),
                ("fixed_step_mode", ctypes.c_int), ("fixed_step", ctypes.c_double),
                ("max_zero_steps", ctypes.c_int), ("step_size_floor", ctypes.c_double),
                ("drift_check", ctypes.c_int), ("target_volume", ctypes.c_double),
                ("volume_tolerance", ctypes.c_double), ("project_on_drift", ctypes.c_int),
                ("relax_tilts", ctypes.c_int), ("relax", ms_tilt_relax_params)]


class ms_minimize_result(ctypes.Structure):
    _fields_ = [("iterations", ctypes.c_int), ("converged", ctypes.c_int), ("zero_step_exit", ctypes.c_int),
                ("step_success", ctypes.c_int), ("accepted", ctypes.c_int), ("trials", ctypes.c_int),
                ("guard_rejects", ctypes.c_int), ("moved", ctypes.c_int), ("step_size", ctypes.c_double),
                ("energy_eval", ctypes.c_double), ("grad_norm", ctypes.c_double),
                ("volume_cache_current", ctypes.c_int), ("energy_current_valid", ctypes.c_int),
                ("energy_current", ctypes.c_double)]


class ms_step_result(ctypes.Structure):
    _fields_ = [("success", ctypes.c_int), ("converged", ctypes.c_int), ("trials", ctypes.c_int),
                ("guard_rejects", ctypes.c_int), ("next_step", ctypes.c_double),
                ("energy", ctypes.c_double), ("alpha", ctypes.c_double),
                ("energy_eval", ctypes.c_double), ("grad_norm", ctypes.c_double),
                ("g_dot_d", ctypes.c_double), ("volume", ctypes.c_double)]


_P = ctypes.c_void_p
_D = ctypes.POINTER(ctypes.c_double)
_I32 = ctypes.POINTER(ctypes.c_int32)
_U8 = ctypes.POINTER(ctypes.c_uint8)
_I64 = ctypes.POINTER(ctypes.c_int64)

# name -> (restype, argtypes): every symbol include/membrane_hip.h declares.
# ms_allgather_fn(user, send_dev, recv_dev, bytes_per_rank) -> int
ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)

SIGNATURES = {
    "ms_version": (ctypes.c_char_p, []),
    "ms_device_count": (ctypes.c_int, []),
    "ms_last_error": (ctypes.c_char_p, [_P]),
    "ms_create": (ctypes.c_int, [ctypes.POINTER(_P), ctypes.c_int, ctypes.c_int, ctypes.c_int, _D,
                                 _I32, _U8, _U8, _U8, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "ms_destroy": (None, [_P]),
    "ms_set_stream": (ctypes.c_int, [_P, _P]),
    "ms_set_surface_tension": (ctypes.c_int, [_P, _D]),
    "ms_set_bending_params": (ctypes.c_int, [_P, _D, _D]),
    "ms_set_params": (ctypes.c_int, [_P, ctypes.POINTER(ms_params)]),
    "ms_set_tilts": (ctypes.c_int, [_P, _D, ctypes.c_double]),
    "ms_get_tilts": (ctypes.c_int, [_P, _D]),
    "ms_get_tilt_gradient": (ctypes.c_int, [_P, _D]),
    "ms_project_tilts_to_tangent": (ctypes.c_int, [_P]),
    "ms_angle_defects": (ctypes.c_int, [_P, _D]),
    "ms_curvature_fields": (ctypes.c_int, [_P, _D, _D, _D, _D]),
    "ms_set_tilt_fixed": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_uint8)]),
    "ms_set_tilt_smoothness": (ctypes.c_int, [_P, ctypes.c_double]),
    "ms_set_deterministic": (ctypes.c_int, [_P, ctypes.c_int]),
    "ms_tilt_energy_and_gradient": (ctypes.c_int, [_P, _D, _D]),
    "ms_relax_tilts": (ctypes.c_int, [_P, ctypes.POINTER(ms_tilt_relax_params),
                                      ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "ms_set_leaflet_tilts": (ctypes.c_int, [_P, ctypes.c_int, _D, ctypes.POINTER(ctypes.c_uint8),
                                            ctypes.POINTER(ms_leaflet_params)]),
    "ms_get_leaflet_tilts": (ctypes.c_int, [_P, ctypes.c_int, _D]),
    "ms_set_leaflet_bending": (ctypes.c_int, [_P, ctypes.c_int, _D, _D]),
    "ms_set_leaflet_disk_target": (ctypes.c_int, [_P, ctypes.c_int, ctypes.POINTER(ctypes.c_uint8),
                                                  ctypes.POINTER(ms_disk_target_params)]),
    "ms_leaflet_tilt_energy_and_gradient": (ctypes.c_int, [_P, _D, _D, _D]),
    "ms_leaflet_tilt_energy_and_gradient_ex": (ctypes.c_int, [_P, ctypes.c_int, _D, _D, _D]),
    "ms_relax_leaflet_tilts": (ctypes.c_int, [_P, ctypes.POINTER(ms_tilt_relax_params),
                                              ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "ms_set_positions": (ctypes.c_int, [_P, _D]),
    "ms_get_positions": (ctypes.c_int, [_P, _D]),
    "ms_get_gradient": (ctypes.c_int, [_P, _D]),
    "ms_get_vertex_buffer": (ctypes.c_int, [_P, ctypes.c_int, _D]),
    "ms_energy_and_gradient": (ctypes.c_int, [_P, _D, _D]),
    "ms_energy_and_raw_gradient": (ctypes.c_int, [_P, _D, _D]),
    "ms_energy": (ctypes.c_int, [_P, _D]),
    "ms_step": (ctypes.c_int, [_P, ctypes.POINTER(ms_stepper_params), ctypes.c_double,
                               ctypes.c_double, ctypes.POINTER(ms_step_result)]),
    "ms_reset_stepper": (ctypes.c_int, [_P]),
    "ms_minimize": (ctypes.c_int, [_P, ctypes.POINTER(ms_minimize_params), ctypes.c_int,
                                   ctypes.POINTER(ms_minimize_result), _D]),
    "ms_project_volume": (ctypes.c_int, [_P, ctypes.c_double, ctypes.c_double, ctypes.c_int,
                                         ctypes.POINTER(ctypes.c_int), _D]),
    "ms_project_volume_cached": (ctypes.c_int, [_P, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int,
                                                ctypes.POINTER(ctypes.c_int), _D]),
    "ms_phase_energy": (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_double, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_int]),
    "ms_phase_gradient": (ctypes.c_int, [_P]),
    "ms_phase_direction": (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_int]),
    "ms_phase_accept": (ctypes.c_int, [_P, ctypes.c_int]),
    "ms_phase_commit_trial": (ctypes.c_int, [_P, ctypes.c_double, ctypes.c_int]),
    "ms_phase_gradient_direction": (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_int]),
    "ms_phase_set_factors_valid": (ctypes.c_int, [_P, ctypes.c_int]),
    "ms_boundary_info": (ctypes.c_int, [_P, _I64]),
    "ms_exchange_bytes": (ctypes.c_size_t, [_P, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    "ms_pack_boundary": (ctypes.c_int, [_P, ctypes.c_int, ctypes.POINTER(ctypes.c_int), _P,
                                        ctypes.c_size_t]),
    "ms_unpack_boundary": (ctypes.c_int, [_P, ctypes.c_int, ctypes.POINTER(ctypes.c_int), _P,
                                          ctypes.c_size_t, _D]),
    "ms_shard_unique_id": (ctypes.c_int, [_P]),
    "ms_shard_comm_init": (ctypes.c_int, [_P, _P]),
    "ms_shard_set_allgather": (ctypes.c_int, [_P, ALLGATHER_FN, _P]),
    "ms_shard_step": (ctypes.c_int, [_P, ctypes.POINTER(ms_stepper_params), ctypes.c_double, ctypes.c_double,
                                     ctypes.POINTER(ms_step_result)]),
    "ms_shard_exchange_count": (ctypes.c_int64, [_P]),
    "ms_shard_chain_stats": (ctypes.c_int, [_P, _I64]),
    "ms_shard_comm_ranks": (ctypes.c_int, [_P]),
    "ms_shard_peer_memory_kind": (ctypes.c_int, [_P]),
    "ms_state_bytes": (ctypes.c_size_t, [_P]),
    "ms_rebind_state": (ctypes.c_int, [_P, _P, ctypes.c_size_t]),
    "ms_fetch_scalars": (ctypes.c_int, [_P, _D]),
    "ms_store_scalars": (ctypes.c_int, [_P, _D]),
    "ms_device_buffer": (ctypes.c_int, [_P, ctypes.c_int, ctypes.POINTER(_P),
                                        ctypes.POINTER(ctypes.c_size_t)]),
    "ms_shard_info": (ctypes.c_int, [_P, _I64, _I64, _I64, _I64]),
    "ms_tile_stats": (ctypes.c_int, [_P, _I64, _I64, _I64, _I64, _I64]),
    "ms_profile_enable": (ctypes.c_int, [_P, ctypes.c_int]),
    "ms_profile_read": (ctypes.c_int, [_P, _D, _I64]),
    "ms_queue_stats": (ctypes.c_int, [_P, _I64]),
    "ms_exec_stats": (ctypes.c_int, [_P, _I64]),
    "ms_resident_stats": (ctypes.c_int, [_P, _I64]),
    "ms_tsearch_stats": (ctypes.c_int, [_P, _I64]),
    "ms_exec_trace": (ctypes.c_int, [_P, ctypes.c_int, _D, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    "ms_shard_peer_export": (ctypes.c_int, [_P, _P]),
    "ms_shard_peer_open": (ctypes.c_int, [_P, _P]),
    "ms_shard_peer_local": (ctypes.c_int, [_P, ctypes.POINTER(_P), ctypes.POINTER(_P)]),
    "ms_shard_peer_set_pointers": (ctypes.c_int, [_P, ctypes.POINTER(_P), ctypes.POINTER(_P)]),
    "ms_shard_peer_set_barrier": (ctypes.c_int, [_P, _P, _P]),
    "ms_plan_tiling": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _D, _I32, ctypes.c_int, ctypes.c_int,
                                      _I64, _I32]),
    "ms_plan_tiling_conflicts": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _D, _I32, ctypes.c_int, _D]),
    "ms_surface_energy_and_gradient_host": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _D, _I32, _D,
                                                           _D, _D]),
    "ms_grad_cotan_batch_host": (ctypes.c_int, [ctypes.c_int, _D, _D, _D, _D]),
    "ms_apply_beltrami_laplacian_host": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, _D,
                                                        _I32, _D, _D]),
    "ms_p1_triangle_divergence_host": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _D, _D, _I32, _D,
                                                      _D, _D, _D, _D]),
    "ms_compute_curvature_data_host": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _D, _I32, _D, _D,
                                                      _D, _D, _D, _D]),
}

_lib = None


def build(force: bool = False, fp_contract: str | None = None) -> str:
    """Compile libmembrane_hip.so for gfx950 with hipcc (csrc/Makefile)."""
    srcs = [os.path.join(_CSRC, f) for f in ("ms_kernels.hip", "ms_api.cpp", "ms_tiles.cpp",
                                             "ms_internal.h", "Makefile")]
    srcs.append(os.path.join(_PKG, "..", "include", "membrane_hip.h"))
    stale = (not os.path.exists(LIB_PATH)) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        cmd = ["make", "-s", "-C", _CSRC]
        if force:
            cmd.append("-B")
        if fp_contract:
            cmd.append(f"MS_FP_CONTRACT={fp_contract}")
        subprocess.check_call(cmd)
    return LIB_PATH


HIP_RUNTIME_PATH = None


def _preload_hip_runtime() -> str:
    """Load exactly ONE HIP runtime into the process, globally, before our library.

    libmembrane_hip.so carries no DT_NEEDED on libamdhip64 (csrc/Makefile).  PyTorch
    wheels bundle their own libamdhip64/libhsa-runtime64; if this process will also use
    torch (device memory for RCCL, bench.py) both must share one runtime, whichever is
    imported first, or the second one to start finds no GPU.  So: torch's copy when
    torch is installed, else the system ROCm one.  MEMBRANE_HIP_RUNTIME overrides.
    """
    global HIP_RUNTIME_PATH
    if HIP_RUNTIME_PATH is not None:
        return HIP_RUNTIME_PATH
    candidates = []
    if os.environ.get("MEMBRANE_HIP_RUNTIME"):
        candidates.append(os.environ["MEMBRANE_HIP_RUNTIME"])
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
        if spec is not None and spec.submodule_search_locations:
            candidates.append(os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so"))
    except Exception:  # pragma: no cover - torch is optional plumbing
        pass
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    candidates += [os.path.join(rocm, "lib", "libamdhip64.so"), "libamdhip64.so.7", "libamdhip64.so"]
    errors = []
    for c in candidates:
        if os.path.isabs(c) and not os.path.exists(c):
            continue
        try:
            ctypes.CDLL(c, mode=ctypes.RTLD_GLOBAL)
            HIP_RUNTIME_PATH = c
            return c
        except OSError as e:
            errors.append(f"{c}: {e}")
    raise MembraneHipError("no HIP runtime (libamdhip64) could be loaded: " + "; ".join(errors))


def lib() -> ctypes.CDLL:
    """Load the shared library; fail loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MembraneHipError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback for this path.")
        _preload_hip_runtime()
        cd = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(cd, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = cd
    return _lib


def check(rc: int, ctx=None, what: str = "") -> None:
    if rc == MS_OK:
        return
    msg = lib().ms_last_error(ctx)
    text = msg.decode("utf-8", "replace") if msg else ""
    raise MembraneHipError(f"{what or 'libmembrane_hip'} failed: {MS_ERR.get(rc, rc)}: {text}")
