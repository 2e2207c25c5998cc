"""ctypes front-end for the CPU oracle (oracle/ms_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under membrane_solver_amd/ imports it.

Each wrapper takes/returns NumPy arrays in the reference's row order
((n,3) float64 C-order, int32 triangle rows) and cites the reference routine
the underlying C function restates.
"""

from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libms_oracle.so")
_lib = None

_D = ctypes.POINTER(ctypes.c_double)
_I = ctypes.POINTER(ctypes.c_int32)
_B = ctypes.POINTER(ctypes.c_uint8)


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "ms_oracle.c")
    if (
        force
        or not os.path.exists(_LIB_PATH)
        or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)
        or not os.path.exists(os.path.join(_HERE, "libms_oracle_omp.so"))
        or not os.path.exists(os.path.join(_HERE, "libms_oracle_ld.so"))
        or os.path.getmtime(os.path.join(_HERE, "libms_oracle_ld.so")) < os.path.getmtime(src)
    ):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB_PATH


def _load(path: str) -> ctypes.CDLL:
    h = ctypes.CDLL(path)
    h.orc_volume.restype = ctypes.c_double
    h.orc_bending_energy_and_gradient.restype = ctypes.c_int
    h.orc_bending_energy.restype = ctypes.c_int
    h.orc_bending_backprop.restype = ctypes.c_int
    return h


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = _load(_LIB_PATH)
    return _lib


def use_openmp(on: bool) -> None:
    """Switch every wrapper below to libms_oracle_omp.so (the same source built with
    -fopenmp: facet loops over all host cores, vertex sums by `omp atomic`) or back to the
    serial checker.  The OpenMP build is for bench.py's all-cores cpu_baseline leg ONLY:
    its sums arrive in a different order, so nothing is ever checked against it."""
    global _lib
    build()
    _lib = _load(os.path.join(_HERE, "libms_oracle_omp.so") if on else _LIB_PATH)


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != shape:
        raise ValueError(f"expected shape {shape}, got {a.shape}")
    return a


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def _pd(a):
    return None if a is None else a.ctypes.data_as(_D)


def _pi(a):
    return None if a is None else a.ctypes.data_as(_I)


def _pb(a):
    return None if a is None else a.ctypes.data_as(_B)


# --- surface ---------------------------------------------------------------
def surface_energy_and_gradient(pos, tri, gamma, grad=None) -> float:
    """fortran_kernels/surface_energy.f90:27-99 (grad accumulated in place)."""
    pos, tri, gamma = _f64(pos), _i32(tri), _f64(gamma)
    if grad is not None and not (
        grad.dtype == np.float64 and grad.flags["C_CONTIGUOUS"]
    ):
        raise ValueError("grad must be C-contiguous float64 (in-place accumulate)")
    E = ctypes.c_double(0.0)
    lib().orc_surface_energy_and_gradient(
        ctypes.c_int(pos.shape[0]), ctypes.c_int(tri.shape[0]), _pd(pos), _pi(tri),
        _pd(gamma), _pd(grad), ctypes.byref(E),
    )
    return float(E.value)


# --- bending kernels -----------------------------------------------------------
def grad_cotan_batch(u, v):
    """fortran_kernels/bending_kernels.f90:32-74."""
    u, v = _f64(u), _f64(v)
    gu, gv = np.empty_like(u), np.empty_like(v)
    lib().orc_grad_cotan_batch(ctypes.c_int(u.shape[0]), _pd(u), _pd(v), _pd(gu), _pd(gv))
    return gu, gv


def apply_beltrami_laplacian(weights, tri, field):
    """fortran_kernels/bending_kernels.f90:87-131."""
    weights, tri, field = _f64(weights), _i32(tri), _f64(field)
    out = np.empty_like(field)
    dim = 1 if field.ndim == 1 else field.shape[1]
    lib().orc_apply_beltrami_laplacian(
        ctypes.c_int(dim), ctypes.c_int(field.shape[0]), ctypes.c_int(tri.shape[0]),
        _pd(weights), _pi(tri), _pd(field), _pd(out),
    )
    return out


def p1_triangle_divergence(pos, tilts, tri):
    """fortran_kernels/tilt_kernels.f90:26-86 -> (div, area, g0, g1, g2)."""
    pos, tilts, tri = _f64(pos), _f64(tilts), _i32(tri)
    nf = tri.shape[0]
    div = np.empty(nf)
    area = np.empty(nf)
    g0, g1, g2 = np.empty((nf, 3)), np.empty((nf, 3)), np.empty((nf, 3))
    lib().orc_p1_triangle_divergence(
        ctypes.c_int(pos.shape[0]), ctypes.c_int(nf), _pd(pos), _pd(tilts), _pi(tri),
        _pd(div), _pd(area), _pd(g0), _pd(g1), _pd(g2),
    )
    return div, area, g0, g1, g2


def compute_curvature_data(pos, tri, want_corner_areas: bool = False):
    """fortran_kernels/tilt_kernels.f90:88-190 -> (k_vecs, vertex_areas, weights[, va0, va1, va2])."""
    pos, tri = _f64(pos), _i32(tri)
    nv, nf = pos.shape[0], tri.shape[0]
    k = np.empty((nv, 3))
    A = np.empty(nv)
    w = np.empty((nf, 3))
    va = [np.empty(nf) for _ in range(3)] if want_corner_areas else [None] * 3
    lib().orc_compute_curvature_data(
        ctypes.c_int(nv), ctypes.c_int(nf), _pd(pos), _pi(tri), _pd(k), _pd(A), _pd(w),
        _pd(va[0]), _pd(va[1]), _pd(va[2]),
    )
    if want_corner_areas:
        return k, A, w, va[0], va[1], va[2]
    return k, A, w


def effective_areas(pos, tri, weights, is_boundary):
    """modules/energy/bending_utils.py:37-171 -> (vertex_areas_eff, va_eff (nf,3))."""
    pos, tri, weights = _f64(pos), _i32(tri), _f64(weights)
    isb = _u8(is_boundary)
    A = np.empty(pos.shape[0])
    va = np.empty((tri.shape[0], 3))
    lib().orc_effective_areas(
        ctypes.c_int(pos.shape[0]), ctypes.c_int(tri.shape[0]), _pd(pos), _pi(tri),
        _pd(weights), _pb(isb), _pd(A), _pd(va),
    )
    return A, va


def vertex_normals(pos, tri):
    """modules/energy/bending_utils.py:13-34."""
    pos, tri = _f64(pos), _i32(tri)
    n = np.empty_like(pos)
    lib().orc_vertex_normals(ctypes.c_int(pos.shape[0]), ctypes.c_int(tri.shape[0]), _pd(pos), _pi(tri), _pd(n))
    return n


_MODEL = {"helfrich": 0, "willmore": 1}
_MODE = {"analytic": 0, "approx": 1}


def bending_energy_and_gradient(pos, tri, kappa, c0, is_boundary, *, model="helfrich",
                                mode="analytic", grad=None, want_factors=False):
    """modules/energy/bending.py:90-181 (+ bending_gradient.py:17-175)."""
    pos, tri = _f64(pos), _i32(tri)
    nv = pos.shape[0]
    kappa, c0 = _f64(kappa, (nv,)), _f64(c0, (nv,))
    isb = _u8(is_boundary)
    if grad is not None and not (grad.dtype == np.float64 and grad.flags["C_CONTIGUOUS"]):
        raise ValueError("grad must be C-contiguous float64")
    fK = np.empty((nv, 3)) if want_factors else None
    fAe = np.empty(nv) if want_factors else None
    fAv = np.empty(nv) if want_factors else None
    E = ctypes.c_double(0.0)
    rc = lib().orc_bending_energy_and_gradient(
        ctypes.c_int(nv), ctypes.c_int(tri.shape[0]), _pd(pos), _pi(tri), _pd(kappa),
        _pd(c0), _pb(isb), ctypes.c_int(_MODEL[model]), ctypes.c_int(_MODE[mode]),
        _pd(grad), ctypes.byref(E), _pd(fK), _pd(fAe), _pd(fAv),
    )
    if rc != 0:
        raise MemoryError("oracle allocation failed")
    if want_factors:
        return float(E.value), fK, fAe, fAv
    return float(E.value)


def bending_backprop(pos, tri, is_boundary, fA_eff, fA_vor, fK, grad) -> None:
    """modules/energy/bending_gradient.py:17-175 with given factors (grad accumulated)."""
    pos, tri = _f64(pos), _i32(tri)
    isb = _u8(is_boundary)
    fae, fav, fk = _f64(fA_eff), _f64(fA_vor), _f64(fK)
    rc = lib().orc_bending_backprop(ctypes.c_int(pos.shape[0]), ctypes.c_int(tri.shape[0]), _pd(pos),
                                    _pi(tri), _pb(isb), _pd(fae), _pd(fav), _pd(fk), _pd(grad))
    if rc != 0:
        raise MemoryError("oracle allocation failed")


def bending_energy(pos, tri, kappa, c0, is_boundary, *, model="helfrich", per_vertex=False):
    """modules/energy/bending.py:62-87 compute_energy_array."""
    pos, tri = _f64(pos), _i32(tri)
    nv = pos.shape[0]
    kappa, c0 = _f64(kappa, (nv,)), _f64(c0, (nv,))
    isb = _u8(is_boundary)
    pv = np.empty(nv) if per_vertex else None
    E = ctypes.c_double(0.0)
    rc = lib().orc_bending_energy(
        ctypes.c_int(nv), ctypes.c_int(tri.shape[0]), _pd(pos), _pi(tri), _pd(kappa),
        _pd(c0), _pb(isb), ctypes.c_int(_MODEL[model]), ctypes.byref(E), _pd(pv),
    )
    if rc != 0:
        raise MemoryError("oracle allocation failed")
    return (float(E.value), pv) if per_vertex else float(E.value)


# --- volume -------------------------------------------------------------------
def volume(pos, tri, body_rows=None) -> float:
    """geometry/body.py:70-148 (vectorised branch)."""
    pos, tri = _f64(pos), _i32(tri)
    rows = None if body_rows is None else _i32(body_rows)
    n = tri.shape[0] if rows is None else rows.shape[0]
    return float(lib().orc_volume(ctypes.c_int(pos.shape[0]), ctypes.c_int(n), _pd(pos), _pi(tri), _pi(rows)))


def volume_gradient(pos, tri, grad, factor=1.0, body_rows=None) -> None:
    """geometry/body.py:150-190 accumulate_volume_gradient (in place)."""
    pos, tri = _f64(pos), _i32(tri)
    rows = None if body_rows is None else _i32(body_rows)
    n = tri.shape[0] if rows is None else rows.shape[0]
    if not (grad.dtype == np.float64 and grad.flags["C_CONTIGUOUS"]):
        raise ValueError("grad must be C-contiguous float64")
    lib().orc_volume_gradient(
        ctypes.c_int(pos.shape[0]), ctypes.c_int(n), _pd(pos), _pi(tri), _pi(rows),
        ctypes.c_double(factor), _pd(grad),
    )


# --- tilt -----------------------------------------------------------------------
def tilt_energy_and_gradient(pos, tilts, tri, k_tilt, grad=None, tilt_grad=None) -> float:
    """modules/energy/tilt.py:99-172."""
    pos, tilts, tri = _f64(pos), _f64(tilts), _i32(tri)
    E = ctypes.c_double(0.0)
    lib().orc_tilt_energy_and_gradient(
        ctypes.c_int(pos.shape[0]), ctypes.c_int(tri.shape[0]), _pd(pos), _pd(tilts),
        _pi(tri), ctypes.c_double(k_tilt), _pd(grad), _pd(tilt_grad), ctypes.byref(E),
    )
    return float(E.value)


# --- bending + tilt-splay coupling ------------------------------------------------
def bending_tilt_energy_and_gradient(pos, tilts, tri, kappa, c0, is_boundary, *, mode="analytic",
                                     grad=None, tilt_grad=None) -> float:
    """modules/energy/bending_tilt.py:151-482: E = 1/2 sum_f sum_k kappa_k (2H_k - c0_k + div_f t)^2
    va_eff[f,k] on the bending module's discretisation; shape gradient = bending back-propagation
    with term -> base + div_eff (div treated as constant in x, :24-28), tilt gradient exact.
    Composition of the C oracle's kernels; the per-vertex / per-corner arithmetic is NumPy."""
    pos, tilts, tri = _f64(pos), _f64(tilts), _i32(tri)
    nv = pos.shape[0]
    kappa, c0 = _f64(kappa, (nv,)), _f64(c0, (nv,))
    isb = np.asarray(is_boundary, dtype=bool)
    if tri.shape[0] == 0:
        return 0.0
    k_vecs, A_vor, weights = compute_curvature_data(pos, tri)          # :170-172
    div_tri, _area, g0, g1, g2 = p1_triangle_divergence(pos, tilts, tri)  # :194-205
    A_eff, va_eff = effective_areas(pos, tri, weights, isb)              # :208-210
    safe = np.maximum(A_vor, 1e-12)
    k_mag = np.linalg.norm(k_vecs, axis=1)
    H = k_mag / (2.0 * safe)
    ratio = np.zeros_like(A_eff)
    m = safe > 1e-15
    ratio[m] = A_eff[m] / safe[m]
    base = (2.0 * H) - c0
    base[isb] = 0.0
    term_tri = base[tri] + div_tri[:, None]
    kappa_tri = kappa[tri]
    E = float(0.5 * np.sum(kappa_tri * term_tri**2 * va_eff))           # :236-239
    if tilt_grad is not None:                                           # :438-448 / :258-268
        f = np.sum(kappa_tri * term_tri * va_eff, axis=1)[:, None]
        np.add.at(tilt_grad, tri[:, 0], f * g0)
        np.add.at(tilt_grad, tri[:, 1], f * g1)
        np.add.at(tilt_grad, tri[:, 2], f * g2)
    if grad is None:
        return E
    num = np.zeros(nv)                                                   # :243-253
    for kcol in range(3):
        np.add.at(num, tri[:, kcol], va_eff[:, kcol] * div_tri)
    div_eff = np.zeros(nv)
    me = A_eff > 1e-20
    div_eff[me] = num[me] / A_eff[me]
    term = base + div_eff
    term[isb] = 0.0
    normals = vertex_normals(pos, tri)                                   # :271-276
    K_dir = np.zeros_like(k_vecs)
    mk = k_mag > 1e-15
    K_dir[mk] = k_vecs[mk] / k_mag[mk][:, None]
    K_dir[~mk] = normals[~mk]
    fK = np.ascontiguousarray(K_dir * (kappa * term * ratio)[:, None])
    fA_eff = 0.5 * kappa * term**2
    fA_vor = -2.0 * kappa * term * ratio * H
    if mode == "approx":                                                 # :296-299
        grad -= apply_beltrami_laplacian(weights, tri, fK)
        grad[isb] = 0.0
    else:
        bending_backprop(pos, tri, isb, fA_eff, fA_vor, fK, grad)        # :300-436
    return E


# --- tilt smoothness --------------------------------------------------------------
def tilt_smoothness_energy_and_gradient(pos, tilts, tri, k_smooth, tilt_grad=None) -> float:
    """modules/energy/tilt_smoothness.py:84-198 (ambient_v1 transport): cotangent Dirichlet energy
    E = k_s/4 sum_f [c0|t1-t2|^2 + c1|t2-t0|^2 + c2|t0-t1|^2] with the cotans of
    compute_curvature_data at `pos`; exact tilt gradient, NO shape gradient (:21-23)."""
    pos, tilts, tri = _f64(pos), _f64(tilts), _i32(tri)
    if k_smooth == 0.0 or tri.shape[0] == 0:
        return 0.0
    _k, _a, w = compute_curvature_data(pos, tri)
    c0, c1, c2 = w[:, 0], w[:, 1], w[:, 2]
    t0, t1, t2 = tilts[tri[:, 0]], tilts[tri[:, 1]], tilts[tri[:, 2]]
    d12, d20, d01 = t1 - t2, t2 - t0, t0 - t1
    n12 = np.einsum("ij,ij->i", d12, d12)
    n20 = np.einsum("ij,ij->i", d20, d20)
    n01 = np.einsum("ij,ij->i", d01, d01)
    E = float(0.25 * k_smooth * np.sum(c0 * n12 + c1 * n20 + c2 * n01))
    if tilt_grad is not None:
        f = 0.5 * k_smooth
        np.add.at(tilt_grad, tri[:, 0], f * (c1[:, None] * (t0 - t2) + c2[:, None] * (t0 - t1)))
        np.add.at(tilt_grad, tri[:, 1], f * (c2[:, None] * (t1 - t0) + c0[:, None] * (t1 - t2)))
        np.add.at(tilt_grad, tri[:, 2], f * (c0[:, None] * (t2 - t1) + c1[:, None] * (t2 - t0)))
    return E


def tilt_leaflet_energy_and_gradient(pos, tilts, tri, k_tilt, mass_mode="lumped", grad=None, tilt_grad=None) -> float:
    """modules/energy/tilt_leaflet.py:26-169 (tilt_in / tilt_out), default options: no absent-leaflet
    presets, no active-row weights, no shared-rim shell mode.  NumPy restatement.
    lumped: coeff_f = 1/2 k (sum |t_k|^2)/3 ; consistent: coeff_f = k/12 (sum |t_k|^2 + t0.t1 + t1.t2 + t2.t0);
    E = sum coeff_f A_f over facets with |n| >= 1e-12; shape gradient coeff_f dA/dx; tilt gradient
    k A/3 t_k (lumped) or k A/12 (2 t_k + t_a + t_b) (consistent)."""
    pos = _f64(pos)
    tilts = _f64(tilts)
    tri = _i32(tri)
    if k_tilt == 0.0 or tri.shape[0] == 0:
        return 0.0
    v0, v1, v2 = pos[tri[:, 0]], pos[tri[:, 1]], pos[tri[:, 2]]
    n = np.cross(v1 - v0, v2 - v0)
    n_norm = np.linalg.norm(n, axis=1)
    mask = n_norm >= 1e-12  # tilt_utils._triangle_geometry
    if not np.any(mask):
        return 0.0
    rows = tri[mask]
    areas = 0.5 * n_norm[mask]
    t0, t1, t2 = tilts[rows[:, 0]], tilts[rows[:, 1]], tilts[rows[:, 2]]
    sq = np.einsum("ij,ij->i", t0, t0) + np.einsum("ij,ij->i", t1, t1) + np.einsum("ij,ij->i", t2, t2)
    if mass_mode == "consistent":
        cs = sq + np.einsum("ij,ij->i", t0, t1) + np.einsum("ij,ij->i", t1, t2) + np.einsum("ij,ij->i", t2, t0)
        coeff = (k_tilt / 12.0) * cs
    else:
        coeff = 0.5 * k_tilt * (sq / 3.0)
    energy = float(np.dot(coeff, areas))
    if tilt_grad is not None:
        if mass_mode == "consistent":
            f = (k_tilt * areas / 12.0)[:, None]
            np.add.at(tilt_grad, rows[:, 0], f * (2.0 * t0 + t1 + t2))
            np.add.at(tilt_grad, rows[:, 1], f * (2.0 * t1 + t2 + t0))
            np.add.at(tilt_grad, rows[:, 2], f * (2.0 * t2 + t0 + t1))
        else:
            f = (k_tilt * areas / 3.0)[:, None]
            np.add.at(tilt_grad, rows[:, 0], f * t0)
            np.add.at(tilt_grad, rows[:, 1], f * t1)
            np.add.at(tilt_grad, rows[:, 2], f * t2)
    if grad is not None:
        n_hat = n[mask] / n_norm[mask][:, None]
        m0, m1, m2 = v0[mask], v1[mask], v2[mask]
        c = coeff[:, None]
        np.add.at(grad, rows[:, 0], c * (0.5 * np.cross(n_hat, m2 - m1)))
        np.add.at(grad, rows[:, 1], c * (0.5 * np.cross(n_hat, m0 - m2)))
        np.add.at(grad, rows[:, 2], c * (0.5 * np.cross(n_hat, m1 - m0)))
    return energy


def barycentric_vertex_areas(pos, tri):
    """Mesh.barycentric_vertex_areas (geometry/mesh.py:671-730): A_v = sum_f A_f / 3 over |n| >= 1e-12."""
    pos = _f64(pos)
    tri = _i32(tri)
    n = np.cross(pos[tri[:, 1]] - pos[tri[:, 0]], pos[tri[:, 2]] - pos[tri[:, 0]])
    n_norm = np.linalg.norm(n, axis=1)
    mask = n_norm >= 1e-12  # mesh.py:703-707
    thirds = 0.5 * n_norm[mask] / 3.0
    rows = tri[mask]
    va = np.zeros(pos.shape[0])
    for k in range(3):
        np.add.at(va, rows[:, k], thirds)
    return va


# --- leaflet bending + tilt-splay coupling (bending_tilt_in / bending_tilt_out) ----
def unit_vertex_normals_tri(pos, tri):
    """bending_utils._vertex_normals restated on arrays (same routine the C oracle exposes)."""
    return vertex_normals(pos, tri)


def _backprop_corner_areas(pos, tri, weights, is_boundary, corner_fA_eff, fA_vor, fK, grad):
    """modules/energy/bt_gradient.py:139-387 with tri_rows == tri_rows_full (no absent-leaflet mask, no
    transition operator): the bending back-propagation whose effective-area factor is PER CORNER."""
    i0, i1, i2 = tri[:, 0], tri[:, 1], tri[:, 2]
    v0, v1, v2 = pos[i0], pos[i1], pos[i2]
    e0, e1, e2 = v2 - v1, v0 - v2, v1 - v0
    c0, c1, c2 = weights[:, 0], weights[:, 1], weights[:, 2]
    grad_linear = -apply_beltrami_laplacian(weights, tri, fK)
    dE_dc0 = -0.5 * np.einsum("ij,ij->i", fK[i1] - fK[i2], v1 - v2)
    dE_dc1 = -0.5 * np.einsum("ij,ij->i", fK[i2] - fK[i0], v2 - v0)
    dE_dc2 = -0.5 * np.einsum("ij,ij->i", fK[i0] - fK[i1], v0 - v1)
    g0u, g0v = grad_cotan_batch(v1 - v0, v2 - v0)
    g1u, g1v = grad_cotan_batch(v2 - v1, v0 - v1)
    g2u, g2v = grad_cotan_batch(v0 - v2, v1 - v2)
    grad_cot = np.zeros_like(pos)
    a0, a1, a2 = dE_dc0[:, None], dE_dc1[:, None], dE_dc2[:, None]
    np.add.at(grad_cot, i1, a0 * g0u)
    np.add.at(grad_cot, i2, a0 * g0v)
    np.add.at(grad_cot, i0, a0 * -(g0u + g0v))
    np.add.at(grad_cot, i2, a1 * g1u)
    np.add.at(grad_cot, i0, a1 * g1v)
    np.add.at(grad_cot, i1, a1 * -(g1u + g1v))
    np.add.at(grad_cot, i0, a2 * g2u)
    np.add.at(grad_cot, i1, a2 * g2v)
    np.add.at(grad_cot, i2, a2 * -(g2u + g2v))
    tri_is_int = (~np.asarray(is_boundary, dtype=bool))[tri]
    counts = np.sum(tri_is_int, axis=1)
    sum_int = np.sum(corner_fA_eff * tri_is_int, axis=1)
    avg = np.zeros(tri.shape[0])
    m = counts > 0
    avg[m] = sum_int[m] / counts[m]
    C = np.where(tri_is_int, corner_fA_eff, avg[:, None]) + fA_vor[tri]
    grad_area = np.zeros_like(pos)
    obt = (c0 < 0) | (c1 < 0) | (c2 < 0)
    ms = ~obt
    if np.any(ms):
        c0s, c1s, c2s = c0[ms], c1[ms], c2[ms]
        C0, C1, C2 = C[ms, 0], C[ms, 1], C[ms, 2]
        E0, E1, E2 = e0[ms], e1[ms], e2[ms]
        j0, j1, j2 = i0[ms], i1[ms], i2[ms]
        for coeff, ev, plus, minus in ((0.25 * c1s * C0, E1, j0, j2), (0.25 * c2s * C0, E2, j1, j0),
                                       (0.25 * c2s * C1, E2, j1, j0), (0.25 * c0s * C1, E0, j2, j1),
                                       (0.25 * c0s * C2, E0, j2, j1), (0.25 * c1s * C2, E1, j0, j2)):
            np.add.at(grad_area, plus, coeff[:, None] * ev)
            np.add.at(grad_area, minus, -coeff[:, None] * ev)
        l0 = np.einsum("ij,ij->i", E0, E0)
        l1 = np.einsum("ij,ij->i", E1, E1)
        l2 = np.einsum("ij,ij->i", E2, E2)
        k0 = (0.125 * l0 * (C1 + C2))[:, None]
        k1 = (0.125 * l1 * (C0 + C2))[:, None]
        k2 = (0.125 * l2 * (C0 + C1))[:, None]
        h0u, h0v = grad_cotan_batch(E2, -E1)
        h1u, h1v = grad_cotan_batch(E0, -E2)
        h2u, h2v = grad_cotan_batch(E1, -E0)
        np.add.at(grad_area, j1, k0 * h0u)
        np.add.at(grad_area, j2, k0 * h0v)
        np.add.at(grad_area, j0, k0 * -(h0u + h0v))
        np.add.at(grad_area, j2, k1 * h1u)
        np.add.at(grad_area, j0, k1 * h1v)
        np.add.at(grad_area, j1, k1 * -(h1u + h1v))
        np.add.at(grad_area, j0, k2 * h2u)
        np.add.at(grad_area, j1, k2 * h2v)
        np.add.at(grad_area, j2, k2 * -(h2u + h2v))
    if np.any(obt):
        for i, sub in enumerate(((c0 < 0), (c1 < 0), (c2 < 0))):
            md = sub & obt
            if not np.any(md):
                continue
            u, v = pos[i1[md]] - pos[i0[md]], pos[i2[md]] - pos[i0[md]]
            w = np.cross(u, v)
            S = np.linalg.norm(w, axis=1)
            ok = S > 1e-15
            gu, gv = np.zeros_like(u), np.zeros_like(v)
            gu[ok] = 0.5 * np.cross(v[ok], w[ok]) / S[ok][:, None]
            gv[ok] = 0.5 * np.cross(w[ok], u[ok]) / S[ok][:, None]
            Co = C[md]
            other = [k for k in range(3) if k != i]
            factor = (0.5 * Co[:, i] + 0.25 * Co[:, other[0]] + 0.25 * Co[:, other[1]])[:, None]
            np.add.at(grad_area, i1[md], factor * gu)
            np.add.at(grad_area, i2[md], factor * gv)
            np.add.at(grad_area, i0[md], factor * -(gu + gv))
    grad += grad_linear
    grad += grad_cot
    grad += grad_area


def _ambient_p1_divergence_shape_gradient(pos, tilts, tri, coefficient, grad):
    """modules/energy/bt_gradient.py:20-64: coefficient_f * d(div_P1 t)/dx for ambient tilts."""
    v0, v1, v2 = pos[tri[:, 0]], pos[tri[:, 1]], pos[tri[:, 2]]
    a, b = v1 - v0, v2 - v0
    n = np.cross(a, b)
    n2 = np.maximum(np.einsum("ij,ij->i", n, n), 1.0e-20)
    t0, t1, t2 = tilts[tri[:, 0]], tilts[tri[:, 1]], tilts[tri[:, 2]]
    e0, e1, e2 = b - a, -b, a
    w = np.cross(e0, t0) + np.cross(e1, t1) + np.cross(e2, t2)
    ndw = np.einsum("ij,ij->i", n, w)
    ddn = w / n2[:, None] - 2.0 * ndw[:, None] * n / (n2**2)[:, None]
    d0 = np.cross(t0, n) / n2[:, None]
    d1 = np.cross(t1, n) / n2[:, None]
    d2 = np.cross(t2, n) / n2[:, None]
    da = np.cross(b, ddn) - d0 + d2
    db = np.cross(ddn, a) + d0 - d1
    f = np.asarray(coefficient, dtype=float)[:, None]
    ga, gb = f * da, f * db
    np.add.at(grad, tri[:, 1], ga)
    np.add.at(grad, tri[:, 2], gb)
    np.add.at(grad, tri[:, 0], -(ga + gb))


def bending_tilt_leaflet_energy_and_gradient(pos, tilts, tri, kappa, c0, is_boundary, div_sign, *, mode="analytic",
                                             grad=None, tilt_grad=None) -> float:
    """modules/energy/bending_tilt_leaflet.py:231-758 (bending_tilt_in: div_sign = -1, bending_tilt_out: +1),
    default options (ambient_v1 transport; no absent-leaflet presets, recovered / reconstructed divergence,
    update modes, assume-J0 presets, base-term groups, stage-A lanes or scaffold stencils):
        E = 1/2 sum_f sum_k kappa_k (base_k + s div_f t)^2 va_eff[f,k],  base = 2 H - c0 with the SIGNED
        H = (K . n)/(2 max(A_vor, 1e-12)) (:457-459), 0 on boundary rows;
    shape gradient = back-propagation with K_dir = n, per-corner fA_eff = 1/2 kappa_k term_tri^2 (:608-610) plus the
    exact s dE/ddiv d(div)/dx (:692-699); tilt gradient s (sum_k kappa_k term_tri_k va_eff_k) g_k."""
    pos, tilts, tri = _f64(pos), _f64(tilts), _i32(tri)
    nv = pos.shape[0]
    kappa, c0 = _f64(kappa, (nv,)), _f64(c0, (nv,))
    isb = np.asarray(is_boundary, dtype=bool)
    if tri.shape[0] == 0:
        return 0.0
    k_vecs, A_vor, weights = compute_curvature_data(pos, tri)
    div_tri, _area, g0, g1, g2 = p1_triangle_divergence(pos, tilts, tri)
    div_term = float(div_sign) * div_tri
    A_eff, va_eff = effective_areas(pos, tri, weights, isb)
    safe = np.maximum(A_vor, 1e-12)
    normals = vertex_normals(pos, tri)
    H = np.einsum("ij,ij->i", k_vecs, normals) / (2.0 * safe)
    base = (2.0 * H) - c0
    base[isb] = 0.0
    base_tri = base[tri]
    term_tri = base_tri + div_term[:, None]
    kappa_tri = kappa[tri]
    E = float(0.5 * np.sum(kappa_tri * term_tri**2 * va_eff))
    dE_ddiv = float(div_sign) * np.sum(kappa_tri * term_tri * va_eff, axis=1)
    if tilt_grad is not None:
        f = dE_ddiv[:, None]
        np.add.at(tilt_grad, tri[:, 0], f * g0)
        np.add.at(tilt_grad, tri[:, 1], f * g1)
        np.add.at(tilt_grad, tri[:, 2], f * g2)
    if grad is None:
        return E
    ratio = np.zeros_like(A_eff)
    m = safe > 1e-15
    ratio[m] = A_eff[m] / safe[m]
    num = np.zeros(nv)
    for kcol in range(3):
        np.add.at(num, tri[:, kcol], va_eff[:, kcol] * div_term)
    div_eff = np.zeros(nv)
    me = A_eff > 1e-20
    div_eff[me] = num[me] / A_eff[me]
    term = base + div_eff
    term[isb] = 0.0
    fK = np.ascontiguousarray(normals * (kappa * term * ratio)[:, None])
    corner_fA_eff = 0.5 * kappa_tri * term_tri**2
    fA_vor = -2.0 * kappa * term * ratio * H
    if mode != "analytic":
        # approx mode (bt_gradient.py:112-137) zeroes boundary rows of what other modules accumulated and adds
        # the d(div)/dx term only when the caller also asked for a tilt gradient (:631-633): not restated
        raise NotImplementedError("bending_tilt_in/out: only bending_gradient_mode=analytic is restated")
    _backprop_corner_areas(pos, tri, weights, isb, corner_fA_eff, fA_vor, fK, grad)
    _ambient_p1_divergence_shape_gradient(pos, tilts, tri, dE_ddiv, grad)
    return E


# --- disk tilt target (soft profile enforcement) -----------------------------------
def bessel_i1_series(x, n_terms: int = 30):
    """modules/energy/tilt_disk_target_in.py:148-157."""
    t = 0.5 * np.asarray(x, dtype=float)
    t2 = t * t
    term = t.copy()
    out = term.copy()
    for k in range(1, int(n_terms)):
        term = term * (t2 / (k * (k + 1)))
        out = out + term
    return out


def tilt_disk_target_energy_and_gradient(pos, tilts, tri, disk_rows, k_target, theta_b, lam, center, normal,
                                         radius=None, grad=None, tilt_grad=None) -> float:
    """modules/energy/tilt_disk_target_in.py:160-286 (and _out): E = sum_f 1/2 k (sum_k |t_k - target_k|^2)/3 A_f
    with target = theta(r) r_hat on the disk rows and the difference zeroed elsewhere; theta(r) =
    theta_B I1(lam r)/I1(lam R) (30-term series) or theta_B r/R when |lam| < 1e-12; R = ``radius`` or the largest
    in-plane distance of a disk row.  Shape gradient coeff_f dA/dx (the target is not differentiated);
    tilt gradient k diff_v A_v with the barycentric areas.  ``normal`` must be given (unit length is enforced
    as :67-77 does); the SVD plane fit of :80-93 is not restated."""
    pos, tilts, tri = _f64(pos), _f64(tilts), _i32(tri)
    disk_rows = np.asarray(disk_rows, dtype=int)
    if k_target == 0.0 or theta_b == 0.0 or disk_rows.size == 0 or tri.shape[0] == 0:
        return 0.0
    center = np.asarray(center, dtype=float).reshape(3)
    normal = np.asarray(normal, dtype=float).reshape(3)
    normal = normal / np.linalg.norm(normal)
    r_vec = pos[disk_rows] - center[None, :]
    r_vec = r_vec - np.einsum("ij,j->i", r_vec, normal)[:, None] * normal[None, :]
    r_len = np.linalg.norm(r_vec, axis=1)
    good = r_len > 1e-12
    if not np.any(good):
        return 0.0
    r_hat = np.zeros_like(r_vec)
    r_hat[good] = r_vec[good] / r_len[good][:, None]
    if radius is None:
        radius = float(np.max(r_len))
    if radius <= 0.0:
        return 0.0
    if abs(lam) < 1e-12:
        theta = theta_b * r_len / radius
    else:
        den = bessel_i1_series(np.array([lam * radius]))[0]
        if abs(den) < 1e-15:
            return 0.0
        theta = theta_b * bessel_i1_series(lam * r_len) / den
    diff = np.zeros_like(tilts)
    diff[disk_rows] = tilts[disk_rows] - theta[:, None] * r_hat
    diff_sq = np.einsum("ij,ij->i", diff, diff)
    v0, v1, v2 = pos[tri[:, 0]], pos[tri[:, 1]], pos[tri[:, 2]]
    n = np.cross(v1 - v0, v2 - v0)
    n_norm = np.linalg.norm(n, axis=1)
    mask = n_norm >= 1e-12
    if not np.any(mask):
        return 0.0
    areas = 0.5 * n_norm[mask]
    rows = tri[mask]
    coeff = 0.5 * k_target * (diff_sq[rows].sum(axis=1) / 3.0)
    energy = float(np.dot(coeff, areas))
    if grad is not None:
        n_hat = n[mask] / n_norm[mask][:, None]
        c = coeff[:, None]
        np.add.at(grad, rows[:, 0], c * (0.5 * np.cross(n_hat, v2[mask] - v1[mask])))
        np.add.at(grad, rows[:, 1], c * (0.5 * np.cross(n_hat, v0[mask] - v2[mask])))
        np.add.at(grad, rows[:, 2], c * (0.5 * np.cross(n_hat, v1[mask] - v0[mask])))
    if tilt_grad is not None:
        va = np.zeros(pos.shape[0])
        for kcol in range(3):
            np.add.at(va, rows[:, kcol], areas / 3.0)
        tilt_grad += k_target * diff * va[:, None]
    return energy


# --- angle defects / Gaussian curvature ------------------------------------------------
def angle_defects(pos, tri, is_boundary):
    """geometry/curvature.py:335-403 compute_angle_defects: 2 pi - sum of incident angles (law of cosines, edge
    lengths clamped at 1e-15, cosines clipped to [-1, 1]), 0 on boundary rows."""
    pos, tri = _f64(pos), _i32(tri)
    nv = pos.shape[0]
    if tri.shape[0] == 0:
        return np.zeros(nv)
    v0, v1, v2 = pos[tri[:, 0]], pos[tri[:, 1]], pos[tri[:, 2]]
    a = np.maximum(np.linalg.norm(v2 - v1, axis=1), 1e-15)
    b = np.maximum(np.linalg.norm(v0 - v2, axis=1), 1e-15)
    c = np.maximum(np.linalg.norm(v1 - v0, axis=1), 1e-15)
    ang0 = np.arccos(np.clip((b * b + c * c - a * a) / (2.0 * b * c), -1.0, 1.0))
    ang1 = np.arccos(np.clip((c * c + a * a - b * b) / (2.0 * c * a), -1.0, 1.0))
    ang2 = np.arccos(np.clip((a * a + b * b - c * c) / (2.0 * a * b), -1.0, 1.0))
    sums = np.zeros(nv)
    np.add.at(sums, tri[:, 0], ang0)
    np.add.at(sums, tri[:, 1], ang1)
    np.add.at(sums, tri[:, 2], ang2)
    defects = (2.0 * np.pi) - sums
    defects[np.asarray(is_boundary, dtype=bool)] = 0.0
    return defects


def angle_sums(pos, tri):
    """Per-vertex sums of the incident triangle angles (runtime/diagnostics/gauss_bonnet.py:204-257, the same
    law-of-cosines arithmetic as compute_angle_defects)."""
    pos, tri = _f64(pos), _i32(tri)
    nv = pos.shape[0]
    sums = np.zeros(nv)
    if tri.shape[0] == 0:
        return sums
    v0, v1, v2 = pos[tri[:, 0]], pos[tri[:, 1]], pos[tri[:, 2]]
    a = np.maximum(np.linalg.norm(v2 - v1, axis=1), 1e-15)
    b = np.maximum(np.linalg.norm(v0 - v2, axis=1), 1e-15)
    c = np.maximum(np.linalg.norm(v1 - v0, axis=1), 1e-15)
    np.add.at(sums, tri[:, 0], np.arccos(np.clip((b * b + c * c - a * a) / (2.0 * b * c), -1.0, 1.0)))
    np.add.at(sums, tri[:, 1], np.arccos(np.clip((c * c + a * a - b * b) / (2.0 * c * a), -1.0, 1.0)))
    np.add.at(sums, tri[:, 2], np.arccos(np.clip((a * a + b * b - c * c) / (2.0 * a * b), -1.0, 1.0)))
    return sums


def curvature_fields(pos, tri, is_boundary) -> dict:
    """geometry/curvature.py:404-448 compute_curvature_fields."""
    k_vecs, areas, _w = compute_curvature_data(pos, tri)
    safe = np.maximum(areas, 1e-12)
    hn = k_vecs / (2.0 * safe[:, None])
    H = np.linalg.norm(hn, axis=1)
    defect = angle_defects(pos, tri, is_boundary)
    KG = defect / safe
    root = np.sqrt(np.maximum(H * H - KG, 0.0))
    return {"mean_curvature_normal": hn, "mean_curvature": H, "mixed_area": areas, "angle_defect": defect,
            "gaussian_curvature": KG, "principal_curvatures": np.column_stack([H + root, H - root])}


def gauss_bonnet_invariant(pos, tri, is_boundary):
    """runtime/diagnostics/gauss_bonnet.py:305-340 for a manifold triangle mesh whose boundary vertices are those of
    its boundary loops: G = sum_interior (2 pi - theta_v) + sum_boundary (pi - theta_v) -> (G, interior, boundary)."""
    th = angle_sums(pos, tri)
    isb = np.asarray(is_boundary, dtype=bool)
    k_int = float(np.sum(2.0 * np.pi - th[~isb]))
    b_tot = float(np.sum(np.pi - th[isb]))
    return k_int + b_tot, k_int, b_tot


def euler_characteristic(nv, tri):
    """modules/energy/gaussian_curvature.py:41-43: V - E + F of the triangle complex."""
    tri = _i32(tri)
    e = np.sort(np.concatenate([tri[:, [0, 1]], tri[:, [1, 2]], tri[:, [2, 0]]]), axis=1)
    n_edges = np.unique(e, axis=0).shape[0] if e.size else 0
    return int(nv - n_edges + tri.shape[0])
