"""ctypes access to oracle/_ref/*.so: the reference's OWN Fortran kernels,
compiled in place by ``make -C oracle ref`` (amdflang; nothing is copied).

TEST INFRASTRUCTURE ONLY.  Used to validate the C restatement against the real
reference code and, on the GPU box, as the "reference" CPU timing for the
kernels the Fortran covers.  ``available()`` is False when the shared objects
were not built (no reference checkout / no Fortran compiler).

Symbols are flang-mangled (``_QM<module>P<procedure>``); every argument is
passed by reference; an absent ``optional`` argument is a NULL pointer.
"""

from __future__ import annotations

import ctypes
import os

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")
_libs: dict = {}

_D = ctypes.POINTER(ctypes.c_double)
_I = ctypes.POINTER(ctypes.c_int32)


def _load(name: str):
    if name not in _libs:
        path = os.path.join(_DIR, f"lib{name}.so")
        _libs[name] = ctypes.CDLL(path) if os.path.exists(path) else None
    return _libs[name]


def available() -> bool:
    return all(_load(n) is not None for n in ("surface_energy", "bending_kernels", "tilt_kernels"))


def _ci(v):
    return ctypes.byref(ctypes.c_int32(int(v)))


def _pd(a):
    return None if a is None else a.ctypes.data_as(_D)


def _pi(a):
    return a.ctypes.data_as(_I)


def surface_energy_and_gradient(pos, tri, gamma, grad) -> float:
    """fortran_kernels/surface_energy.f90:27  (pos/tri/grad as (n,3) C-order == (3,n) F-order)."""
    fn = _load("surface_energy")._QMsurface_energy_modPsurface_energy_and_gradient
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    tri = np.ascontiguousarray(tri, dtype=np.int32)
    gamma = np.ascontiguousarray(gamma, dtype=np.float64)
    E = ctypes.c_double(0.0)
    fn(_ci(pos.shape[0]), _ci(tri.shape[0]), _pd(pos), _pi(tri), _pd(gamma), _pd(grad),
       ctypes.byref(E), _ci(1))
    return float(E.value)


def grad_cotan_batch(u, v):
    fn = _load("bending_kernels")._QMbending_kernels_modPgrad_cotan_batch
    u = np.ascontiguousarray(u, dtype=np.float64)
    v = np.ascontiguousarray(v, dtype=np.float64)
    gu, gv = np.zeros_like(u), np.zeros_like(v)
    fn(_ci(u.shape[0]), _pd(u), _pd(v), _pd(gu), _pd(gv))
    return gu, gv


def apply_beltrami_laplacian(weights, tri, field):
    fn = _load("bending_kernels")._QMbending_kernels_modPapply_beltrami_laplacian
    weights = np.ascontiguousarray(weights, dtype=np.float64)
    tri = np.ascontiguousarray(tri, dtype=np.int32)
    field = np.ascontiguousarray(field, dtype=np.float64)
    out = np.zeros_like(field)
    dim = 1 if field.ndim == 1 else field.shape[1]
    fn(_ci(dim), _ci(field.shape[0]), _ci(tri.shape[0]), _pd(weights), _pi(tri), _pd(field),
       _pd(out), _ci(1))
    return out


def p1_triangle_divergence(pos, tilts, tri):
    fn = _load("tilt_kernels")._QMtilt_kernels_modPp1_triangle_divergence
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    tilts = np.ascontiguousarray(tilts, dtype=np.float64)
    tri = np.ascontiguousarray(tri, dtype=np.int32)
    nf = tri.shape[0]
    div, area = np.zeros(nf), np.zeros(nf)
    g0, g1, g2 = np.zeros((nf, 3)), np.zeros((nf, 3)), np.zeros((nf, 3))
    fn(_ci(pos.shape[0]), _ci(nf), _pd(pos), _pd(tilts), _pi(tri), _pd(div), _pd(area),
       _pd(g0), _pd(g1), _pd(g2), _ci(1))
    return div, area, g0, g1, g2


def compute_curvature_data(pos, tri):
    fn = _load("tilt_kernels")._QMtilt_kernels_modPcompute_curvature_data
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    tri = np.ascontiguousarray(tri, dtype=np.int32)
    nv, nf = pos.shape[0], tri.shape[0]
    k, A, w = np.zeros((nv, 3)), np.zeros(nv), np.zeros((nf, 3))
    va0, va1, va2 = np.zeros(nf), np.zeros(nf), np.zeros(nf)
    fn(_ci(nv), _ci(nf), _pd(pos), _pi(tri), _pd(k), _pd(A), _pd(w), _ci(1),
       _pd(va0), _pd(va1), _pd(va2))
    return k, A, w, va0, va1, va2
