"""Kernel-provider seam: ``get_*_kernel() -> KernelSpec | None`` with the same
names, argument order and in-place output semantics as the reference's
fortran_kernels/loader.py:15-20,30,85,139,193,247, backed by the HIP library.

Arrays are taken in the reference's natural C-order ``(n,3)`` layouts
(``expects_transpose=False``): a row-major ``(n,3)`` array has the same bytes as
the ``(3,n)`` Fortran-order arrays the f2py kernels see, so no copies happen.
Every call is H2D + kernel + D2H (parity seam; the minimizer path keeps state
in HBM instead).  Kill switches follow the reference's style:
``MEMBRANE_DISABLE_HIP=1`` makes every getter return None.
"""

from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import Callable

import numpy as np

from .. import _lib as L


@dataclass(frozen=True)
class KernelSpec:
    """Resolved kernel callable with metadata (fortran_kernels/loader.py:15-20)."""

    func: Callable
    expects_transpose: bool


def _disabled() -> bool:
    return os.environ.get("MEMBRANE_DISABLE_HIP") in {"1", "true", "TRUE"}


def _in(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _out(a, shape, name):
    if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
            and a.shape == tuple(shape)):
        raise ValueError(f"{name} must be a C-contiguous float64 array of shape {tuple(shape)} "
                         "(intent(inout): it is written in place)")
    return a.ctypes.data_as(L._D)


def _pd(a):
    return a.ctypes.data_as(L._D)


def _pi(a):
    return a.ctypes.data_as(L._I32)


def _surface(pos, tri, gamma, grad, zero_based=1):
    """surface_energy.f90:27-99: returns E, accumulates into grad."""
    pos, tri, gamma = _in(pos, np.float64), _in(tri, np.int32), _in(gamma, np.float64)
    if not zero_based:
        tri = tri - 1
    E = ctypes.c_double(0.0)
    rc = L.lib().ms_surface_energy_and_gradient_host(pos.shape[0], tri.shape[0], _pd(pos), _pi(tri),
                                                     _pd(gamma), _out(grad, pos.shape, "grad"),
                                                     ctypes.byref(E))
    L.check(rc, None, "ms_surface_energy_and_gradient_host")
    return float(E.value)


def _grad_cotan(u, v, grad_u, grad_v):
    """bending_kernels.f90:32-74."""
    u, v = _in(u, np.float64), _in(v, np.float64)
    rc = L.lib().ms_grad_cotan_batch_host(u.shape[0], _pd(u), _pd(v), _out(grad_u, u.shape, "grad_u"),
                                          _out(grad_v, v.shape, "grad_v"))
    L.check(rc, None, "ms_grad_cotan_batch_host")


def _laplacian(weights, tri, field, out, zero_based=1):
    """bending_kernels.f90:87-131 (out is overwritten)."""
    weights, tri, field = _in(weights, np.float64), _in(tri, np.int32), _in(field, np.float64)
    if not zero_based:
        tri = tri - 1
    dim = 1 if field.ndim == 1 else field.shape[1]
    rc = L.lib().ms_apply_beltrami_laplacian_host(dim, field.shape[0], tri.shape[0], _pd(weights),
                                                  _pi(tri), _pd(field), _out(out, field.shape, "out"))
    L.check(rc, None, "ms_apply_beltrami_laplacian_host")


def _divergence(pos, tilts, tri, div_tri, area, g0, g1, g2, zero_based=1):
    """tilt_kernels.f90:26-86."""
    pos, tilts, tri = _in(pos, np.float64), _in(tilts, np.float64), _in(tri, np.int32)
    if not zero_based:
        tri = tri - 1
    nf = tri.shape[0]
    rc = L.lib().ms_p1_triangle_divergence_host(
        pos.shape[0], nf, _pd(pos), _pd(tilts), _pi(tri), _out(div_tri, (nf,), "div_tri"),
        _out(area, (nf,), "area"), _out(g0, (nf, 3), "g0"), _out(g1, (nf, 3), "g1"), _out(g2, (nf, 3), "g2"))
    L.check(rc, None, "ms_p1_triangle_divergence_host")


def _curvature(pos, tri, k_vecs, vertex_areas, weights, zero_based=1, va0=None, va1=None, va2=None):
    """tilt_kernels.f90:88-190 (optional per-corner outputs)."""
    pos, tri = _in(pos, np.float64), _in(tri, np.int32)
    if not zero_based:
        tri = tri - 1
    nv, nf = pos.shape[0], tri.shape[0]
    opt = [None if a is None else _out(a, (nf,), "va") for a in (va0, va1, va2)]
    rc = L.lib().ms_compute_curvature_data_host(
        nv, nf, _pd(pos), _pi(tri), _out(k_vecs, (nv, 3), "k_vecs"),
        _out(vertex_areas, (nv,), "vertex_areas"), _out(weights, (nf, 3), "weights"), *opt)
    L.check(rc, None, "ms_compute_curvature_data_host")


def get_surface_energy_kernel() -> KernelSpec | None:
    return None if _disabled() else KernelSpec(_surface, False)


def get_bending_grad_cotan_kernel() -> KernelSpec | None:
    return None if _disabled() else KernelSpec(_grad_cotan, False)


def get_bending_laplacian_kernel() -> KernelSpec | None:
    return None if _disabled() else KernelSpec(_laplacian, False)


def get_tilt_divergence_kernel() -> KernelSpec | None:
    return None if _disabled() else KernelSpec(_divergence, False)


def get_tilt_curvature_kernel() -> KernelSpec | None:
    return None if _disabled() else KernelSpec(_curvature, False)
