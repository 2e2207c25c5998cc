"""Extended-precision evaluation of the surface + bending energy and gradient (libms_oracle_ld.so: oracle/ms_oracle.c
with every double replaced by the x87 80-bit long double).

TEST INFRASTRUCTURE ONLY (tests/ and tools/): the full-size parity test measures the fp64 oracle's and the HIP path's
gradient errors against this evaluation instead of against each other.  The functions restated are the ones the fp64
wrappers of oracle/ms_oracle.py cite: fortran_kernels/surface_energy.f90:27-99 and
modules/energy/bending.py:90-181 with bending_gradient.py:17-175.
"""

from __future__ import annotations

import ctypes
import os

import numpy as np

from . import ms_oracle as orc

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None
_LD = ctypes.POINTER(ctypes.c_longdouble)


def available() -> bool:
    """x86-64: numpy.longdouble is the 80-bit extended type stored in 16 bytes, like C's long double."""
    return np.finfo(np.longdouble).nmant == 63 and np.dtype(np.longdouble).itemsize == ctypes.sizeof(ctypes.c_longdouble)


def lib():
    global _lib
    if _lib is None:
        orc.build()
        _lib = ctypes.CDLL(os.path.join(_HERE, "libms_oracle_ld.so"))
        _lib.orc_bending_energy_and_gradient.restype = ctypes.c_int
    return _lib


def _ld(a):
    return np.ascontiguousarray(a, dtype=np.longdouble)


def _p(a):
    return None if a is None else a.ctypes.data_as(_LD)


def surface_bending_energy_and_gradient(pos, tri, gamma, kappa, c0, is_boundary, *, model="helfrich", mode="analytic",
                                        surface=True, bending=True):
    """-> (E, grad) as numpy.longdouble: the module loop of oracle.minimizer_port.energy_and_gradient for
    ["surface", "bending"] (no constraint row, no fixed rows), every operation in extended precision."""
    if not available():
        raise RuntimeError("numpy.longdouble is not the x87 extended type on this platform")
    pos_l = _ld(pos)
    tri = np.ascontiguousarray(tri, dtype=np.int32)
    nv, nf = pos_l.shape[0], tri.shape[0]
    grad = np.zeros((nv, 3), dtype=np.longdouble)
    E = np.longdouble(0)
    pi = tri.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
    if surface:
        e = ctypes.c_longdouble(0.0)
        lib().orc_surface_energy_and_gradient(ctypes.c_int(nv), ctypes.c_int(nf), _p(pos_l), pi, _p(_ld(gamma)),
                                              _p(grad), ctypes.byref(e))
        E = E + np.longdouble(e.value)
    if bending:
        e = ctypes.c_longdouble(0.0)
        isb = np.ascontiguousarray(is_boundary, dtype=np.uint8)
        rc = lib().orc_bending_energy_and_gradient(
            ctypes.c_int(nv), ctypes.c_int(nf), _p(pos_l), pi, _p(_ld(kappa)), _p(_ld(c0)),
            isb.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), ctypes.c_int(orc._MODEL[model]),
            ctypes.c_int(orc._MODE[mode]),
            _p(grad), ctypes.byref(e), None, None, None)
        if rc != 0:
            raise MemoryError("orc_bending_energy_and_gradient (long double build) failed")
        E = E + np.longdouble(e.value)
    return E, grad
