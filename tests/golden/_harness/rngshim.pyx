# cython: language_level=3
"""Harness-only shim: exposes the reference RandomPool's cdef methods (simrandom.pyx:24-55) to
Python so known-answer vectors can be recorded at exactly the boundary the simulator uses."""
from cythonsim.simrandom cimport RandomPool
import numpy as np


def draw_pattern(RandomPool rp, str pattern, double a=0.0, double b=0.0):
    """pattern chars: d=get() double, u=getint() uint32, l=lognormal(a,b), g=gamma(a,b) float."""
    cdef int i, n = len(pattern)
    out = np.empty(n, dtype=np.float64)
    cdef double[::1] o = out
    cdef float fa = a, fb = b
    for i in range(n):
        c = pattern[i]
        if c == 'd':
            o[i] = rp.get()
        elif c == 'u':
            o[i] = <double> rp.getint()
        elif c == 'l':
            o[i] = rp.lognormal(a, b)
        elif c == 'g':
            o[i] = <double> rp.gamma(fa, fb)
        else:
            raise ValueError(c)
    return out


def chance_pattern(RandomPool rp, ps):
    return np.array([1 if rp.chance(p) else 0 for p in ps], dtype=np.int32)
