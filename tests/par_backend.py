"""Test-side glue: drives oracle B (oracle/reina_par.c, the CPU restatement of the parallel day
step) through the SAME host code as the product (`reina_model_amd.model.Context`) by handing it
an `engine_factory` that binds the `par_*` symbols with host memory.  Test infrastructure only."""
import ctypes
import os
import subprocess

from reina_model_amd import engine as eng

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, 'oracle')
LIB = os.path.join(ORACLE, 'libreina_par.so')


def build(force=False):
    src = os.path.join(ORACLE, 'reina_par.c')
    deps = [src, os.path.join(ROOT, 'include', 'reina_hip.h'),
            os.path.join(ROOT, 'reina_model_amd', 'csrc', 'reina_prims.h'),
            os.path.join(ROOT, 'reina_model_amd', 'csrc', 'reina_contacts.h')]
    if (not force and os.path.exists(LIB)
            and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps)):
        return LIB
    subprocess.run(['gcc', '-O2', '-fPIC', '-shared', '-ffp-contract=off', '-fno-fast-math', '-std=gnu11',
                    '-o', LIB, src, '-lm'], check=True)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(LIB)
    return _lib


def par_engine_factory(cfg, disease):
    return eng.Engine(lib(), 'par_', eng.NumpyAllocator(), cfg, disease)
