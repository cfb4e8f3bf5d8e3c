"""Build the HIP engine in-tree: `python -m reina_model_amd.build` -> csrc/libreina_hip.so.

hipcc cross-compiles gfx950 without a GPU.  FP contraction is OFF: the engine's integer results
must be bit-identical to the CPU checker, which needs the same rounding sequence on both sides."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(CSRC, 'libreina_hip.so')
SOURCES = [os.path.join(CSRC, 'reina_hip.hip')]
DEPS = SOURCES + [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(('.inc', '.h'))] + [
                  os.path.join(os.path.dirname(HERE), 'include', 'reina_hip.h')]
HIPCC_FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-shared', '-std=c++17', '-ffp-contract=off',
               '-fno-fast-math']


def hipcc_path():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    return 'hipcc'


def build(force=False, verbose=False):
    if (not force and os.path.exists(LIB)
            and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in DEPS)):
        return LIB
    cmd = [hipcc_path()] + HIPCC_FLAGS + ['-o', LIB] + SOURCES
    if verbose:
        print(' '.join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv, verbose=True)
    print(LIB)
