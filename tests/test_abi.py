"""The C-ABI library loads on a CPU-only box and exports every symbol include/reina_hip.h
declares (no compute calls here); the product path fails LOUDLY without a GPU."""
import ctypes
import os
import re

import pytest

from reina_model_amd import engine as eng

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, 'include', 'reina_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(reina_[a-z_]+)\s*\(', text)))


def test_header_declares_the_abi_the_binding_uses():
    declared = _declared_functions()
    assert declared == sorted('reina_' + f for f in eng.ABI_FUNCTIONS)


def test_library_exports_every_declared_symbol():
    from reina_model_amd import build
    build.build()
    lib = eng.load_hip_library()
    for name in _declared_functions():
        assert hasattr(lib, name), name
    f = eng.bind_abi(lib, 'reina_')
    assert f['abi_version']() == 1


def test_struct_layouts_match_the_header_sizes():
    # sizes computed from the header's field lists (all 4/8-byte naturally aligned members)
    A, V, E = eng.MAX_AGES, eng.MAX_VARIANTS, eng.MAX_ENTRIES
    assert ctypes.sizeof(eng.Config) == 4 * 4 + 8 + 6 * 4 + 4 * (A + 1) + 4  # tail padding to 8
    assert ctypes.sizeof(eng.Disease) == 4 * (11 * V + V * 24 + V * A + 5 * A + 1 + 3 * 16)
    assert ctypes.sizeof(eng.Buffers) == 8 * len(eng.BUFFER_FIELDS)
    assert ctypes.sizeof(eng.Day) == 8 * 4 + 16 * 16 + 16 * 16 + 8


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    cfg, dis = eng.Config(), eng.Disease()
    cfg.n_agents, cfg.nr_ages, cfg.nr_variants = 16, 2, 1
    with pytest.raises(eng.EngineError):
        eng.hip_engine(cfg, dis)


def test_product_package_never_references_the_oracle():
    pkg = os.path.join(ROOT, 'reina_model_amd')
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(('.py', '.hip', '.h', '.inc', '.cpp')):
                text = open(os.path.join(dirpath, fn)).read()
                assert 'import oracle' not in text and 'from oracle' not in text, fn
                assert 'libreina_par' not in text and 'libreina_seq' not in text, fn
