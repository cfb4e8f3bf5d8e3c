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
    assert f["abi_version"]() == eng.ABI_VERSION == 7


def test_struct_layouts_match_the_header_sizes():
    # sizes computed from the header's field lists (all 4/8-byte naturally aligned members)
    A, V, E = eng.MAX_AGES, eng.MAX_VARIANTS, eng.MAX_ENTRIES
    # (4 words, the 64-bit seed, 7 words, age_start, 4 words of exact attribution, the shard_age_start pointer)
    assert ctypes.sizeof(eng.Config) == 4 * 4 + 8 + 7 * 4 + 4 * (A + 1) + 4 * 4 + 8
    assert eng.Config.shard_age_start.offset % 8 == 0
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


def test_kernels_fit_the_lds_they_ask_for():
    """static + dynamic LDS of the two big kernels within a CU's 160 KB (hipFuncSetAttribute fails at engine
    creation otherwise -- on the GPU box only), and no kernel spills vector registers to scratch; read from
    the code object's metadata"""
    import shutil
    import subprocess
    import tempfile
    from reina_model_amd import build
    tools = '/opt/rocm/lib/llvm/bin'
    if not os.path.exists(os.path.join(tools, 'llvm-objdump')):
        pytest.skip('no llvm tools')
    build.build()
    with tempfile.TemporaryDirectory() as tmp:
        so = os.path.join(tmp, 'lib.so')
        shutil.copy(build.LIB, so)
        subprocess.run([os.path.join(tools, 'llvm-objdump'), '--offloading', so], check=True, capture_output=True, cwd=tmp)
        co = [f for f in os.listdir(tmp) if 'gfx950' in f]
        assert co, os.listdir(tmp)
        notes = subprocess.run([os.path.join(tools, 'llvm-readelf'), '--notes', os.path.join(tmp, co[0])],
                               check=True, capture_output=True, text=True).stdout
    kernels = {}
    name = None
    fields = {}
    for line in notes.splitlines():
        line = line.strip()
        for key in ('.group_segment_fixed_size:', '.vgpr_spill_count:', '.private_segment_fixed_size:', '.vgpr_count:'):
            if line.startswith(key):
                fields[key] = int(line.split()[-1])
        if line.startswith('.name:'):
            name = line.split()[-1]
        if line.startswith('.wavefront_size:') and name:
            kernels[name] = fields
            fields, name = {}, None
    day = [v for k, v in kernels.items() if 'k_day' in k]
    hosp = [v for k, v in kernels.items() if 'k_hosp_install' in k]
    # (single engine / engine group / shard under exact attribution; k_day in its two forms, dense and sparse stream)
    assert len(day) == 6 and len(hosp) == 3, sorted(kernels)
    LDS = 160 * 1024
    # dynamic requests: reina_hip.hip (day_shared_bytes(REINA_LDS_ROWS, sharded), REINA_MAX_HOSP_EVENTS * 8)
    for h in hosp:
        assert h['.group_segment_fixed_size:'] + eng.MAX_HOSP_EVENTS * 8 <= LDS
    text = open(os.path.join(ROOT, 'reina_model_amd', 'csrc', 'k_contacts.inc')).read()
    assert 'struct DayShared' in text
    assert all(d['.group_segment_fixed_size:'] <= 3072 for d in day)     # (its big arrays are carved from the dynamic part: the
    #                                                          static_assert in k_contacts.inc leaves 3 KB for the rest)
    # k_day addresses v104..v127 by hand (three tiles in flight, inline asm): the kernel descriptor must allocate them
    assert all(d['.vgpr_count:'] == 128 for d in day), day
    for k, v in kernels.items():
        # no kernel touches scratch (round 6: the engine-group instantiation of the day's last launch parked 8 registers in 28
        # bytes of it until its member reference came through the constant address space: k_common.inc MEMBER_OF_LAUNCH)
        # (k_small_days, the one-launch form of a small population's days that is OFF by default, is the exception: the loop over
        # its days carries the three phases' invariants in spilled registers -- part of why it measured slower, k_small.inc)
        if 'k_small_days' in k:
            continue
        assert v.get('.vgpr_spill_count:', 0) == 0 and v.get('.private_segment_fixed_size:', 0) == 0, (k, v)


def test_k_day_keeps_its_hand_reserved_registers_to_itself():
    """k_day holds three in-flight tiles in v104..v127 by hand (inline asm; the kernel is compiled for 104 VGPRs and the
    descriptor allocates 128).  That is sound only while (a) nothing is CALLED from the kernel -- a callee could use any
    register --, (b) nothing is spilled to scratch, and (c) no instruction the compiler emitted touches v104 and above: the
    only writers are the asm block's global_load_dwordx4, the only readers its v_bfe_u32 (round-2 advisor finding)."""
    import re
    import shutil
    import subprocess
    import tempfile
    from reina_model_amd import build
    tools = '/opt/rocm/lib/llvm/bin'
    assert os.path.exists(os.path.join(tools, 'llvm-objdump')), 'llvm-objdump is part of the ROCm image: this test does not skip'
    build.build()
    with tempfile.TemporaryDirectory() as tmp:
        so = os.path.join(tmp, 'lib.so')
        shutil.copy(build.LIB, so)
        subprocess.run([os.path.join(tools, 'llvm-objdump'), '--offloading', so], check=True, capture_output=True, cwd=tmp)
        co = [f for f in os.listdir(tmp) if 'gfx950' in f][0]
        dis = subprocess.run([os.path.join(tools, 'llvm-objdump'), '-d', os.path.join(tmp, co)], check=True, capture_output=True, text=True).stdout
        notes = subprocess.run([os.path.join(tools, 'llvm-readelf'), '--notes', os.path.join(tmp, co)], check=True, capture_output=True, text=True).stdout
    inside, body = False, []
    for line in dis.splitlines():
        m = re.match(r'^[0-9a-f]+ <(.*)>:$', line)
        if m:
            inside = 'k_day' in m.group(1) and not m.group(1).endswith('.kd')
            continue
        if inside and line.strip():
            body.append(line.split('//')[0].strip())
    assert len(body) > 1000
    high = re.compile(r'\bv(?:\[)?(1(?:0[4-9]|1[0-9]|2[0-7]))\b')   # v104..v127, alone or as the start of a range
    loads = reads = moves = 0
    for ins in body:
        op = ins.split()[0]
        assert not op.startswith(('s_swappc', 's_call', 's_setpc')), 'k_day calls out: %s' % ins
        assert not op.startswith('scratch_'), 'k_day touches scratch: %s' % ins
        # (a flat_* access counts against vmcnt AND lgkmcnt and may retire out of order with the global loads: the counted
        # wait in front of a tile's registers, s_waitcnt vmcnt(4), would no longer mean that the tile has landed)
        assert not op.startswith('flat_'), 'k_day uses a generic-address access: %s' % ins
        regs = [int(x) for x in re.findall(r'\bv\[?(\d+)', ins)]
        if not any(r >= 104 for r in regs):
            # (a range that starts below 104 must not reach into the reserved set either)
            for a, b in re.findall(r'v\[(\d+):(\d+)\]', ins):
                assert int(b) < 104, ins
            continue
        if op == 'global_load_dword':   # (a sparse day's fetch of the queued agents' hot words)
            assert re.match(r'global_load_dword\s+v124,', ins), ins
            loads += 1
        elif op == 'global_load_dwordx4':
            dst = re.match(r'global_load_dwordx4\s+v\[(\d+):(\d+)\]', ins)
            assert dst and 104 <= int(dst.group(1)) and int(dst.group(2)) <= 127, ins
            loads += 1
        else:
            # readers: the dense day's v_bfe_u32 (ACTIVE bit of a hot word) and the sparse day's v_mov_b32 (a set of bit words
            # taken into compiler-visible registers), both written in the asm blocks
            assert op in ('v_bfe_u32', 'v_mov_b32_e32'), 'a compiler-scheduled instruction touches the reserved tile registers: %s' % ins
            ops = [x.strip(' ,') for x in ins.split()[1:]]
            assert not re.match(r'v\[?1(0[4-9]|1\d|2[0-7])', ops[0]), '%s writes a reserved register: %s' % (op, ins)
            reads += 1
            moves += op == 'v_mov_b32_e32'
    assert loads >= 12 and reads >= 24 + 24 and moves >= 24, (loads, reads, moves)
    # kernel descriptor: no scratch at all
    name, fields = None, {}
    for line in notes.splitlines():
        line = line.strip()
        if line.startswith('.private_segment_fixed_size:') or line.startswith('.vgpr_spill_count:') or line.startswith('.sgpr_spill_count:'):
            fields[line.split(':')[0]] = int(line.split()[-1])
        if line.startswith('.name:'):
            name = line.split()[-1]
        if line.startswith('.wavefront_size:'):
            if name and 'k_day' in name:
                assert fields.get('.private_segment_fixed_size', 0) == 0 and fields.get('.vgpr_spill_count', 0) == 0, fields
            name, fields = None, {}
