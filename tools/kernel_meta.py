"""Resource table of every kernel of the built library (code-object metadata): VGPRs, SGPRs, spills,
scratch, static LDS.  usage: python tools/kernel_meta.py [path/to/lib.so]"""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS = '/opt/rocm/lib/llvm/bin'
KEYS = ('.vgpr_count:', '.sgpr_count:', '.vgpr_spill_count:', '.sgpr_spill_count:', '.private_segment_fixed_size:',
        '.group_segment_fixed_size:')


def kernel_meta(lib):
    with tempfile.TemporaryDirectory() as tmp:
        so = os.path.join(tmp, 'lib.so')
        shutil.copy(lib, so)
        subprocess.run([os.path.join(TOOLS, 'llvm-objdump'), '--offloading', so], check=True, capture_output=True, cwd=tmp)
        co = [f for f in os.listdir(tmp) if 'gfx950' in f]
        notes = subprocess.run([os.path.join(TOOLS, 'llvm-readelf'), '--notes', os.path.join(tmp, co[0])],
                               check=True, capture_output=True, text=True).stdout
    out, name, fields = {}, None, {}
    for line in notes.splitlines():
        line = line.strip()
        for key in KEYS:
            if line.startswith(key):
                fields[key] = int(line.split()[-1])
        if line.startswith('.name:'):
            name = line.split()[-1]
        if line.startswith('.wavefront_size:') and name:
            out[name] = fields
            fields, name = {}, None
    return out


if __name__ == '__main__':
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'reina_model_amd', 'csrc', 'libreina_hip.so')
    print('%-28s %5s %5s %6s %6s %8s %8s' % ('kernel', 'vgpr', 'sgpr', 'vspill', 'sspill', 'scratch', 'lds'))
    for k, v in sorted(kernel_meta(lib).items()):
        short = k[2:].split('PK')[0] if k.startswith('_Z') else k
        print('%-28s %5d %5d %6d %6d %8d %8d' % (short[:28], *[v.get(x, 0) for x in KEYS]))
