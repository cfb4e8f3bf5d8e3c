"""Instruction histogram of one kernel of libreina_hip.so (gfx950 code object, llvm-objdump -d):
python tools/kernel_isa.py k_day [top]   -- works on a COPY of the library in a temporary directory (llvm-objdump
--offloading writes the extracted bundles next to its input)."""
import collections
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS = '/opt/rocm/lib/llvm/bin'
kernel = sys.argv[1] if len(sys.argv) > 1 else 'k_day'
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
with tempfile.TemporaryDirectory() as tmp:
    so = os.path.join(tmp, 'lib.so')
    shutil.copy(os.path.join(ROOT, 'reina_model_amd', 'csrc', 'libreina_hip.so'), so)
    subprocess.run([os.path.join(TOOLS, 'llvm-objdump'), '--offloading', so], check=True, capture_output=True, cwd=tmp)
    co = [f for f in os.listdir(tmp) if 'gfx950' in f][0]
    dis = subprocess.run([os.path.join(TOOLS, 'llvm-objdump'), '-d', os.path.join(tmp, co)], check=True, capture_output=True, text=True).stdout
hist = collections.Counter()
inside = False
n = 0
for line in dis.splitlines():
    m = re.match(r'^[0-9a-f]+ <(.*)>:$', line)
    if m:
        inside = kernel in m.group(1) and not m.group(1).endswith('.kd')
        continue
    if inside:
        parts = line.split()
        if parts and re.match(r'^[sv]_|^(global|buffer|ds|flat|scratch)_', parts[0]):
            hist[parts[0]] += 1
            n += 1
print('%s: %d static instructions' % (kernel, n))
for k, v in hist.most_common(top):
    print('  %-28s %5d' % (k, v))
