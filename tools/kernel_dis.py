"""Disassembly of one kernel of a built library to stdout: python tools/kernel_dis.py k_hosp_install [lib.so]"""
import os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS = '/opt/rocm/lib/llvm/bin'
kernel = sys.argv[1]
lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, 'reina_model_amd', 'csrc', 'libreina_hip.so')
with tempfile.TemporaryDirectory() as tmp:
    so = os.path.join(tmp, 'lib.so')
    shutil.copy(lib, so)
    subprocess.run([os.path.join(TOOLS, 'llvm-objdump'), '--offloading', so], check=True, capture_output=True, cwd=tmp)
    co = [f for f in os.listdir(tmp) if 'gfx950' in f][0]
    dis = subprocess.run([os.path.join(TOOLS, 'llvm-objdump'), '-d', os.path.join(tmp, co)], check=True, capture_output=True, text=True).stdout
inside = False
for line in dis.splitlines():
    m = re.match(r'^[0-9a-f]+ <(.*)>:$', line)
    if m:
        inside = kernel in m.group(1) and not m.group(1).endswith('.kd')
    if inside:
        print(line)
