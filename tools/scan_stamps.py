import sys, os, copy, shutil
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np
shutil.copy('reina_model_amd/csrc/libreina_hip_stamps.so', 'reina_model_amd/csrc/libreina_hip.so')
import bench
from reina_model_amd import simulation, engine as eng
from reina_model_amd.variables import VARIABLE_DEFAULTS
v, ages = bench.scaled_scenario(copy.deepcopy(VARIABLE_DEFAULTS), 50_000_000)
ctx = simulation.make_context(v, age_counts=ages, seed=0)
base = eng.C_NR * eng.MAX_AGES
prev = np.zeros(6)
for start, n in ((0, 5), (5, 70), (75, 30), (105, 95), (200, 40)):
    ctx.run(n, record_history=False)
    c = ctx.engine.read_counters()[base:]
    cur = np.array([c[12], c[13], c[14], c[15], c[28], c[29]], dtype=np.float64)
    d = cur - prev; prev = cur
    waves = d[5]
    us = d[:5] * 1024 / 100.0 / max(waves, 1)   # s_memtime ticks at 100 MHz -> us per wave per day-sum
    print('days %3d-%3d  per wave per day: load %.1f us  rounds %.1f us  exp %.1f us  ill %.1f us  ev %.1f us' % (
        start, start + n, *(us)))
