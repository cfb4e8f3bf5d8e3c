"""Diagnostic (library built with -DREINA_INSTALL_STAMPS): where k_hosp_install's installing workgroups spend their time, by window of
the scenario.  Stamps accumulate in buffers.mirror (100 MHz ticks): [k] summed over workgroups, [4 + k] the slowest workgroup ever; k = 0 prologue, 1 work, 2 flush.  usage: python tools/stamps_install.py <agents>"""
import copy, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
from reina_model_amd import simulation, datasets
from reina_model_amd.variables import VARIABLE_DEFAULTS
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
if n > 2_000_000:
    v, ages = bench.scaled_scenario(copy.deepcopy(VARIABLE_DEFAULTS), n)
else:
    v, ages = copy.deepcopy(VARIABLE_DEFAULTS), datasets.get_population_for_area()
ctx = simulation.make_context(v, age_counts=ages, seed=0)
for lo, hi in ((0, 60), (60, 85), (85, 125), (125, 160), (160, 365)):
    ctx.engine.tensors['mirror'].zero_()
    ctx.run(hi - lo)
    ctx.synchronize()
    m = ctx.engine.alloc.to_host(ctx.engine.tensors['mirror']).astype(np.float64)[:32] / 100.0
    d = hi - lo
    print('days %3d-%3d installing workgroups: sum/day prologue %.0f work %.0f flush %.0f, slowest workgroup %.1f %.1f %.1f' % (
        lo, hi, m[0] / d, m[1] / d, m[2] / d, m[4], m[5], m[6]), flush=True)
