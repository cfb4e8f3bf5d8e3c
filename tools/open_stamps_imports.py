"""Diagnostic (-DREINA_OPEN_STAMPS): the import placement's phases on the import days of the scaled scenario"""
import copy, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
from reina_model_amd import simulation
from reina_model_amd.variables import VARIABLE_DEFAULTS
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
v, ages = bench.scaled_scenario(copy.deepcopy(VARIABLE_DEFAULTS), n)
ctx = simulation.make_context(v, age_counts=ages, seed=0)
for lo, hi in ((0, 20), (20, 21), (21, 196), (196, 197)):
    ctx.engine.tensors['mirror'].zero_()
    ctx.run(hi - lo); ctx.synchronize()
    m = ctx.engine.alloc.to_host(ctx.engine.tensors['mirror']).astype(np.float64) / 100.0
    print('days %d-%d: opening until flag %.1f us, total %.1f | placement: propose %.1f verify %.1f infect %.1f flush %.1f' % (lo, hi, m[5], m[6], m[10], m[11], m[12], m[13]), flush=True)
