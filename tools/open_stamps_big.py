"""Diagnostic (library built with -DREINA_OPEN_STAMPS): where the day-opening launch spends its time at a large size."""
import copy, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
from reina_model_amd import simulation
from reina_model_amd.variables import VARIABLE_DEFAULTS
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
v, ages = bench.scaled_scenario(copy.deepcopy(VARIABLE_DEFAULTS), n)
ctx = simulation.make_context(v, age_counts=ages, seed=0)
for lo, hi in ((0, 118), (118, 200), (200, 365)):
    ctx.engine.tensors['mirror'].zero_()
    ctx.run(hi - lo); ctx.synchronize()
    m = ctx.engine.alloc.to_host(ctx.engine.tensors['mirror']).astype(np.float64)
    d = hi - lo
    us = m / 100.0
    print('days %3d-%3d: opening until flag %.1f us/day, opening total %.1f, weekly-import workgroup %.1f | tracing workgroups: level0 loop mean %.1f (slowest ever %.1f, n=%d), wait %.1f, flush %.1f, level1 %.1f' % (
        lo, hi, us[5] / d, us[6] / d, us[9] / d, us[1] / max(1, m[25]), us[17], int(m[25]), us[2] / max(1, m[26]), us[3] / max(1, m[27]), us[4] / max(1, m[28])), flush=True)
    print('      slowest wave ever, us since the block started -- level 0: queue entry read %.1f, its record / slots / word in %.1f, flags set %.1f, overflow list walked %.1f | level 1: %.1f %.1f %.1f %.1f' % tuple(us[40:48]), flush=True)
    print('      weekly-import workgroup, us/day: set-up %.1f propose %.1f any %.1f verify %.1f any %.1f (left the rounds %.1f) infect %.1f tail %.1f wait-open %.1f flush %.1f' % tuple(
        us[48 + k] / d for k in (0, 1, 2, 3, 15, 4, 5, 6, 7, 8)), flush=True)
