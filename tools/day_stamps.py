"""Diagnostic (-DREINA_DAY_STAMPS): thread 0 of every k_day workgroup, time since its first instruction -- tables staged and the
barrier behind them passed / its wave's tiles and rounds done / last contacts resolved and counts written / workgroup barrier
before the counter flush.  python tools/day_stamps.py [agents]"""
import copy, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
from reina_model_amd import simulation, datasets
from reina_model_amd.variables import VARIABLE_DEFAULTS
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1685983
if n > 2_000_000:
    v, ages = bench.scaled_scenario(copy.deepcopy(VARIABLE_DEFAULTS), n)
else:
    v, ages = copy.deepcopy(VARIABLE_DEFAULTS), datasets.get_population_for_area()
ctx = simulation.make_context(v, age_counts=ages, seed=0)
for lo, hi in ((0, 5), (5, 25), (25, 90), (90, 110), (110, 300), (300, 365)):
    ctx.engine.tensors['mirror'].zero_()
    ctx.run(hi - lo); ctx.synchronize()
    m = ctx.engine.alloc.to_host(ctx.engine.tensors['mirror']).astype(np.float64)
    c = max(1.0, m[60])
    print('days %3d-%3d: mean over workgroups, us since the first instruction: tables staged %.2f | stream + rounds done %.2f | last contacts resolved %.2f | barrier before the flush %.2f' % (
        lo, hi, m[56] / c / 100, m[57] / c / 100, m[58] / c / 100, m[59] / c / 100), flush=True)
