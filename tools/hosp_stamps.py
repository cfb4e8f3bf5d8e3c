"""Diagnostic (build with -DREINA_HOSP_STAMPS): where the bed/ICU event walk spends its time."""
import copy, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from reina_model_amd import simulation
from reina_model_amd.variables import VARIABLE_DEFAULTS
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
v, ages = bench.scaled_scenario(copy.deepcopy(VARIABLE_DEFAULTS), n)
ctx = simulation.make_context(v, age_counts=ages, seed=0)
names = ['-', 'count+prefix', 'place', 'sort', 'walk', 'apply', 'flush', '-']
prev = np.zeros(8)
for lo, hi in ((0, 80), (80, 130), (130, 365)):
    ctx.engine.tensors['mirror'].zero_()
    ctx.run(hi - lo)
    ctx.synchronize()
    m = ctx.engine.alloc.to_host(ctx.engine.tensors['mirror']).astype(np.float64) / 100.0 / (hi - lo)
    print('days %d-%d us/day:' % (lo, hi), ' '.join('%s %.1f' % (names[k], m[k]) for k in range(1, 7)), 'total %.1f' % m[1:7].sum(), flush=True)
