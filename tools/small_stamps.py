"""Diagnostic (-DREINA_SMALL_STAMPS): where k_small_day's workgroups spend a day -- mean and longest time per phase over the
workgroups and days of a window: opening | first barrier | stream + contacts | second barrier | installs.
python tools/small_stamps.py [wgs]"""
import copy, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from reina_model_amd import simulation, datasets
from reina_model_amd.variables import VARIABLE_DEFAULTS
v, ages = copy.deepcopy(VARIABLE_DEFAULTS), datasets.get_population_for_area()
ctx = simulation.make_context(v, age_counts=ages, seed=0)
names = ('opening', 'barrier 1', 'stream + contacts', 'barrier 2', 'installs', 'barrier 3')
for lo, hi in ((0, 5), (5, 25), (25, 90), (90, 110), (110, 300), (300, 365)):
    ctx.engine.tensors['mirror'].zero_()
    ctx.run(hi - lo); ctx.synchronize()
    m = ctx.engine.alloc.to_host(ctx.engine.tensors['mirror']).astype(np.float64)
    c = max(1.0, m[120])
    print('days %3d-%3d (%d workgroup-days): ' % (lo, hi, c) + ' | '.join('%s %.2f (max %.2f)' % (n, m[100 + k] / c / 100, m[110 + k] / 100) for k, n in enumerate(names)) +
          ' | sum of means %.2f us' % (sum(m[100:106]) / c / 100), flush=True)
