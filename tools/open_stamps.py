"""Diagnostic (build with -DREINA_OPEN_STAMPS): phases of the day-opening launch, HUS."""
import copy, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from reina_model_amd import datasets, simulation
from reina_model_amd.variables import VARIABLE_DEFAULTS
ctx = simulation.make_context(copy.deepcopy(VARIABLE_DEFAULTS), age_counts=datasets.get_population_for_area(), seed=0)
ctx.run(130); ctx.synchronize()
ctx.engine.tensors['mirror'].zero_()
ctx.run(200); ctx.synchronize()
m = ctx.engine.alloc.to_host(ctx.engine.tensors['mirror']).astype(np.float64) / 100.0 / 200
print('us/day: level0 loop %.1f, wait %.1f, flush+last %.1f, level1 %.1f | opening until flag %.1f, opening total %.1f' % (m[1], m[2], m[3], m[4], m[5], m[6]))
